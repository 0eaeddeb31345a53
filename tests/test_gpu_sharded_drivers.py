"""The sharded DRIVER path on one GPU: gvamp_sim / gvamp_main_real started as N processes by scripts/run_sharded.py --comm host
--same-gpu -- every rank its own divide_work shard (utilities.cpp:259-291), its own slab of the .bed (file offset 3 + S*mbytes,
data.cpp:215), its own Hutchinson probe (seed + S, vamp.cpp:875), its own writes into the shared .bin files at byte offset S*8
(utilities.cpp:293-301), text files by rank 0 only -- with the sums of data.cpp:928/:995 and utilities.cpp:203 carried by the
shared-memory host transport (host/shm_comm.cpp) instead of RCCL, which needs one GPU per rank.  The files the ranks wrote are
compared with what the REAL reference wrote under `mpirun -np 2` (tests/golden/survey_probe) and with the oracle's sharded runs.

(The GPU boxes of this pool allow six processes on the card at once, the test runner included: five ranks at most here; the
reference's np = 8 outputs are covered by the in-process shards of tests/test_gpu_vamp.py.)"""
import lzma
import os
import re
import socket
import subprocess
import sys

import numpy as np
import pytest

from gvamp_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "survey_probe")
SIM = os.path.join(ROOT, "gvamp_amd", "gvamp_sim")
REAL = os.path.join(ROOT, "gvamp_amd", "gvamp_main_real")
TIGHT = 1e-7


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def sharded(n, exe, args, timeout=900):
    cmd = [sys.executable, os.path.join(ROOT, "scripts", "run_sharded.py"), "-n", str(n), "--comm", "host", "--same-gpu",
           "--master-port", str(_free_port()), "--", exe] + [str(a) for a in args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def _toy_bed(tmp_path):
    p = tmp_path / "toy.bed"
    p.write_bytes(lzma.open(os.path.join(G, "toy.bed.xz")).read())
    return str(p)


SIM_ARGS = ["--N", "2000", "--Mt", "10000", "--out-name", "toy", "--iterations", "3", "--num-mix-comp", "3", "--probs",
            "0.90,0.07,0.03", "--vars", "0,0.001,0.01", "--CV", "500", "--h2", "0.5", "--rho", "0.5", "--CG-max-iter", "20",
            "--model", "linear", "--seed", "7", "--store-pvals", "0"]


@pytest.mark.parametrize("kmode,fuse", [(1, 4), (1, 0), (1, 2), (0, 1)])
def test_gvamp_sim_two_processes_vs_real_reference_np2(tmp_path, kmode, fuse):
    """The reference's own command line under `mpirun -np 2`, here as two processes sharing GPU 0: the .bin files assembled by the
    two ranks (each wrote its M doubles at byte offset S*8) against the reference's, the rank-0 text files likewise."""
    bed = _toy_bed(tmp_path)
    out = str(tmp_path / "out") + "/"
    os.makedirs(out)
    stdout = sharded(2, SIM, ["--bed-file", bed, "--out-dir", out, "--kernel-mode", kmode, "--fuse-solves", fuse] + SIM_ARGS)
    assert "rank    0 has 5000 markers" in stdout.replace("INFO   : ", "")
    assert np.array_equal(np.fromfile(out + "toy_beta_true.bin"), np.fromfile(os.path.join(G, "sim_beta_true.bin")))
    for name in ("it_1_x2_hat", "it_3", "it_3_x2_hat", "r1_it_3"):
        mine = np.fromfile(out + "toy_%s.bin" % name)
        assert mine.size == 10000
        assert rel(mine, np.fromfile(os.path.join(G, "sim_np2_%s.bin" % name))) < TIGHT, name
    assert np.allclose(np.loadtxt(out + "toy_gam1s.csv"), np.loadtxt(os.path.join(G, "sim_np2_gam1s.csv")), rtol=1e-5)
    assert np.allclose(np.loadtxt(out + "toy_gam2s.csv"), np.loadtxt(os.path.join(G, "sim_np2_gam2s.csv")), rtol=1e-5)
    assert np.loadtxt(out + "toy_z1_it_2.csv").size == 2000            # one writer: a second one would interleave lines


@pytest.mark.parametrize("n", [3, 5])
def test_gvamp_sim_uneven_shards_vs_oracle(tmp_path, oracle, n):
    """Mt = 10000 over 3 ranks (3334 / 3333 / 3333: the remainder goes to the low ranks) and over 5: every rank's slab offset,
    probe seed and file offset differ; the assembled iterates against the oracle's run with the same shards."""
    bed = _toy_bed(tmp_path)
    out = str(tmp_path / "out") + "/"
    os.makedirs(out)
    sharded(n, SIM, ["--bed-file", bed, "--out-dir", out] + SIM_ARGS)
    raw = np.frombuffer(lzma.open(os.path.join(G, "toy.bed.xz")).read(), dtype=np.uint8)[3:]
    beta, y = oracle.sim_phen(raw, 2000, 10000, 0.5, 500, 7)
    ref = oracle.infere(raw, 2000, 10000, y, [0.90, 0.07, 0.03], [0, 0.001, 0.01], nshards=n, iterations=3, CG_max_iter=20,
                        rho=0.5, seed=7, gam1=1e-8, gamw=2.0, true_signal=beta)
    assert np.array_equal(np.fromfile(out + "toy_beta_true.bin"), beta)
    for k in (1, 2, 3):
        assert rel(np.fromfile(out + "toy_it_%d.bin" % k), ref.x1[k - 1]) < TIGHT, k
        assert rel(np.fromfile(out + "toy_it_%d_x2_hat.bin" % k), ref.x2[k - 1]) < TIGHT, k


N, NT, M = 600, 301, 901           # 901 markers: uneven over 2 and over 3 ranks
PROBS, VARS = "0.9,0.1", "0,0.01"


def read_phen_scaled(path):
    """data::read_phen (data.cpp:128-192)"""
    raw, na = [], []
    for line in open(path):
        t = line.split()
        na.append(t[2] == "NA")
        raw.append(0.0 if t[2] == "NA" else float(t[2]))
    raw, na = np.array(raw), np.array(na)
    avg = raw[~na].mean()
    sqn = np.sqrt((np.sum(~na) - 1) / np.sum((raw[~na] - avg) ** 2))
    return raw, raw * sqn, na, avg, sqn


@pytest.fixture(scope="module")
def world(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("sharded_modes")
    rng = np.random.default_rng(5)
    bed = synth.synth_bed(N, M, seed=311, miss_ppm=5000)
    bed_t = synth.synth_bed(NT, M, seed=312, miss_ppm=5000)
    synth.write_bed(str(d / "tr.bed"), bed)
    synth.write_bed(str(d / "te.bed"), bed_t)
    beta = rng.standard_normal(M) * (rng.random(M) < 0.05) * 0.15
    for name, b, n in (("tr", bed, N), ("te", bed_t, NT)):
        mave, msig = oracle.marker_stats(b, n, M)
        g = oracle.ax(b, n, M, mave, msig, beta * np.sqrt(n))[:n]
        y = 1.5 + 2.0 * (g + 0.7 * rng.standard_normal(n))
        with open(d / (name + ".phen"), "w") as f:
            for i in range(n):
                f.write("F%d I%d %s\n" % (i, i, "NA" if (name == "tr" and i % 97 == 5) else repr(float(y[i]))))
    out = str(d / "out") + "/"
    os.makedirs(out)
    base = ["--bed-file", d / "tr.bed", "--phen-files", d / "tr.phen", "--N", N, "--Mt", M, "--out-dir", out, "--probs",
            PROBS, "--vars", VARS, "--rho", "0.5", "--CG-max-iter", "20", "--seed", "4"]
    return dict(d=d, out=out, bed=bed, bed_t=bed_t, base=base)


def oracle_test_r2(oracle, w, x_est, intercept=0.0, scale=1.0):
    _, yt, na, _, _ = read_phen_scaled(w["d"] / "te.phen")
    mave, msig = oracle.marker_stats(w["bed_t"], NT, M)
    z = oracle.ax(w["bed_t"], NT, M, mave, msig, x_est * np.sqrt(NT))[:NT]
    err2 = np.sum((yt - (intercept + scale * z)) ** 2)
    sd2 = (np.sum(yt ** 2) - NT * yt.mean() ** 2) / (NT - 1)
    return 1 - err2 / (sd2 * NT), err2


def test_main_real_both_as_two_processes(world, oracle):
    """--run-mode both (main_real.cpp:214-283) sharded: TWO data objects per process -- the training shard, then the test shard --
    on the process's ONE communicator (host/data.cpp: gv_host_world); NA phenotypes, ragged N_test; the iterate the two ranks
    assembled against the oracle's 2-shard run, the printed test R2 against the oracle evaluated on that file."""
    w = world
    out = sharded(2, REAL, ["--run-mode", "both", "--out-name", "b", "--iterations", "3", "--bed-file-test", w["d"] / "te.bed",
                            "--phen-files-test", w["d"] / "te.phen", "--N-test", NT, "--Mt-test", M] + w["base"])
    raw, _, na, avg, sqn = read_phen_scaled(w["d"] / "tr.phen")
    ref = oracle.infere(w["bed"], N, M, raw, [0.9, 0.1], [0, 0.01], CG_max_iter=20, rho=0.5, seed=4, iterations=3,
                        is_na=na.astype(np.uint8), nshards=2, gam1=1e-6, gamw=2.0)
    x = np.fromfile(w["out"] + "b_it_3.bin")
    assert x.size == M and rel(x, ref.x1[2]) < TIGHT
    assert np.isclose(float(re.search(r"intercept = ([-0-9.e+]+)", out).group(1)), avg, rtol=1e-5)
    o_r2, _ = oracle_test_r2(oracle, w, x, intercept=avg, scale=sqn)
    assert np.isclose(float(re.search(r"test R2 = ([-0-9.e+]+)", out).group(1)), o_r2, rtol=1e-5, atol=1e-8)


def test_main_real_test_mode_as_three_processes(world, oracle):
    """--run-mode test (main_real.cpp:129-211) over three ranks: each reads its own slice of the estimate file (offset S*8) and its
    own slab of the test .bed; the predictor is summed over the ranks inside Ax; R2 over an iteration range and for one file."""
    w = world
    if not os.path.exists(w["out"] + "b_it_3.bin"):
        pytest.skip("needs the iterates of test_main_real_both_as_two_processes")
    targs = ["--bed-file-test", w["d"] / "te.bed", "--phen-files-test", w["d"] / "te.phen", "--N-test", NT, "--Mt-test", M]
    out = sharded(3, REAL, ["--run-mode", "test", "--estimate-file", w["out"] + "b_it_3.bin"] + targs)
    o_r2, o_err2 = oracle_test_r2(oracle, w, np.fromfile(w["out"] + "b_it_3.bin"))
    assert np.isclose(float(re.search(r"test R2 = ([-0-9.e+]+)", out).group(1)), o_r2, rtol=1e-5)
    assert np.isclose(float(re.search(r"test l2 pred err\^2 = ([-0-9.e+]+)", out).group(1)), o_err2, rtol=1e-5)
    out = sharded(3, REAL, ["--run-mode", "test", "--estimate-file", w["out"] + "b_it_1.bin", "--test-iter-range", "1,3"] + targs)
    vals = [float(v) for v in re.search(r"\n([-0-9.e+, ]+), \n", out).group(1).split(", ")]
    ref = [oracle_test_r2(oracle, w, np.fromfile(w["out"] + "b_it_%d.bin" % k))[0] for k in (1, 2, 3)]
    assert np.allclose(vals, ref, rtol=1e-4, atol=1e-6)
