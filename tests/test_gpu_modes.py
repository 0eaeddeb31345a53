"""gvamp_main_real run modes (main_real.cpp:129-594): test, both, pvals-calc, restart, predict, predict_single, and
--init-est, each against the CPU oracle evaluated on the same files."""
import os
import re
import subprocess

import numpy as np
import pytest

from gvamp_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "gvamp_amd", "gvamp_main_real")
N, NT, M = 600, 301, 900
PROBS, VARS = "0.9,0.1", "0,0.01"


def run(args):
    r = subprocess.run([EXE] + [str(a) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    return r.stdout


def read_phen_scaled(path):
    """data::read_phen (data.cpp:128-192): 3rd column, NA -> masked, values * sqrt((n-1)/sum (y-mean)^2)."""
    raw, na = [], []
    for line in open(path):
        t = line.split()
        na.append(t[2] == "NA")
        raw.append(0.0 if t[2] == "NA" else float(t[2]))
    raw, na = np.array(raw), np.array(na)
    avg = raw[~na].mean()
    sqn = np.sqrt((np.sum(~na) - 1) / np.sum((raw[~na] - avg) ** 2))
    return raw * sqn, na, avg, sqn


@pytest.fixture(scope="module")
def world(tmp_path_factory, oracle):
    d = tmp_path_factory.mktemp("modes")
    rng = np.random.default_rng(5)
    bed = synth.synth_bed(N, M, seed=301, miss_ppm=5000)
    bed_t = synth.synth_bed(NT, M, seed=302, miss_ppm=5000)
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
    chrom = np.repeat(np.arange(1, 10), M // 9)
    with open(d / "m.bim", "w") as f:
        for i, ch in enumerate(chrom):
            f.write("%d\trs%d\t0\t%d\tA\tG\n" % (ch, i, i + 1))
    out = str(d / "out") + "/"
    base = ["--bed-file", d / "tr.bed", "--phen-files", d / "tr.phen", "--N", N, "--Mt", M, "--out-dir", out, "--probs",
            PROBS, "--vars", VARS, "--rho", "0.5", "--CG-max-iter", "20", "--seed", "4"]
    run(["--run-mode", "infere", "--out-name", "r", "--iterations", "3"] + base)
    return dict(d=d, out=out, bed=bed, bed_t=bed_t, base=base, chrom=chrom.astype(np.int32))


def oracle_test_r2(oracle, w, x_est, intercept=0.0, scale=1.0):
    yt, na, _, _ = read_phen_scaled(w["d"] / "te.phen")
    mave, msig = oracle.marker_stats(w["bed_t"], NT, M)
    z = oracle.ax(w["bed_t"], NT, M, mave, msig, x_est * np.sqrt(NT))[:NT]
    err2 = np.sum((yt - (intercept + scale * z)) ** 2)
    sd2 = (np.sum(yt ** 2) - NT * yt.mean() ** 2) / (NT - 1)
    return 1 - err2 / (sd2 * NT), err2


def test_mode_test_single_and_range(world, oracle):
    w = world
    targs = ["--bed-file-test", w["d"] / "te.bed", "--phen-files-test", w["d"] / "te.phen", "--N-test", NT, "--Mt-test", M]
    out = run(["--run-mode", "test", "--estimate-file", w["out"] + "r_it_3.bin"] + targs)
    r2 = float(re.search(r"test R2 = ([-0-9.e+]+)", out).group(1))
    o_r2, o_err2 = oracle_test_r2(oracle, w, np.fromfile(w["out"] + "r_it_3.bin"))
    assert np.isclose(r2, o_r2, rtol=1e-5) and r2 > 0.05
    assert np.isclose(float(re.search(r"test l2 pred err\^2 = ([-0-9.e+]+)", out).group(1)), o_err2, rtol=1e-5)
    out = run(["--run-mode", "test", "--estimate-file", w["out"] + "r_it_1.bin", "--test-iter-range", "1,3"] + targs)
    vals = [float(v) for v in re.search(r"\n([-0-9.e+, ]+), \n", out).group(1).split(", ")]
    ref = [oracle_test_r2(oracle, w, np.fromfile(w["out"] + "r_it_%d.bin" % k))[0] for k in (1, 2, 3)]
    assert np.allclose(vals, ref, rtol=1e-4, atol=1e-6)
    assert int(re.search(r"max ind = (\d+)", out).group(1)) == int(np.argmax(ref)) + 1


def test_mode_both(world, oracle):
    w = world
    out = run(["--run-mode", "both", "--out-name", "b", "--iterations", "3", "--bed-file-test", w["d"] / "te.bed",
               "--phen-files-test", w["d"] / "te.phen", "--N-test", NT, "--Mt-test", M] + w["base"])
    _, _, avg, sqn = read_phen_scaled(w["d"] / "tr.phen")
    assert np.isclose(float(re.search(r"intercept = ([-0-9.e+]+)", out).group(1)), avg, rtol=1e-5)
    assert np.isclose(float(re.search(r"scale = ([-0-9.e+]+)", out).group(1)), sqn, rtol=1e-5)
    x = np.fromfile(w["out"] + "b_it_3.bin")
    assert np.allclose(x, np.fromfile(w["out"] + "r_it_3.bin"), rtol=0, atol=0)      # same run as infere: deterministic
    o_r2, _ = oracle_test_r2(oracle, w, x, intercept=avg, scale=sqn)
    assert np.isclose(float(re.search(r"test R2 = ([-0-9.e+]+)", out).group(1)), o_r2, rtol=1e-5, atol=1e-8)


def test_mode_pvals_calc(world, oracle):
    w = world
    run(["--run-mode", "pvals-calc", "--out-name", "p", "--estimate-file", w["out"] + "r_it_3.bin", "--bim-file",
         w["d"] / "m.bim", "--store-pvals", "0"] + w["base"])
    y, na, _, _ = read_phen_scaled(w["d"] / "tr.phen")
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(~na)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(np.sum(~na))
    x1 = np.fromfile(w["out"] + "r_it_3.bin") * np.sqrt(N)
    mave, msig = oracle.marker_stats(w["bed"], N, M, mask4=m4, nonas=nonas)
    z1 = oracle.ax(w["bed"], N, M, mave, msig, x1, mask4=m4)
    yp = np.zeros(z1.size)
    yp[:N] = y * ~na
    assert np.allclose(np.fromfile(w["out"] + "p_pvals.bin"),
                       oracle.pvals(w["bed"], N, M, z1, yp, x1, mask4=m4, nonas=nonas), rtol=1e-7)
    assert np.allclose(np.fromfile(w["out"] + "p_pvals_LOCO.bin"),
                       oracle.pvals(w["bed"], N, M, z1, yp, x1, chrom=w["chrom"], mask4=m4, nonas=nonas), rtol=1e-7)


def _oracle_run(oracle, w, **kw):
    y, na, _, _ = read_phen_scaled(w["d"] / "tr.phen")
    raw = np.array([0.0 if t.split()[2] == "NA" else float(t.split()[2]) for t in open(w["d"] / "tr.phen")])
    return oracle.infere(w["bed"], N, M, raw, [0.9, 0.1], [0, 0.01], CG_max_iter=20, rho=0.5, seed=4,
                         is_na=na.astype(np.uint8), **kw)


def test_mode_restart(world, oracle):
    """--run-mode restart: gam1 / gamw from the command line, r1 reloaded from the stored r1 file (vamp.cpp:226-233)."""
    w = world
    run(["--run-mode", "restart", "--out-name", "s", "--iterations", "2", "--estimate-file", w["out"] + "r_r1_it_3.bin",
         "--gam1-init", "0.8", "--gamw-init", "1.7"] + w["base"])
    ref = _oracle_run(oracle, w, iterations=2, gam1=0.8, gamw=1.7, r1_init=np.fromfile(w["out"] + "r_r1_it_3.bin"))
    mine = np.fromfile(w["out"] + "s_it_2.bin")
    assert np.linalg.norm(mine - ref.x1[1]) / np.linalg.norm(ref.x1[1]) < 1e-7


def test_init_est(world, oracle):
    """--init-est 1: x1_hat = r1 = estimate * sqrt(N) at iteration 1 (vamp.cpp:244-258, :295-296)."""
    w = world
    run(["--run-mode", "infere", "--out-name", "e", "--iterations", "2", "--init-est", "1", "--estimate-file",
         w["out"] + "r_it_2.bin"] + w["base"])
    ref = _oracle_run(oracle, w, iterations=2, gam1=1e-6, gamw=2.0, x_init=np.fromfile(w["out"] + "r_it_2.bin"))
    mine = np.fromfile(w["out"] + "e_it_2.bin")
    assert np.linalg.norm(mine - ref.x1[1]) / np.linalg.norm(ref.x1[1]) < 1e-7
    assert np.allclose(np.fromfile(w["out"] + "e_it_1.bin"), np.fromfile(w["out"] + "r_it_2.bin"), rtol=1e-14)


def test_modes_predict(world, oracle):
    w = world
    targs = ["--bed-file-test", w["d"] / "te.bed", "--N-test", NT, "--Mt-test", M, "--out-dir", w["out"]]
    run(["--run-mode", "predict_single", "--out-name", "q", "--estimate-file", w["out"] + "r_it_3.bin"] + targs)
    mave, msig = oracle.marker_stats(w["bed_t"], NT, M)
    z = oracle.ax(w["bed_t"], NT, M, mave, msig, np.fromfile(w["out"] + "r_it_3.bin") * np.sqrt(NT))
    assert np.allclose(np.loadtxt(w["out"] + "q_predict.csv"), z, rtol=2e-5, atol=1e-7)       # csv: 6 digits
    # predict: <prefix>temp_<it>_<it>_gibbs_est.<ext> files over an iteration range, one csv per individual
    for it in (1, 2):
        np.fromfile(w["out"] + "r_it_%d.bin" % (it + 1)).tofile(w["out"] + "gtemp_%d_%d_gibbs_est.bin" % (it, it))
    run(["--run-mode", "predict", "--out-name", "g", "--estimate-file", w["out"] + "gtemp_1_1_gibbs_est.bin",
         "--test-iter-range", "1,2"] + targs)
    z2 = oracle.ax(w["bed_t"], NT, M, mave, msig, np.fromfile(w["out"] + "r_it_2.bin") * np.sqrt(NT))
    for i in (0, 7, NT - 1):
        row = np.loadtxt(w["out"] + "g_predict_%d.csv" % i)
        assert np.allclose(row, [z2[i], z[i]], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("kmode", [0, 1])
def test_driver_xxt_denoiser(world, oracle, kmode):
    """--use-XXT-denoiser 1 through the driver.  Kernel mode 1 keeps only the stripes resident (data.cpp ingest), so the
    people statistics come from the stripes as well."""
    w = world
    run(["--run-mode", "infere", "--out-name", "x%d" % kmode, "--iterations", "3", "--use-XXT-denoiser", "1",
         "--kernel-mode", kmode] + w["base"])
    x = np.fromfile(w["out"] + "x%d_it_3.bin" % kmode)
    ref = _oracle_run(oracle, w, iterations=3, use_XXT_denoiser=1)
    assert np.linalg.norm(x - ref.x1[2]) / np.linalg.norm(ref.x1[2]) < 1e-6
