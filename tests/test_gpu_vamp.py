"""Full vamp::infere() runs of the product (host C++ mirror + HIP kernels) against the CPU oracle and against the
outputs of the real reference (tests/golden/survey_probe).  north_star tolerance: x_hat within 1e-5 relative l2."""
import lzma
import os
import subprocess

import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden", "survey_probe")
XHAT_TOL = 1e-5      # BASELINE.json north_star
TIGHT = 1e-7         # what is actually observed is ~1e-9 (g1d at gam1 = 1e-8 cancels 8 digits, vamp.cpp:866)
PROBS, VARS = [0.90, 0.07, 0.03], [0, 0.001, 0.01]


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


@pytest.mark.parametrize("mode", [0, 1])
def test_sim_run_vs_oracle(oracle, mode):
    """BASELINE config 1 shape: N=2000, M=10000, 3 mixture components, seeded synthetic .bed and sim.cpp phenotype."""
    N, M = 2000, 10000
    bed = synth.synth_bed(N, M, seed=2024, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 500, 7, nthreads=4)
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, iterations=4, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8, gamw=2.0,
                        true_signal=beta)
    with capi.Shard(N, M, anchor=(mode == 0)) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(mode)
        b2, y2 = hostapi.sim_phen(sh, 0.5, 500, 7)
        assert np.allclose(b2, beta, rtol=1e-13, atol=0) and rel(y2, y) < 1e-13
        r = hostapi.infere_linear(sh, y, PROBS, VARS, iterations=4, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8,
                                  gamw=2.0, true_signal=beta)
    assert r.niter == ref.niter == 4
    for it in range(4):
        assert rel(r.x1[it], ref.x1[it]) < TIGHT if it else np.all(r.x1[0] == 0)
        assert rel(r.x2[it], ref.x2[it]) < TIGHT
        t, o = r.trace[it], ref.trace[it]
        assert (t["cg_iters"], t["onsager_iters"], t["revar_rounds"], t["L_after"]) == \
               (o["cg_iters"], o["onsager_iters"], o["revar_rounds"], o["L_after"])
        for f in ("gam1_denoise", "alpha1", "eta1", "gam2", "alpha2", "eta2", "gam2_reest", "gam1_next", "gamw", "rho"):
            assert np.isclose(t[f], o[f], rtol=1e-6), (it, f, t[f], o[f])
        # vector products: the product skips the reference's print-only diagnostics (3 Ax, it > 1), the duplicate Ax of
        # err_measures(2) and recomputing the constant A^T y (it > 1); passes over the shard are fewer still, because the
        # LMMSE and the Onsager solve share them (gv_cg_solve2) and so do the two Ax of updateNoisePrec
        assert t["n_atx"] == o["n_atx"] - (1 if it else 0) and t["n_ax"] == o["n_ax"] - (4 if it else 1)
        k1, k2 = t["cg_iters"] + (1 if it else 0), t["onsager_iters"]
        if mode == 1:
            assert t["n_ax_pass"] == max(k1, k2) + 2 and t["n_atx_pass"] == max(k1, k2) + 1 + (0 if it else 1)
        else:
            assert t["n_ax_pass"] == t["n_ax"] and t["n_atx_pass"] == t["n_atx"]
    assert rel(r.x_est, ref.x_est) < XHAT_TOL and rel(r.x_est, ref.x_est) < TIGHT
    assert np.allclose(r.probs, ref.probs, rtol=1e-6) and np.allclose(r.vars, ref.vars, rtol=1e-6)


def test_na_phenotypes_and_ragged_N_vs_oracle(oracle):
    """read_phen semantics (scaling, NA -> mask) with N % 4 != 0, main_real settings (gam1 = 1e-6)."""
    N, M = 1999, 3000
    rng = np.random.default_rng(4)
    bed = synth.synth_bed(N, M, seed=31, miss_ppm=10000)
    raw = rng.standard_normal(N) * 2.0 + 0.3
    is_na = rng.random(N) < 0.01
    ref = oracle.infere(bed, N, M, np.where(is_na, 0.0, raw), PROBS, VARS, iterations=3, CG_max_iter=20, rho=0.5, seed=3,
                        gam1=1e-6, gamw=2.0, is_na=is_na.astype(np.uint8))
    # host-side read_phen restated here only to build the inputs of the adopted-context entry point
    avg = raw[~is_na].mean()
    sqn = np.sqrt((np.sum(~is_na) - 1) / np.sum((raw[~is_na] - avg) ** 2))
    y = np.where(is_na, np.inf, raw * sqn)
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(~is_na)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    for mode in (0, 1):
        with capi.Shard(N, M, anchor=(mode == 0)) as sh:
            sh.upload_bed(bed)
            sh.set_kernel_mode(mode)
            r = hostapi.infere_linear(sh, y, PROBS, VARS, iterations=3, CG_max_iter=20, rho=0.5, seed=3, gam1=1e-6,
                                      gamw=2.0, mask4=m4, nonas=int(np.sum(~is_na)))
        assert rel(r.x_est, ref.x_est) < TIGHT, mode
        assert rel(r.x2[2], ref.x2[2]) < TIGHT


def test_default_23_component_prior_run(oracle):
    """No --probs / --vars: initialize_prior (utilities.cpp:91-140) needs Mt > 50000; components merge (L shrinks)."""
    N, M = 400, 50400
    bed = synth.synth_bed(N, M, seed=8, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 300, 2, nthreads=4)
    ref = oracle.infere(bed, N, M, y, None, None, iterations=3, CG_max_iter=15, rho=0.3, seed=2, gam1=1e-8, gamw=2.0,
                        nthreads=4)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        r = hostapi.infere_linear(sh, y, None, None, iterations=3, CG_max_iter=15, rho=0.3, seed=2, gam1=1e-8, gamw=2.0)
    assert [t["L_after"] for t in r.trace] == [int(t["L_after"]) for t in ref.trace]
    assert rel(r.x_est, ref.x_est) < XHAT_TOL


def _toy_bed(tmp_path):
    raw = lzma.open(os.path.join(G, "toy.bed.xz")).read()
    p = tmp_path / "toy.bed"
    p.write_bytes(raw)
    return str(p)


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_gvamp_sim_executable_vs_reference_outputs(tmp_path, mode):
    """The reference-style driver with the reference's own command line, compared with the .bin files the real
    reference wrote for it (survey probe, np = 1); --kernel-mode 0 the fp64 VALU family, 1 the default, 2 the two-level fixed point."""
    bed = _toy_bed(tmp_path)
    out = str(tmp_path / "out") + "/"
    cmd = [os.path.join(ROOT, "gvamp_amd", "gvamp_sim"), "--bed-file", bed, "--N", "2000", "--Mt", "10000",
           "--out-dir", out, "--out-name", "toy", "--iterations", "3", "--num-mix-comp", "3", "--probs", "0.90,0.07,0.03",
           "--vars", "0,0.001,0.01", "--CV", "500", "--h2", "0.5", "--rho", "0.5", "--CG-max-iter", "20", "--model",
           "linear", "--seed", "7", "--store-pvals", "0", "--kernel-mode", str(mode)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert np.array_equal(np.fromfile(out + "toy_beta_true.bin"), np.fromfile(os.path.join(G, "sim_beta_true.bin")))
    for name in ("it_1_x2_hat", "it_3", "it_3_x2_hat", "r1_it_3"):
        mine = np.fromfile(out + "toy_%s.bin" % name)
        ref = np.fromfile(os.path.join(G, "sim_np1_%s.bin" % name))
        assert rel(mine, ref) < XHAT_TOL, name
        assert rel(mine, ref) < TIGHT, name
    assert np.allclose(np.loadtxt(out + "toy_gam1s.csv"), np.loadtxt(os.path.join(G, "sim_np1_gam1s.csv")), rtol=1e-5)
    assert np.allclose(np.loadtxt(out + "toy_gam2s.csv"), np.loadtxt(os.path.join(G, "sim_np1_gam2s.csv")), rtol=1e-5)


def test_gvamp_main_real_executable_vs_reference_outputs(tmp_path):
    """main_real --run-mode infere with the NA-bearing toy.phen, against the real reference's (scalar build) outputs."""
    bed = _toy_bed(tmp_path)
    out = str(tmp_path / "outr") + "/"
    cmd = [os.path.join(ROOT, "gvamp_amd", "gvamp_main_real"), "--run-mode", "infere", "--model", "linear", "--bed-file",
           bed, "--phen-files", os.path.join(G, "toy.phen"), "--N", "2000", "--Mt", "10000", "--out-dir", out,
           "--out-name", "r", "--iterations", "3", "--probs", "0.90,0.07,0.03", "--vars", "0,0.001,0.01", "--rho", "0.5",
           "--CG-max-iter", "20", "--seed", "7", "--h2", "0.5"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    for name in ("it_1_x2_hat", "it_3", "it_3_x2_hat", "r1_it_3"):
        mine = np.fromfile(out + "r_%s.bin" % name)
        ref = np.fromfile(os.path.join(G, "real_%s.bin" % name))
        assert rel(mine, ref) < TIGHT, name


def test_unknown_flag_and_missing_bed_are_fatal():
    exe = os.path.join(ROOT, "gvamp_amd", "gvamp_sim")
    r = subprocess.run([exe, "--no-such-flag", "1"], capture_output=True, text=True)
    assert r.returncode != 0 and "unknown" in r.stdout
    r = subprocess.run([exe, "--N", "10"], capture_output=True, text=True)
    assert r.returncode != 0 and "no bed file" in r.stdout


@pytest.mark.parametrize("nshards,mode,fuse", [(2, 1, 1), (8, 1, 1), (2, 0, 1), (2, 1, 2), (8, 1, 2), (4, 1, 0), (2, 1, 3), (8, 1, 4)])
def test_sharded_run_vs_real_reference_mpi_outputs(tmp_path, nshards, mode, fuse):
    """Marker-sharded VAMP exactly as the reference shards over MPI ranks (divide_work, one N-vector all-reduce per
    Ax, Hutchinson probe seeded seed + S): `nshards` contexts on this GPU joined by the in-process communicator,
    compared with what the REAL reference wrote at np = 2 and np = 8 (survey probe).  fuse = --fuse-solves level (2 is
    what bench.py runs on N GPUs); np = 4 has no reference files and is compared with the oracle's 4-shard run."""
    import threading
    raw = np.frombuffer(lzma.open(os.path.join(G, "toy.bed.xz")).read(), dtype=np.uint8)[3:]
    N, Mt = 2000, 10000
    mb = N // 4
    beta = np.fromfile(os.path.join(G, "sim_beta_true.bin"))
    # y as sim.cpp makes it: A (beta sqrt(N)) + noise, from a single-shard context
    with capi.Shard(N, Mt, anchor=(mode == 0)) as sh:
        sh.upload_bed(raw)
        sh.set_kernel_mode(mode)
        b, y = hostapi.sim_phen(sh, 0.5, 500, 7)
        assert np.allclose(b, beta, rtol=1e-13, atol=0)
    results, errors = [None] * nshards, []
    group = 1000 + 100 * fuse + 10 * nshards + mode

    def work(rank):
        try:
            size, modu = divmod(Mt, nshards)
            M = size + 1 if rank < modu else size
            S = sum(size + 1 if r < modu else size for r in range(rank))
            with capi.Shard(N, M, Mt=Mt, S=S, anchor=(mode == 0)) as sh:
                sh.upload_bed(raw[S * mb:(S + M) * mb])
                sh.set_kernel_mode(mode)
                sh.comm_init_local(group, nshards, rank)
                r = hostapi.infere_linear(sh, y, PROBS, VARS, iterations=3, CG_max_iter=20, rho=0.5, seed=7, gam1=1e-8,
                                          gamw=2.0, true_signal=beta[S:S + M], rank=rank, fuse_solves=fuse)
                results[rank] = (S, M, r)
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nshards)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    x1_3 = np.concatenate([res[2].x1[2] for res in results])
    x2_1 = np.concatenate([res[2].x2[0] for res in results])
    x2_3 = np.concatenate([res[2].x2[2] for res in results])
    r1_3 = np.concatenate([res[2].r1[2] for res in results])
    pre = os.path.join(G, "sim_np%d_" % nshards)
    if os.path.exists(pre + "it_3.bin"):
        assert rel(x2_1, np.fromfile(pre + "it_1_x2_hat.bin")) < TIGHT
        assert rel(x1_3, np.fromfile(pre + "it_3.bin")) < TIGHT
        assert rel(x2_3, np.fromfile(pre + "it_3_x2_hat.bin")) < TIGHT
        assert rel(r1_3, np.fromfile(pre + "r1_it_3.bin")) < TIGHT
    else:
        from oracle import gvoracle
        ref = gvoracle.infere(raw, N, Mt, y, PROBS, VARS, nshards=nshards, iterations=3, CG_max_iter=20, rho=0.5, seed=7,
                              gam1=1e-8, gamw=2.0, true_signal=beta)
        assert rel(x1_3, ref.x1[2]) < TIGHT and rel(x2_3, ref.x2[2]) < TIGHT and rel(r1_3, ref.r1[2]) < TIGHT
    # every rank saw the same scalars
    g = [[t["gamw"] for t in res[2].trace] for res in results]
    assert all(gg == g[0] for gg in g)


def test_freeze_mask_vs_oracle(tmp_path, oracle):
    """--use-freeze 1 --freeze-index-file (vamp.cpp:205-209,:308,:353): frozen markers are left out of alpha1 and are not
    damped.  Product (masks applied on the device) against the oracle, and against the unfrozen run."""
    N, M = 1500, 1200
    rng = np.random.default_rng(8)
    bed = synth.synth_bed(N, M, seed=21)
    freeze = (rng.random(M) < 0.3).astype(float)
    np.savetxt(tmp_path / "freeze.txt", freeze, fmt="%d")
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        beta, y = hostapi.sim_phen(sh, 0.5, 100, 5)
        kw = dict(iterations=4, CG_max_iter=30, rho=0.3, seed=5, gam1=1e-8, gamw=2.0, true_signal=beta)
        r = hostapi.infere_linear(sh, y, PROBS, VARS, freeze_index_file=str(tmp_path / "freeze.txt"), **kw)
        r_plain = hostapi.infere_linear(sh, y, PROBS, VARS, **kw)
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, freeze_ind=freeze, **kw)
    assert r.niter == ref.niter
    for it in range(r.niter):
        assert rel(r.x1[it], ref.x1[it]) < TIGHT and rel(r.x2[it], ref.x2[it]) < TIGHT
        for f in ("alpha1", "gam2", "alpha2", "gamw"):
            assert np.isclose(r.trace[it][f], ref.trace[it][f], rtol=1e-6), (it, f)
    assert rel(r.x_est, ref.x_est) < TIGHT
    assert rel(r_plain.x_est, ref.x_est) > 1e-3                  # the mask does change the run


def test_unbuilt_variants_are_refused_loudly(tmp_path):
    """--red 1 (CG on a sub-range of individuals, vamp.cpp:594) is not built: the driver says so instead of ignoring it."""
    N, M = 400, 300
    bedp = str(tmp_path / "t.bed")
    synth.write_bed(bedp, synth.synth_bed(N, M, seed=2))
    exe = os.path.join(ROOT, "gvamp_amd", "gvamp_sim")
    r = subprocess.run([exe, "--bed-file", bedp, "--N", str(N), "--Mt", str(M), "--out-dir", str(tmp_path) + "/", "--out-name",
                        "t", "--iterations", "1", "--probs", "0.9,0.1", "--vars", "0,0.01", "--CV", "10", "--h2", "0.5",
                        "--model", "linear", "--red", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--red" in r.stdout and "not built" in r.stdout


@pytest.mark.parametrize("fuse", [1, 2, 3, 4])
def test_long_run_stays_on_the_oracle(oracle, fuse):
    """20 iterations (the tests above stop at 3-6): the product must not drift away from the oracle over a long run --
    in particular the CG by-products of --fuse-solves 2, which replace explicit products by recurrences."""
    N, M = 1500, 2500
    bed = synth.synth_bed(N, M, seed=31)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        beta, y = hostapi.sim_phen(sh, 0.5, 120, 9)
        kw = dict(iterations=20, CG_max_iter=40, rho=0.5, seed=9, gam1=1e-8, gamw=2.0, true_signal=beta, stop_criteria_thr=1e-12)
        r = hostapi.infere_linear(sh, y, PROBS, VARS, fuse_solves=fuse, **kw)
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, **kw)
    assert r.niter == ref.niter == 20
    for it in range(r.niter):
        t, o = r.trace[it], ref.trace[it]
        assert (t["cg_iters"], t["onsager_iters"], t["L_after"]) == (o["cg_iters"], o["onsager_iters"], o["L_after"]), it
        assert np.isclose(t["gamw"], o["gamw"], rtol=1e-6) and np.isclose(t["gam1_next"], o["gam1_next"], rtol=1e-6), it
    worst = max(rel(r.x1[it], ref.x1[it]) for it in range(1, r.niter))
    assert worst < TIGHT and rel(r.x_est, ref.x_est) < TIGHT


def test_bench_prints_exactly_one_json_line():
    """The driver parses bench.py's stdout: one line, JSON, with the contract's keys -- at a tiny size, CPU baseline included."""
    import json
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--N", "20000", "--Mt", "60000", "--steps", "2", "--warmup", "1",
                        "--vamp-iterations", "2", "--cpu-markers", "500"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "vamp"):
        assert k in d, k
    assert d["steps"] == 2 and d["n_gpus"] == 1 and d["roofline"]["traffic"] is None     # no PMC profile for this size


def test_host_exceptions_stop_at_the_c_boundary():
    """initialize_prior (utilities.cpp:91-140) throws when no prior is given and Mt < 50 000; the reference lets that end the
    program.  Behind the flat C entry points it comes back as an error code and a message, not as an abort of the caller."""
    N, M = 300, 500
    bed = synth.synth_bed(N, M, seed=5)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        with pytest.raises(capi.GvError, match="Mt < 50,000"):
            hostapi.infere_linear(sh, np.zeros(N), None, None, iterations=1)
        r = hostapi.infere_linear(sh, np.random.default_rng(0).standard_normal(N), [0.9, 0.1], [0, 0.01], iterations=1)   # still usable
        assert r.niter == 1


def test_graft_entry_smoke_runs():
    """__graft_entry__.smoke() is what the driver runs on the GPU box before the bench: it must keep working whatever the
    library's defaults are (round 3 changed them under it)."""
    import importlib
    sys_path_added = ROOT not in __import__("sys").path
    if sys_path_added:
        __import__("sys").path.insert(0, ROOT)
    g = importlib.import_module("__graft_entry__")
    g.smoke()
