"""--model bin_class (vamp_probit.cpp, BASELINE config 4, with and without covariates): the probit z-denoiser kernel and full
generalised-VAMP runs of the product against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest
from scipy.stats import norm

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBS, VARS = [0.9, 0.1], [0, 0.05]


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def make_case_control(oracle, N, M, seed):
    rng = np.random.default_rng(seed)
    bed = synth.synth_bed(N, M, seed=seed, miss_ppm=5000)
    mave, msig = oracle.marker_stats(bed, N, M)
    beta = rng.standard_normal(M) * (rng.random(M) < 0.05) * 0.2
    g = oracle.ax(bed, N, M, mave, msig, beta * np.sqrt(N))[:N]
    y = (rng.random(N) < norm.cdf(3 * g)).astype(float)
    return bed, beta, y


def test_probit_denoiser_kernel_vs_oracle(oracle):
    N = 5003
    rng = np.random.default_rng(3)
    p = np.zeros(4 * ((N + 3) // 4))
    p[:N] = rng.standard_normal(N) * np.where(rng.random(N) < 0.05, 30.0, 1.5)     # includes far tails (|c| ~ 40)
    y = np.zeros_like(p)
    y[:N] = rng.random(N) < 0.4
    with capi.Shard(N, 8) as sh:
        dp, dy, dz = sh.vecN(p), sh.vecN(y), sh.vecN()
        for tau1, pv in ((1e-8, 1.0), (0.7, 1.0), (25.0, 0.3)):
            sums = sh.probit_denoise(dp, dy, tau1, pv, dz)
            g, gd = oracle.probit_g(p[:N], y[:N], tau1, pv)
            z = dz.download()
            assert np.allclose(z[:N], g, rtol=1e-11, atol=1e-13) and np.all(z[N:] == 0)
            assert np.isclose(sums[0], gd.sum(), rtol=1e-10)
            assert np.isclose(sums[1], ((g - p[:N]) ** 2).sum(), rtol=1e-10)


@pytest.mark.parametrize("mode", [0, 1])
def test_probit_run_vs_oracle(oracle, mode):
    N, M = 1001, 1500
    bed, beta, y = make_case_control(oracle, N, M, 11)
    kw = dict(iterations=6, CG_max_iter=30, rho=0.5, seed=3, gam1=1e-8, gamw=1.0, model="bin_class")
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, **kw)
    with capi.Shard(N, M, anchor=(mode == 0)) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(mode)
        r = hostapi.infere_linear(sh, y, PROBS, VARS, **kw)
    assert r.niter == ref.niter
    for it in range(r.niter):
        t, o = r.trace[it], ref.trace[it]
        assert (t["cg_iters"], t["onsager_iters"], t["revar_rounds"]) == (o["cg_iters"], o["onsager_iters"], o["revar_rounds"])
        for f in ("gam1_denoise", "alpha1", "gam2", "alpha2", "gam1_next", "beta1", "tau2", "tau1_next"):
            assert np.isclose(t[f], o[f], rtol=1e-6), (it, f, t[f], o[f])
        assert rel(r.x2[it], ref.x2[it]) < 1e-7
    assert rel(r.x_est, ref.x_est) < 1e-7                        # unscaled x1_hat (vamp_probit.cpp:657)
    assert np.corrcoef(r.x_est, beta)[0, 1] > 0.6                # and it recovers the simulated effects


def test_gvamp_main_real_probit_executable(tmp_path, oracle):
    """Driver with a case/control .phen: read_phen scales y (data.cpp:172-182) -- restated here for the oracle input."""
    N, M = 800, 1000
    bed, _beta, y = make_case_control(oracle, N, M, 12)
    bedp = str(tmp_path / "c.bed")
    synth.write_bed(bedp, bed)
    with open(tmp_path / "c.phen", "w") as f:
        for i in range(N):
            f.write("F%d I%d %d\n" % (i, i, int(y[i])))
    out = str(tmp_path / "o") + "/"
    cmd = [os.path.join(ROOT, "gvamp_amd", "gvamp_main_real_probit"), "--run-mode", "infere", "--model", "bin_class",
           "--bed-file", bedp, "--phen-files", str(tmp_path / "c.phen"), "--N", str(N), "--Mt", str(M), "--out-dir", out,
           "--out-name", "c", "--iterations", "4", "--probs", "0.9,0.1", "--vars", "0,0.05", "--rho", "0.5",
           "--CG-max-iter", "30", "--seed", "3"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, iterations=4, CG_max_iter=30, rho=0.5, seed=3, gam1=1e-8, gamw=1.0,
                        model="bin_class", is_na=np.zeros(N, dtype=np.uint8))
    assert rel(np.fromfile(out + "c_probit_it_4.bin"), ref.x1[3]) < 1e-7
    assert rel(np.fromfile(out + "c_probit_r1_it_4.bin"), ref.r1[3]) < 1e-7


# ---- covariates (--C > 0, --cov-file): vamp_probit.cpp:84-87,:110-126,:347,:364; data.cpp:286-331,:1050-1058 -------------

def make_covariates(N, C, seed):
    rng = np.random.default_rng(seed)
    Z = np.c_[np.ones(N), rng.standard_normal((N, C - 1))]
    eta = np.array([0.4, -0.6, 0.9, 0.2, -0.3][:C])
    return Z, eta


def test_probit_denoiser_kernel_with_covariate_offset(oracle):
    N = 4097
    rng = np.random.default_rng(4)
    npad = 4 * ((N + 3) // 4)
    p, y, m = np.zeros(npad), np.zeros(npad), np.zeros(npad)
    p[:N] = rng.standard_normal(N) * 1.5
    y[:N] = rng.random(N) < 0.4
    m[:N] = rng.standard_normal(N) * np.where(rng.random(N) < 0.05, 20.0, 0.8)
    with capi.Shard(N, 8) as sh:
        dp, dy, dm, dz = sh.vecN(p), sh.vecN(y), sh.vecN(m), sh.vecN()
        for tau1, pv in ((1e-8, 1.0), (0.7, 1.0), (25.0, 0.3)):
            sums = sh.probit_denoise(dp, dy, tau1, pv, dz, m_cov=dm)
            g, gd = oracle.probit_g_cov(p[:N], y[:N], m[:N], tau1, pv)
            z = dz.download()
            assert np.allclose(z[:N], g, rtol=1e-11, atol=1e-13) and np.all(z[N:] == 0)
            assert np.isclose(sums[0], gd.sum(), rtol=1e-10)
            assert np.isclose(sums[1], ((g - p[:N]) ** 2).sum(), rtol=1e-10)
        # a zero offset is the plain kernel, bit for bit
        sh.probit_denoise(dp, dy, 0.7, 1.0, dz, m_cov=sh.vecN(np.zeros(npad)))
        z0 = dz.download()
        sh.probit_denoise(dp, dy, 0.7, 1.0, dz)
        assert np.array_equal(z0, dz.download())


def test_probit_run_with_covariates_vs_oracle(oracle):
    N, M, C = 1200, 1500, 4
    bed, beta, _y = make_case_control(oracle, N, M, 21)
    Z, eta = make_covariates(N, C, 21)
    rng = np.random.default_rng(22)
    mave, msig = oracle.marker_stats(bed, N, M)
    g = oracle.ax(bed, N, M, mave, msig, beta * np.sqrt(N))[:N]
    y = (rng.random(N) < norm.cdf(3 * g + Z @ eta)).astype(float)
    kw = dict(iterations=5, CG_max_iter=30, rho=0.5, seed=3, gam1=1e-8, gamw=1.0, model="bin_class", covs=Z)
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, **kw)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        r = hostapi.infere_linear(sh, y, PROBS, VARS, **kw)
        r0 = hostapi.infere_linear(sh, y, PROBS, VARS, **{**kw, "covs": None})
    # the covariate effects are the probit-regression fit of y on Z (offset 0): Newton of the product == of the oracle
    assert np.allclose(r.cov_eff, ref.cov_eff, rtol=1e-9, atol=1e-12)
    # (attenuated by the genetic term the fit does not see, but with the simulated signs)
    assert np.all(np.sign(r.cov_eff[:3]) == np.sign(eta[:3]))
    assert r.niter == ref.niter
    for it in range(r.niter):
        t, o = r.trace[it], ref.trace[it]
        assert (t["cg_iters"], t["onsager_iters"], t["revar_rounds"]) == (o["cg_iters"], o["onsager_iters"], o["revar_rounds"])
        for f in ("gam1_denoise", "alpha1", "gam2", "alpha2", "gam1_next", "beta1", "tau2", "tau1_next"):
            assert np.isclose(t[f], o[f], rtol=1e-6), (it, f, t[f], o[f])
    assert rel(r.x_est, ref.x_est) < 1e-7
    assert rel(r0.x_est, ref.x_est) > 1e-3                       # and the covariates do change the answer


def test_gvamp_main_real_probit_covariates_infere_and_test_modes(tmp_path, oracle):
    N, M, C = 900, 1000, 3
    bed, beta, _y = make_case_control(oracle, N, M, 31)
    Z, eta = make_covariates(N, C, 31)
    rng = np.random.default_rng(32)
    mave, msig = oracle.marker_stats(bed, N, M)
    g = oracle.ax(bed, N, M, mave, msig, beta * np.sqrt(N))[:N]
    y = (rng.random(N) < norm.cdf(3 * g + Z @ eta)).astype(float)
    bedp = str(tmp_path / "c.bed")
    synth.write_bed(bedp, bed)
    with open(tmp_path / "c.phen", "w") as f:
        for i in range(N):
            f.write("F%d I%d %d\n" % (i, i, int(y[i])))
    np.savetxt(tmp_path / "c.cov", Z, fmt="%.17g", delimiter=" ")
    out = str(tmp_path / "o") + "/"
    exe = os.path.join(ROOT, "gvamp_amd", "gvamp_main_real_probit")
    common = ["--model", "bin_class", "--N", str(N), "--Mt", str(M), "--cov-file", str(tmp_path / "c.cov"), "--C", str(C)]
    cmd = [exe, "--run-mode", "infere", "--bed-file", bedp, "--phen-files", str(tmp_path / "c.phen"), "--out-dir", out,
           "--out-name", "c", "--iterations", "4", "--probs", "0.9,0.1", "--vars", "0,0.05", "--rho", "0.5",
           "--CG-max-iter", "30", "--seed", "3"] + common
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, iterations=4, CG_max_iter=30, rho=0.5, seed=3, gam1=1e-8, gamw=1.0,
                        model="bin_class", is_na=np.zeros(N, dtype=np.uint8), covs=Z)
    x4 = np.fromfile(out + "c_probit_it_4.bin")
    assert rel(x4, ref.x1[3]) < 1e-7
    # the log carries the fitted effects (the reference prints them and stores nothing)
    import re
    printed = [float(v) for v in re.findall(r"cov_eff\[\d+\] = ([-+0-9.eE]+),", res.stdout)[:C]]
    assert np.allclose(printed, ref.cov_eff, rtol=1e-4)

    # --run-mode test on the same individuals, effects handed over through --cov-estimate-file
    np.savetxt(tmp_path / "c.coveff", ref.cov_eff, fmt="%.17g")
    cmd = [exe, "--run-mode", "test", "--bed-file-test", bedp, "--phen-files-test", str(tmp_path / "c.phen"),
           "--N-test", str(N), "--Mt-test", str(M), "--estimate-file", out + "c_probit_it_4.bin",
           "--cov-estimate-file", str(tmp_path / "c.coveff")] + common
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    # expected counts from the oracle's matvec; read_phen rescales y, so "y == 1" holds for nobody unless the scale is 1:
    # main_real_probit.cpp:141 compares the SCALED phenotype with 1 -- restated here as it is
    sd = np.sqrt(((y - y.mean()) ** 2).sum() / (N - 1))
    y_scaled = y / sd
    z = oracle.ax(bed, N, M, mave, msig, x4 * np.sqrt(N))[:N] + Z @ ref.cov_eff
    pred = norm.cdf(z) >= 0.5
    P = int((y_scaled == 1).sum())
    Ne = N - P
    line = [ln for ln in res.stdout.splitlines() if ln.startswith("P = ")][-1]
    assert line.startswith("P = %d, N = %d," % (P, Ne)), line
    fpr = float(line.split("FPR = ")[1].split(",")[0])
    assert np.isclose(fpr, pred[y_scaled != 1].sum() / Ne, rtol=1e-5)            # printed with 6 significant digits
    # what the classifier is worth against the unscaled labels
    assert (pred == (y == 1)).mean() > 0.7
