"""--model bin_class (vamp_probit.cpp, BASELINE config 4 without covariates): the probit z-denoiser kernel and full
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
    with capi.Shard(N, M) as sh:
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
