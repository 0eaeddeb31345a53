"""Unit properties of the CPU oracle (host logic, no GPU): the restated semantics behave as SURVEY App. A says."""
import numpy as np
import pytest

from gvamp_amd import synth


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def decode(bed, N, M):
    """independent numpy decode: a in {2,0,1,0}, b in {1,0,1,1} for PLINK codes 00,01,10,11 (dotp_lut.hpp)."""
    mb = (N + 3) // 4
    bits = np.unpackbits(np.asarray(bed, dtype=np.uint8).reshape(M, mb), axis=1, bitorder="little").reshape(M, mb * 4, 2)
    code = bits[:, :, 0] + 2 * bits[:, :, 1]
    a = np.array([2.0, 0.0, 1.0, 0.0])[code]
    b = np.array([1.0, 0.0, 1.0, 1.0])[code]
    return a[:, :N], b[:, :N]


def test_divide_work(oracle):
    # utilities.cpp:259-291: remainder to the low ranks, contiguous, covering
    for Mt, n in ((10000, 8), (10, 3), (7, 7), (5, 8)):
        parts = [oracle.divide_work(Mt, n, r) for r in range(n)]
        assert sum(m for m, _ in parts) == Mt
        assert all(parts[i][1] + parts[i][0] == parts[i + 1][1] for i in range(n - 1))
        assert max(m for m, _ in parts) - min(m for m, _ in parts) <= 1 and parts[0][0] >= parts[-1][0]


@pytest.mark.parametrize("N,M,fna", [(200, 37, 0.0), (203, 20, 0.05)])
def test_stats_ax_atx_against_dense_numpy(oracle, N, M, fna):
    rng = np.random.default_rng(N)
    bed = synth.synth_bed(N, M, seed=3, miss_ppm=20000)
    a, b = decode(bed, N, M)
    present = rng.random(N) >= fna
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(present.sum())
    mave, msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    mu = (a * b * present).sum(1) / (b * present).sum(1)
    sg = 1 / np.sqrt((((a - mu[:, None]) * b * present) ** 2).sum(1) / (nonas - 1))
    assert np.allclose(mave, mu, rtol=1e-13) and np.allclose(msig, sg, rtol=1e-12)
    A = ((a - mu[:, None]) * sg[:, None] * b).T / np.sqrt(N)          # N x M standardised design matrix
    x = rng.standard_normal(M)
    z = oracle.ax(bed, N, M, mave, msig, x, mask4=m4)
    assert rel(z[:N], (A * present[:, None]) @ x) < 1e-12 and np.all(z[N:] == 0)
    p = np.zeros(4 * ((N + 3) // 4))
    p[:N] = rng.standard_normal(N) * present
    assert rel(oracle.atx(bed, N, M, mave, msig, p), A.T @ p[:N]) < 1e-12
    # threads do not change the result (same per-element summation order)
    assert np.array_equal(oracle.ax(bed, N, M, mave, msig, x, mask4=m4, nthreads=4), z)


def test_g1_limits(oracle):
    r = np.linspace(-3, 3, 13)
    g1, g1d = oracle.g1_g1d(r, 1e12, [0.9, 0.1], [0.0, 1.0])          # |sigma| < 1e-10 -> identity (vamp.cpp:813,844)
    assert np.array_equal(g1, r) and np.all(g1d == 1)
    g1, g1d = oracle.g1_g1d(r, 2.0, [1.0], [0.0])                      # pure spike at 0 shrinks everything to 0
    assert np.allclose(g1, 0, atol=1e-15)
    g1, _ = oracle.g1_g1d(r, 2.0, [0.0, 1.0], [0.0, 3.0])              # pure slab: Wiener gain v/(v + 1/gam1)
    assert np.allclose(g1, r * 3.0 / (3.0 + 0.5), rtol=1e-13)


def test_update_prior_merges_close_variances(oracle):
    rng = np.random.default_rng(0)
    r1 = rng.standard_normal(5000) * np.where(rng.random(5000) < 0.1, 2.0, 0.05)
    p, v = oracle.update_prior(r1, 5000, 5.0, [0.8, 0.1, 0.1], [0.0, 1.0, 1.2], EM_max_iter=1, learn_vars=0)
    assert len(p) == 2 and np.isclose(p.sum(), 1.0)                    # |1.2-1|/1 < 0.5 -> merged (vamp.cpp:1054-1071)
    p, v = oracle.update_prior(r1, 5000, 5.0, [0.8, 0.1, 0.1], [0.0, 1.0, 4.0], EM_max_iter=3)
    assert len(p) == 3 and np.isclose(p.sum(), 1.0) and v[0] == 0


def test_cg_solves_the_lmmse_system(oracle):
    N, M = 300, 120
    rng = np.random.default_rng(5)
    bed = synth.synth_bed(N, M, seed=9, miss_ppm=5000)
    a, b = decode(bed, N, M)
    mave, msig = oracle.marker_stats(bed, N, M)
    A = ((a - mave[:, None]) * msig[:, None] * b).T / np.sqrt(N)
    v = rng.standard_normal(M)
    tau, gam2 = 2.0, 0.7
    mu, rr = oracle.cg_solve(bed, N, M, v, None, tau, gam2, 1, 200)
    exact = np.linalg.solve(tau * A.T @ A + gam2 * np.eye(M), v)
    assert rr[-1] < 1e-5 and rel(mu, exact) < 1e-4
    assert np.all(np.diff(np.log(rr)) < 0.5)                           # residual trace essentially decreasing


def test_bern_vec_depends_on_shard_start(oracle):
    u0, u5 = oracle.bern_vec(7, 0, 100, 1000), oracle.bern_vec(7, 5, 100, 1000)   # mt19937{seed + S} (vamp.cpp:875)
    assert np.allclose(np.abs(u0), 1 / np.sqrt(1000)) and not np.array_equal(u0, u5)
    assert np.array_equal(oracle.bern_vec(12, 0, 100, 1000), u5)


def test_infere_stops_on_criterion_and_counts_matvecs(oracle):
    N, M = 400, 600
    bed = synth.synth_bed(N, M, seed=4)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 30, 3)
    r = oracle.infere(bed, N, M, y, [0.9, 0.1], [0, 0.01], iterations=30, CG_max_iter=10, rho=0.5, seed=3,
                      stop_criteria_thr=1e-2, true_signal=beta)
    assert 2 <= r.niter < 30                                           # vamp.cpp:745
    t = r.trace[1]
    # per iteration (it > 1): Ax = K1 + K2 + 8 incl. warm start, ATx = K1 + K2 + 3
    assert t["n_ax"] == t["cg_iters"] + t["onsager_iters"] + 1 + 7 and t["n_atx"] == t["cg_iters"] + t["onsager_iters"] + 3


def test_student_t_against_scipy(oracle):
    import scipy.stats as ss
    for nu in (1, 3, 58, 1998, 399998):
        for t in (1e-6, 0.5, 2.0, 5.0, 12.0, 40.0):
            assert np.isclose(oracle.student_t_two_sided(t, nu), 2 * ss.t.sf(t, nu), rtol=5e-10, atol=0)
    assert oracle.student_t_two_sided(0.0, 10) == 1.0


def test_pvals_loo_against_scipy_linregress(oracle):
    """data::pvals_calc (data.cpp:1108-1226): regress y - A_{-k} x_{-k} on the standardised column k."""
    import scipy.stats as ss
    N, M = 300, 25
    rng = np.random.default_rng(2)
    bed = synth.synth_bed(N, M, seed=13, miss_ppm=20000)
    a, b = decode(bed, N, M)
    mave, msig = oracle.marker_stats(bed, N, M)
    A = ((a - mave[:, None]) * msig[:, None] * b).T                  # N x M, not divided by sqrt(N)
    x1 = rng.standard_normal(M)
    z1 = oracle.ax(bed, N, M, mave, msig, x1)
    y = np.zeros(z1.size)
    y[:N] = z1[:N] + rng.standard_normal(N)
    pv = oracle.pvals(bed, N, M, z1, y, x1)
    for k in (0, 7, 24):
        keep = b[k] > 0                                              # individuals with a genotype at marker k
        ymark = (y[:N] - z1[:N]) + A[:, k] / np.sqrt(N) * x1[k]
        ref = ss.linregress(A[keep, k], ymark[keep]).pvalue
        assert np.isclose(pv[k], ref, rtol=1e-8)


def test_erfcx_and_probit_denoiser(oracle):
    from scipy.special import erfcx as s_erfcx
    from scipy.stats import norm
    for x in (-20.0, -3.0, -0.2, 0.0, 0.7, 5.0, 24.99, 25.01, 300.0, 1e6):
        assert np.isclose(oracle.erfcx(x), s_erfcx(x), rtol=1e-13)
    # g1_bin_class = posterior mean of z ~ N(p, 1/tau) given y = 1{z + N(0, probit_var) > 0} (vamp_probit.cpp:661-687)
    p, tau, pv = np.array([-1.2, 0.0, 0.4, 2.5]), 0.8, 1.0
    for yv in (0.0, 1.0):
        g, gd = oracle.probit_g(p, np.full(4, yv), tau, pv)
        s = np.sqrt(pv + 1 / tau)
        sgn = 2 * yv - 1
        c = p / s
        ref = p + sgn * norm.pdf(c) / norm.cdf(sgn * c) / tau / s
        assert np.allclose(g, ref, rtol=1e-12)
        h = 1e-5                                                       # g1d = d g1 / dp
        gp, _ = oracle.probit_g(p + h, np.full(4, yv), tau, pv)
        gm, _ = oracle.probit_g(p - h, np.full(4, yv), tau, pv)
        assert np.allclose(gd, (gp - gm) / (2 * h), rtol=1e-6)


def test_newton_method_cov_is_the_probit_regression_fit(oracle):
    """vamp_probit.cpp:936-1062 with gg = 0 and probit_var = 1 is the MLE of a probit regression; checked against an
    independent optimiser.  The relative-step exit (1e-4) returns the point before the last step, hence ~1e-6."""
    from scipy.optimize import minimize
    from scipy.stats import norm
    rng = np.random.default_rng(0)
    N, C = 4000, 4
    Z = np.c_[np.ones(N), rng.standard_normal((N, C - 1))]
    eta_true = np.array([0.3, -0.5, 0.8, 0.1])
    y = (Z @ eta_true + rng.standard_normal(N) > 0).astype(float)
    eta, ml, grad = oracle.newton_cov(Z, y)
    r = minimize(lambda b: -np.mean(norm.logcdf((2 * y - 1) * (Z @ b))), np.zeros(C), method="BFGS", tol=1e-12)
    assert np.allclose(eta, r.x, atol=5e-6)
    assert np.isclose(ml, r.fun, rtol=1e-10) and np.abs(grad).max() < 1e-6
    # with an offset the fit moves the intercept accordingly
    eta2, _, _ = oracle.newton_cov(Z, y, gg=np.full(N, 0.25))
    assert np.isclose(eta2[0], eta[0] - 0.25, atol=1e-4) and np.allclose(eta2[1:], eta[1:], atol=1e-4)


def test_lu_solve_and_covariate_offset(oracle):
    rng = np.random.default_rng(1)
    A, b = rng.standard_normal((6, 6)), rng.standard_normal(6)
    assert np.allclose(oracle.lu_solve(A, b), np.linalg.solve(A, b), rtol=1e-12)
    A[[0, 3]] = A[[3, 0]]
    A[0, 0] = 0.0                                         # forces a row exchange
    assert np.allclose(oracle.lu_solve(A, b), np.linalg.solve(A, b), rtol=1e-10)
    assert oracle.lu_solve(np.zeros((3, 3)), np.ones(3)) is None
    # m_cov shifts the likelihood argument only: g1(p, m) - p == g1(p + m, 0) - (p + m)   (vamp_probit.cpp:661-687)
    p, m = rng.standard_normal(50), rng.standard_normal(50)
    y = (rng.random(50) < 0.5).astype(float)
    g, gd = oracle.probit_g_cov(p, y, m, 0.7)
    g0, gd0 = oracle.probit_g(p + m, y, 0.7)
    assert np.allclose(g - p, g0 - (p + m), rtol=1e-12) and np.allclose(gd, gd0, rtol=1e-12)
