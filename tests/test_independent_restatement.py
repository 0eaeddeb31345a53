"""Independent dense-numpy restatements of the (f) rows of SURVEY section 8, written FROM THE REFERENCE SOURCE (file:line cited at
every function), not from oracle/: whole VAMP iterations on an explicit N x M design matrix with numpy / scipy only.  They pin
the oracle where nothing reference-generated can (the reference needs Boost to build): if oracle/gv_oracle.cpp misreads a
statement order, a clip, a damping or a scaling of these paths, these tests see it.

  * infere_bin_class        vamp_probit.cpp:20-658 (g1_bin_class / g1d_bin_class :661-705, erfcx via scipy.special)
  * --use-XXT-denoiser 1    vamp.cpp:261-760 with reverse == 1, denoiserXXT.cpp:15-130, compute_people_statistics data.cpp:558-716
  * p-values LOO / LOCO     data.cpp:1108-1226 / :1235-1353, linear_reg1d_pvals utilities.cpp:321-334 (Student t via scipy.stats)

CPU only; N ~ 200, M ~ 40."""
import numpy as np
import pytest
from scipy import special, stats

from gvamp_amd import synth

GMIN, GMAX = 1e-11, 1e11          # gamma_min / gamma_max, vamp.hpp:31-32


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def clip(x):
    return min(max(x, GMIN), GMAX)


# ---- the design matrix, densely (dotp_lut.hpp:3,1030; na_lut.hpp:3; data.cpp:392-546) -----------------------------------------
def decode(bed, N, M):
    mb = (N + 3) // 4
    bits = np.unpackbits(np.asarray(bed, dtype=np.uint8).reshape(M, mb), axis=1, bitorder="little").reshape(M, mb * 4, 2)
    code = bits[:, :, 0] + 2 * bits[:, :, 1]
    a = np.array([2.0, 0.0, 1.0, 0.0])[code]            # dotp_lut_a: 00 -> 2, 01 -> 0 (missing), 10 -> 1, 11 -> 0
    b = np.array([1.0, 0.0, 1.0, 1.0])[code]            # dotp_lut_b: 0 only for the missing code
    return a[:, :N], b[:, :N]


class Dense:
    """A = standardised genotypes / sqrt(N) as an explicit matrix, with the marker statistics of data.cpp:451-484"""

    def __init__(self, bed, N, M, present=None):
        self.N, self.M = N, M
        self.a, self.b = decode(bed, N, M)
        self.present = np.ones(N, bool) if present is None else present
        na = self.present.astype(float)
        nonas = int(self.present.sum())
        self.mave = (self.a * self.b * na).sum(1) / (self.b * na).sum(1)
        self.msig = 1.0 / np.sqrt((((self.a - self.mave[:, None]) * self.b * na) ** 2).sum(1) / (nonas - 1))
        self.G = ((self.a - self.mave[:, None]) * self.msig[:, None] * self.b * na).T        # N x M, not yet / sqrt(N)
        self.A = self.G / np.sqrt(N)

    def Ax(self, x):                                    # data.cpp:848-1009
        return self.A @ x

    def ATx(self, p):                                   # data.cpp:810-835
        return self.A.T @ p


# ---- libstdc++'s <random> on std::mt19937, as the reference uses it ------------------------------------------------------------
def _raw32(seed, n):
    """n outputs of std::mt19937{seed} (numpy's legacy seeding is the same init_genrand)"""
    return np.random.RandomState(seed).randint(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.float64)


def _canonical(raw, k):
    """std::generate_canonical<double, 53>: two 32-bit draws per value"""
    return (raw[2 * k] + raw[2 * k + 1] * 4294967296.0) / 18446744073709551616.0


def bern_probe(seed, S, M, Mt):
    """vamp.cpp:875-882: mt19937{seed + S}, bernoulli_distribution(0.5) (true iff canonical < p), (2 b - 1) / sqrt(Mt)"""
    raw = _raw32(seed + S, 2 * M)
    can = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0
    return (2.0 * (can < 0.5) - 1.0) / np.sqrt(Mt)


def simulate(n, eta, pi, seed):
    """utilities.cpp:48-88: per element a fresh mt19937{seed + i}; one uniform picks the component, then a fresh
    normal_distribution (Marsaglia polar: the FIRST value it returns is y * mult) draws the value"""
    out = np.zeros(n)
    for i in range(n):
        raw = _raw32(seed + i, 64)
        k = 0
        u = _canonical(raw, k)
        k += 1
        c = 0.0
        for j in range(len(eta)):
            c += pi[j]
            if u <= c:
                if eta[j] != 0:
                    while True:
                        x = 2.0 * _canonical(raw, k) - 1.0
                        y = 2.0 * _canonical(raw, k + 1) - 1.0
                        k += 2
                        r2 = x * x + y * y
                        if not (r2 > 1.0 or r2 == 0.0):
                            break
                    out[i] = y * np.sqrt(-2.0 * np.log(r2) / r2) * np.sqrt(eta[j])
                break
    return out


# ---- the signal-side denoiser and EM (vamp.cpp:805-869, :929-1072) --------------------------------------------------------------
def g1_g1d(y, gam1, probs, vars_):
    sigma = 1.0 / gam1
    if -1e-10 < sigma < 1e-10:
        return y.copy(), np.ones_like(y)
    probs, vars_ = np.asarray(probs), np.asarray(vars_)
    emax = vars_.max()
    e = np.exp(-0.5 * y[:, None] ** 2 * (emax - vars_) / (vars_ + sigma) / (emax + sigma))
    z = probs / np.sqrt(vars_ + sigma) * e
    pk = z.sum(1)
    zd = z / (vars_ + sigma) * y[:, None]
    pkd = -zd.sum(1)
    pkdd = (-probs / (vars_ + sigma) ** 1.5 * e + zd / (vars_ + sigma) * y[:, None]).sum(1)
    return y + sigma * pkd / pk, 1.0 + sigma * (pkdd / pk - (pkd / pk) ** 2)


def update_prior(r1, gam1, probs, vars_, Mt, EM_max_iter=2, EM_err_thr=1e-2, learn_vars=1):
    probs, vars_ = list(probs), list(vars_)
    noise_var = 1.0 / gam1
    lam = 1.0 - probs[0]
    omegas = [probs[0]] + [p / lam for p in probs[1:]]
    for _ in range(EM_max_iter):
        vmax = max(vars_)
        pp, vp = list(probs), list(vars_)
        v = np.array(vars_[1:])
        om = np.array(omegas[1:])
        num = lam * om * np.exp(-r1[:, None] ** 2 / 2 * (vmax - v) / (v + noise_var) / (vmax + noise_var)) / np.sqrt(v + noise_var) / np.sqrt(2 * np.pi)
        gam = gam1 * r1[:, None] / (1.0 / v + gam1)
        s = num.sum(1)
        beta = num / s[:, None]
        pin = 1.0 / (1.0 + (1.0 - lam) / np.sqrt(2 * np.pi * noise_var) * np.exp(-r1 ** 2 / 2 * vmax / noise_var / (noise_var + vmax)) / s)
        vv = 1.0 / (1.0 / v + gam1)
        tot = pin.sum()
        lam = tot / Mt
        gg = beta * (gam * gam + vv)
        for j in range(len(v)):
            res, resg = (beta[:, j] * pin).sum(), (gg[:, j] * pin).sum()
            if learn_vars == 1:
                vars_[j + 1] = resg / res
            omegas[j + 1] = res / tot
            probs[j + 1] = lam * omegas[j + 1]
        probs[0] = 1.0 - lam
        dp = np.sqrt(sum((p - q) ** 2 for p, q in zip(probs, pp)) / sum(p * p for p in probs))
        dv = np.sqrt(sum((p - q) ** 2 for p, q in zip(vars_, vp)) / sum(p * p for p in vars_))
        if dp < EM_err_thr and dv < EM_err_thr:
            break
    j = 0
    while j < len(vars_):                               # merge (vamp.cpp:1054-1071)
        k = j + 1
        while k < len(vars_):
            denom = min(vars_[j], vars_[k]) if vars_[j] != 0 else 1e-7
            if abs(vars_[j] - vars_[k]) / denom < 0.5:
                probs[j] += probs[k]
                del vars_[k], probs[k]
            else:
                k += 1
        j += 1
    return probs, vars_


# ---- the solvers (vamp.cpp:1074-1229, denoiserXXT.cpp:15-130) -------------------------------------------------------------------
def precond_cg(D, v, mu_start, tau, gam2, denoiser, CG_max_iter):
    N = D.N
    Q = lambda u: tau * D.ATx(D.Ax(u)) + gam2 * u       # noqa: E731  lmmse_mult, vamp.cpp:1074-1118
    diag = tau * (N - 1) / N + gam2
    mu = mu_start.copy()
    r = v - (Q(mu) if np.any(mu != 0) else 0.0)
    z = r / diag
    p = z.copy()
    prev_ons, steps = 0.0, 0
    for _ in range(CG_max_iter):
        d = Q(p)
        alpha = (r @ z) / (d @ p)
        mu = mu + alpha * p
        steps += 1
        if denoiser == 0:
            ons = gam2 * (v @ mu)
            relerr = abs((ons - prev_ons) / ons) if ons != 0 else 1.0
            if relerr < 1e-8:
                break
            prev_ons = ons
        beta = 1.0 / (r @ z)
        r = r - alpha * d
        z = r / diag
        beta *= r @ z
        p = z + beta * p
        if np.sqrt(r @ r) / np.sqrt(v @ v) < 1e-5:
            break
    return mu, steps


def people_stats(D):
    """data.cpp:558-716"""
    value = D.G                                          # (a - mave) msig b na, per (individual, marker)
    bna = (D.b * D.present.astype(float)).T
    s1, s2, cnt = value.sum(1), (value ** 2).sum(1), bna.sum(1)
    mave_p, msig_p = np.zeros(D.N), np.zeros(D.N)
    ok = D.present
    mave_p[ok] = s1[ok] / cnt[ok]
    msig_p[ok] = np.sqrt((cnt[ok] - 1) / (s2[ok] - cnt[ok] * mave_p[ok] ** 2))
    return mave_p, msig_p, cnt


def cg_aat(D, v, mu_start, tau, gam2, CG_max_iter, ppl):
    """denoiserXXT.cpp:52-130"""
    mave_p, msig_p, numb_p = ppl
    N = D.N
    Q = lambda u: tau * D.Ax(D.ATx(u)) + gam2 * u       # noqa: E731  lmmse_multAAT, :15-35
    with np.errstate(divide="ignore", invalid="ignore"):
        diag = tau * ((numb_p - 1) / msig_p / msig_p + mave_p * mave_p * numb_p) / N + gam2
    mu = mu_start.copy()
    r = v - (Q(mu) if np.any(mu != 0) else 0.0)
    z = r / diag
    p = z.copy()
    steps = 0
    for _ in range(CG_max_iter):
        d = Q(p)
        alpha = (r @ z) / (d @ p)
        mu = mu + alpha * p
        steps += 1
        beta = 1.0 / (r @ z)
        r = r - alpha * d
        z = r / diag
        beta *= r @ z
        p = z + beta * p
        if np.sqrt((r @ r) / (v @ v)) < 1e-4:
            break
    return mu, steps


# ---- infere_bin_class, vamp_probit.cpp:20-658 -----------------------------------------------------------------------------------
def probit_run(D, y, probs, vars_, *, iterations, gam1, rho, CG_max_iter, seed, probit_var=1.0):
    N, M, Mt = D.N, D.M, D.M
    vars_ = [v * N for v in vars_]                      # vamp.cpp:154-155
    probs = list(probs)
    tau1 = gam1                                         # :38
    p1 = simulate(N, [1.0], [1.0], 1)                   # :52 (utilities.hpp:23: default seed 1)
    r1, r2, x1 = np.zeros(M), np.zeros(M), np.zeros(M)
    alpha1, gam2 = 0.0, 0.0
    u = bern_probe(seed, 0, M, Mt)
    out = []

    def gbin(p, t1):                                    # :661-705
        c = p / np.sqrt(probit_var + 1.0 / t1)
        ratio = 2.0 / np.sqrt(2 * np.pi) / special.erfcx(-(2 * y - 1) * c / np.sqrt(2))
        g = p + (2 * y - 1) * ratio / t1 / np.sqrt(probit_var + 1.0 / t1)
        gd = 1 - ratio / (1 + t1 * probit_var) * ((2 * y - 1) * c + ratio)
        return g, gd

    for it in range(1, iterations + 1):
        x1_prev, alpha1_prev = x1.copy(), alpha1
        for it_revar in range(1, 51):                   # auto_var_max_iter = 50 (:164)
            x1, d = g1_g1d(r1, gam1, probs, vars_)
            alpha1 = d.sum() / Mt
            eta1 = gam1 / alpha1
            if it <= 1:
                break
            g_prev = gam1
            gam1 = clip(1.0 / (1.0 / eta1 + ((x1 - r1) ** 2).sum() / Mt))
            probs, vars_ = update_prior(r1, gam1, probs, vars_, Mt)
            if abs(gam1 - g_prev) < 1e-3:
                break
        if it > 1:                                      # :208-214 (rho_it2 = rho)
            x1 = rho * x1 + (1 - rho) * x1_prev
            alpha1 = rho * alpha1 + (1 - rho) * alpha1_prev
        gam2 = clip(eta1 - gam1)                        # :296
        r2 = (eta1 * x1 - gam1 * r1) / gam2             # :303-304
        z1, gd = gbin(p1, tau1)                         # one round (:325: auto_var_max_iter = 1)
        beta1 = gd.sum() / N
        zeta1 = tau1 / beta1
        if it > 1:
            tau1 = clip(1.0 / (1.0 / zeta1 + ((z1 - p1) ** 2).sum() / N))
        p2 = (z1 - beta1 * p1) / (1 - beta1)            # :418-419
        tau2 = tau1 * (1 - beta1) / beta1               # :427
        v = tau2 * D.ATx(p2) + gam2 * r2                # :493-495
        x2, cg_steps = precond_cg(D, v, np.zeros(M), tau2, gam2, 1, CG_max_iter)    # :497: every solve from zero
        invq, ons_steps = precond_cg(D, u, np.zeros(M), tau2, gam2, 0, CG_max_iter)
        alpha2 = gam2 * (u @ invq)                      # g2d_onsager, vamp.cpp:871-889
        eta2 = gam2 / alpha2
        gam2_used = gam2
        if it > 1:                                      # :528-529
            gam2 = clip(1.0 / (1.0 / eta2 + ((x2 - r2) ** 2).sum() / Mt))
        r1 = (x2 - alpha2 * r2) / (1 - alpha2)          # :534-535 with rho_it = 1
        gam1 = gam2 * (1 - alpha2) / alpha2             # :545-546
        z2 = D.Ax(x2)
        beta2 = Mt / N * (1 - alpha2)                   # :556
        zeta2 = tau2 / beta2
        if it > 1:
            tau2 = 1.0 / (1.0 / zeta2 + ((z2 - p2) ** 2).sum() / N)
        p1 = (z2 - beta2 * p2) / (1 - beta2)            # :575-576
        tau1 = tau2 * (1 - beta2) / beta2               # :586-587
        out.append(dict(x1=x1.copy(), x2=x2.copy(), r1=r1.copy(), alpha1=alpha1, eta1=eta1, gam2=gam2_used, alpha2=alpha2,
                        beta1=beta1, tau1_next=tau1, gam1_next=gam1, cg=cg_steps, ons=ons_steps, L=len(probs)))
    return out


def test_probit_iterations_against_the_dense_restatement(oracle):
    N, M = 200, 40
    rng = np.random.default_rng(3)
    bed = synth.synth_bed(N, M, seed=5, miss_ppm=20000)
    D = Dense(bed, N, M)
    beta = rng.standard_normal(M) * (rng.random(M) < 0.3) * 0.4
    y = ((D.Ax(beta * np.sqrt(N)) + 0.5 * rng.standard_normal(N)) > 0).astype(float)
    probs, vars_ = [0.7, 0.2, 0.1], [0.0, 0.01, 0.1]
    kw = dict(iterations=3, gam1=1e-2, rho=0.5, CG_max_iter=30, seed=4)
    mine = probit_run(D, y, probs, vars_, **kw)
    ref = oracle.infere(bed, N, M, y, probs, vars_, model="bin_class", gamw=1.0, stop_criteria_thr=1e-12, **kw)
    assert ref.niter == 3
    for it in range(3):
        m, t = mine[it], ref.trace[it]
        assert (m["cg"], m["ons"], m["L"]) == (t["cg_iters"], t["onsager_iters"], t["L_after"]), (it, m, t)
        for k, f in (("alpha1", "alpha1"), ("eta1", "eta1"), ("gam2", "gam2"), ("alpha2", "alpha2"), ("beta1", "beta1"),
                     ("tau1_next", "tau1_next"), ("gam1_next", "gam1_next")):
            assert np.isclose(m[k], t[f], rtol=1e-8), (it, k, m[k], t[f])
        assert rel(m["x1"], ref.x1[it] * np.sqrt(N)) < 1e-9 and rel(m["x2"], ref.x2[it] * np.sqrt(N)) < 1e-9, it
        if it + 1 < 3:          # the stored r1 of an iteration is the one it STARTED from (vamp_probit.cpp:232-238)
            assert rel(m["r1"], ref.r1[it + 1] * np.sqrt(N)) < 1e-8, it


# ---- infere_linear with --use-XXT-denoiser 1, vamp.cpp:261-760 + denoiserXXT.cpp ------------------------------------------------
def xxt_run(D, y, probs, vars_, *, iterations, gam1, gamw, rho, CG_max_iter, seed):
    N, M, Mt = D.N, D.M, D.M
    vars_ = [v * N for v in vars_]
    probs = list(probs)
    ppl = people_stats(D)                                # vamp.cpp:169-170
    r1, r2, x1 = np.zeros(M), np.zeros(M), np.zeros(M)
    alpha1 = alpha2 = 0.0
    gam2 = 0.0
    mu_last = np.zeros(N)
    u = bern_probe(seed, 0, M, Mt)
    out = []
    for it in range(1, iterations + 1):
        x1_prev, alpha1_prev = x1.copy(), alpha1
        rounds = 0
        for it_revar in range(1, 6):                     # auto_var_max_iter = 5 (vamp.hpp:36)
            rounds += 1
            x1, d = g1_g1d(r1, gam1, probs, vars_)
            alpha1 = d.sum() / Mt
            eta1 = gam1 / alpha1
            if it <= 1:
                break
            g_prev = gam1
            gam1 = clip(1.0 / (1.0 / eta1 + ((x1 - r1) ** 2).sum() / Mt))
            probs, vars_ = update_prior(r1, gam1, probs, vars_, Mt)
            if abs(gam1 - g_prev) < 1e-3:
                break
        if it > 1:
            x1 = rho * x1 + (1 - rho) * x1_prev
            alpha1 = rho * alpha1 + (1 - rho) * alpha1_prev
        gam2 = clip(eta1 - gam1)                         # :472
        r2 = (eta1 * x1 - gam1 * r1) / gam2              # :486
        rho = max(rho, min(2 * min(alpha1, alpha2), 1.0))    # :501-502
        if it <= 1:
            probs, vars_ = update_prior(r1, gam1, probs, vars_, Mt)      # :518-519
        # lmmse_denoiserAAT (denoiserXXT.cpp:37-50)
        v = y - D.Ax(r2)
        uu, cg_steps = cg_aat(D, v, mu_last, gamw, gam2, CG_max_iter, ppl)
        mu_last = uu                                     # save == 1
        x2 = gamw * D.ATx(uu) + r2
        invq, ons_steps = precond_cg(D, u, np.zeros(M), gamw, gam2, 0, CG_max_iter)      # :631
        alpha2 = gam2 * (u @ invq)
        eta2 = gam2 / alpha2
        if it > 2:                                       # :691-693
            gam2 = clip(1.0 / (1.0 / eta2 + ((x2 - r2) ** 2).sum() / Mt))
        gam1 = clip(eta2 - gam2)                         # :702
        r1 = (eta2 * x2 - gam2 * r2) / gam1              # :707
        temp = D.Ax(x2) - y                              # updateNoisePrec, :892-927
        gamw = N / (temp @ temp + Mt * (u @ D.ATx(D.Ax(invq))))
        out.append(dict(x1=x1.copy(), x2=x2.copy(), alpha2=alpha2, gamw=gamw, gam1_next=gam1, cg=cg_steps, ons=ons_steps,
                        L=len(probs), rounds=rounds))
    return out


def test_xxt_denoiser_iterations_against_the_dense_restatement(oracle):
    N, M = 200, 44
    rng = np.random.default_rng(8)
    bed = synth.synth_bed(N, M, seed=6, miss_ppm=30000)
    D = Dense(bed, N, M)
    beta = rng.standard_normal(M) * (rng.random(M) < 0.3) * 0.3
    y = D.Ax(beta * np.sqrt(N)) + 0.7 * rng.standard_normal(N)
    # the oracle's people statistics first (data.cpp:558-716)
    o = oracle.people_stats(bed, N, M)
    mine = people_stats(D)
    for a, b in zip(mine, o):
        assert np.allclose(a, b[:N], rtol=1e-11, atol=1e-13)
    probs, vars_ = [0.7, 0.2, 0.1], [0.0, 0.01, 0.1]
    kw = dict(iterations=3, gam1=1e-6, gamw=2.0, rho=0.5, CG_max_iter=40, seed=4)
    got = xxt_run(D, y, probs, vars_, **kw)
    ref = oracle.infere(bed, N, M, y, probs, vars_, use_XXT_denoiser=1, stop_criteria_thr=1e-12, **kw)
    assert ref.niter == 3
    for it in range(3):
        m, t = got[it], ref.trace[it]
        assert (m["cg"], m["ons"], m["L"], m["rounds"]) == (t["cg_iters"], t["onsager_iters"], t["L_after"], t["revar_rounds"]), (it, m, t)
        assert np.isclose(m["alpha2"], t["alpha2"], rtol=1e-8) and np.isclose(m["gamw"], t["gamw"], rtol=1e-8), it
        assert np.isclose(m["gam1_next"], t["gam1_next"], rtol=1e-7), it
        assert rel(m["x1"], ref.x1[it] * np.sqrt(N)) < 1e-8 and rel(m["x2"], ref.x2[it] * np.sqrt(N)) < 1e-8, it


# ---- p-values, data.cpp:1108-1226 / :1235-1353, utilities.cpp:321-334 -----------------------------------------------------------
def reg_pval(sumx, sumsqx, sumxy, sumy, sumsqy, n):
    s2y = (sumsqy - sumy * sumy / n) / (n - 1)
    s2x = (sumsqx - sumx * sumx / n) / (n - 1)
    sxy = (sumxy - sumx * sumy / n) / (n - 1)
    rxy = sxy / np.sqrt(s2x * s2y)
    t = rxy * np.sqrt((n - 2) / (1 - rxy * rxy))
    return 2.0 * stats.t.sf(abs(t), n - 2)              # boost: 2 cdf(complement(students_t(n - 2), |t|))


def pvals_dense(D, z1, y, x1_hat, chrom=None):
    N, M = D.N, D.M
    na = D.present.astype(float)
    ymod = y - z1
    out = np.zeros(M)
    groups = [None] if chrom is None else range(1, 24)
    for ch in groups:
        if ch is None:
            ychrom = None
        else:
            sel = chrom == ch
            ychrom = (D.G[:, sel] / np.sqrt(N)) @ x1_hat[sel] + ymod          # :1256-1284
        for m in range(M):
            if ch is not None and chrom[m] != ch:
                continue
            value = D.G[:, m]
            w = D.b[m] * na
            ymark = ymod + value / np.sqrt(N) * x1_hat[m] if ch is None else ychrom   # :1145-1148
            out[m] = reg_pval(value.sum(), (value ** 2).sum(), (value * ymark).sum(), (ymark * w).sum(), (ymark ** 2 * w).sum(),
                              int(round(w.sum())))
    return out


@pytest.mark.parametrize("fna", [0.0, 0.04])
def test_pvalues_loo_and_loco_against_the_dense_restatement(oracle, fna):
    N, M = 204, 46
    rng = np.random.default_rng(12)
    bed = synth.synth_bed(N, M, seed=7, miss_ppm=30000)
    present = rng.random(N) >= fna
    D = Dense(bed, N, M, present)
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(present.sum())
    x1 = rng.standard_normal(M) * 0.2 * np.sqrt(N)
    y = (D.Ax(x1) + rng.standard_normal(N)) * present
    z1 = D.Ax(x1)
    chrom = np.sort(rng.integers(1, 6, M)).astype(np.int32)
    chrom[-3:] = 23
    npad = 4 * ((N + 3) // 4)
    zp, yp = np.zeros(npad), np.zeros(npad)
    zp[:N], yp[:N] = z1, y
    loo = oracle.pvals(bed, N, M, zp, yp, x1, mask4=m4, nonas=nonas)
    loco = oracle.pvals(bed, N, M, zp, yp, x1, chrom=chrom, mask4=m4, nonas=nonas)
    assert np.allclose(loo, pvals_dense(D, z1, y, x1), rtol=1e-8, atol=1e-300)
    assert np.allclose(loco, pvals_dense(D, z1, y, x1, chrom=chrom), rtol=1e-8, atol=1e-300)
    assert 0 < loo.min() and loo.max() <= 1
