"""ctypes binding of oracle/libgvoracle.so -- the CPU restatement of the reference's hot path.

TEST INFRASTRUCTURE ONLY (see oracle/gv_oracle.hpp).  Imported by tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never by gvamp_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TRACE_FIELDS = ("gam1_denoise", "alpha1", "eta1", "gam2", "alpha2", "eta2", "gam2_reest", "gam1_next",
                "gamw", "rho", "cg_iters", "onsager_iters", "revar_rounds", "L_after", "n_ax", "n_atx")

ALLREDUCE_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_long, C.c_void_p)


class Params(C.Structure):
    _fields_ = [("N", C.c_int), ("Mt", C.c_int), ("nshards", C.c_int), ("shard_rank", C.c_int),
                ("iterations", C.c_int), ("CG_max_iter", C.c_int), ("EM_max_iter", C.c_int),
                ("EM_err_thr", C.c_double), ("stop_criteria_thr", C.c_double), ("rho", C.c_double),
                ("learn_vars", C.c_int), ("seed", C.c_ulong), ("use_lmmse_damp", C.c_int),
                ("gam1", C.c_double), ("gamw", C.c_double), ("L", C.c_int),
                ("probs", C.POINTER(C.c_double)), ("vars", C.POINTER(C.c_double)),
                ("true_signal", C.POINTER(C.c_double)), ("out_prefix", C.c_char_p),
                ("verbose", C.c_int), ("nthreads", C.c_int), ("alpha_scale", C.c_double),
                ("phen_mode", C.c_int), ("is_na", C.POINTER(C.c_ubyte)),
                ("cb", ALLREDUCE_CB), ("cb_user", C.c_void_p), ("use_XXT_denoiser", C.c_int),
                ("r1_init", C.POINTER(C.c_double)), ("x_init", C.POINTER(C.c_double)),
                ("bin_class", C.c_int), ("probit_var", C.c_double), ("C", C.c_int), ("covs", C.POINTER(C.c_double)),
                ("freeze_ind", C.POINTER(C.c_double))]


_SO = "libgvoracle.so"
TIMING_FLAGS = None      # set by use_timing_build(): the compiler flags of the library in use, None = the parity build


def build(force=False):
    so = os.path.join(_HERE, _SO)
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", _HERE, _SO])
    return so


def use_timing_build():
    """bench.py's cpu_baseline leg only: the same sources compiled -O3 -march=native ON THIS HOST into
    libgvoracle_timing.so, kept apart from the parity library (whose -O2 -mavx2 -mfma flags are chosen for bit-reproducibility
    and portability, oracle/Makefile).  Returns the flags in use; falls back to the parity build when there is no compiler."""
    global _SO, _LIB, TIMING_FLAGS
    try:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgvoracle_timing.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        _SO, _LIB = "libgvoracle_timing.so", None
        TIMING_FLAGS = "-O3 -march=native -fopenmp (oracle/Makefile: timing; built on this host)"
    except (OSError, subprocess.CalledProcessError):
        _SO, _LIB = "libgvoracle.so", None
        TIMING_FLAGS = "-O2 -mavx2 -mfma -fopenmp (parity build: the timing build did not compile on this host)"
    lib()
    return TIMING_FLAGS


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        dp, up = C.POINTER(C.c_double), C.POINTER(C.c_ubyte)
        L.gvo_infere.restype = C.c_void_p
        L.gvo_infere.argtypes = [C.POINTER(Params), up, dp]
        for name in ("gvo_run_x_est", "gvo_run_mave", "gvo_run_msig", "gvo_run_probs", "gvo_run_vars"):
            getattr(L, name).restype = dp
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("gvo_run_x1", "gvo_run_x2", "gvo_run_r1"):
            getattr(L, name).restype = dp
            getattr(L, name).argtypes = [C.c_void_p, C.c_int]
        for name in ("gvo_run_niter", "gvo_run_L", "gvo_run_nsolves"):
            getattr(L, name).restype = C.c_int
            getattr(L, name).argtypes = [C.c_void_p]
        L.gvo_run_free.argtypes = [C.c_void_p]
        L.gvo_run_trace.argtypes = [C.c_void_p, C.c_int, dp]
        L.gvo_run_seconds.restype = C.c_double
        L.gvo_run_seconds.argtypes = [C.c_void_p, C.c_int]
        L.gvo_run_relres.restype = C.c_int
        L.gvo_run_relres.argtypes = [C.c_void_p, C.c_int, dp, C.c_int]
        L.gvo_run_R2trains.restype = C.c_int
        L.gvo_run_R2trains.argtypes = [C.c_void_p, dp, C.c_int]
        L.gvo_marker_stats.argtypes = [up, C.c_int, C.c_int, up, C.c_int, C.c_double, C.c_int, dp, dp]
        L.gvo_ax.argtypes = [up, C.c_int, C.c_int, up, dp, dp, dp, C.c_int, dp]
        L.gvo_atx.argtypes = [up, C.c_int, C.c_int, dp, dp, dp, C.c_int, dp]
        L.gvo_g1_g1d.argtypes = [dp, C.c_long, C.c_double, dp, dp, C.c_int, dp, dp]
        L.gvo_update_prior.restype = C.c_int
        L.gvo_update_prior.argtypes = [dp, C.c_int, C.c_int, C.c_double, dp, dp, C.c_int, C.c_int, C.c_double, C.c_int]
        L.gvo_cg_solve.restype = C.c_int
        L.gvo_cg_solve.argtypes = [up, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, C.c_int, C.c_int,
                                   C.c_int, dp, dp]
        L.gvo_pvals.argtypes = [up, C.c_int, C.c_int, up, C.c_int, dp, dp, dp, C.POINTER(C.c_int), C.c_int, dp]
        L.gvo_people_stats.argtypes = [up, C.c_int, C.c_int, up, C.c_int, dp, dp, dp]
        L.gvo_run_trace_probit.argtypes = [C.c_void_p, C.c_int, dp]
        L.gvo_probit_g.argtypes = [dp, dp, C.c_long, C.c_double, C.c_double, dp, dp]
        L.gvo_probit_g_cov.argtypes = [dp, dp, dp, C.c_long, C.c_double, C.c_double, dp, dp]
        L.gvo_newton_cov.argtypes = [C.c_int, C.c_int, dp, dp, dp, C.c_double, dp, dp, C.POINTER(C.c_double), dp]
        L.gvo_lu_solve.restype = C.c_int
        L.gvo_lu_solve.argtypes = [dp, dp, C.c_int]
        L.gvo_run_cov_eff.restype = C.c_int
        L.gvo_run_cov_eff.argtypes = [C.c_void_p, dp, C.c_int]
        L.gvo_erfcx.restype = C.c_double
        L.gvo_erfcx.argtypes = [C.c_double]
        L.gvo_student_t_two_sided.restype = C.c_double
        L.gvo_student_t_two_sided.argtypes = [C.c_double, C.c_double]
        L.gvo_sim_phen.argtypes = [up, C.c_int, C.c_int, C.c_double, C.c_int, C.c_ulong, C.c_int, dp, dp]
        L.gvo_bern_vec.argtypes = [C.c_ulong, C.c_int, C.c_int, C.c_int, dp]
        L.gvo_divide_work.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.gvo_mbytes.restype = C.c_int
        L.gvo_mbytes.argtypes = [C.c_int]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def mbytes(N):
    return (N + 3) // 4


def divide_work(Mt, nranks, rank):
    M, S = C.c_int(), C.c_int()
    lib().gvo_divide_work(Mt, nranks, rank, C.byref(M), C.byref(S))
    return M.value, S.value


def marker_stats(bed, N, M, mask4=None, nonas=None, alpha_scale=1.0, nthreads=1):
    bed = _u8(bed)
    mave, msig = np.empty(M), np.empty(M)
    m4 = _u8(mask4) if mask4 is not None else None
    lib().gvo_marker_stats(_up(bed), N, M, _up(m4) if m4 is not None else None,
                           N if nonas is None else nonas, alpha_scale, nthreads, _dp(mave), _dp(msig))
    return mave, msig


def ax(bed, N, M, mave, msig, x, mask4=None, nthreads=1):
    bed, mave, msig, x = _u8(bed), _f64(mave), _f64(msig), _f64(x)
    out = np.empty(4 * mbytes(N))
    m4 = _u8(mask4) if mask4 is not None else None
    lib().gvo_ax(_up(bed), N, M, _up(m4) if m4 is not None else None, _dp(mave), _dp(msig), _dp(x), nthreads, _dp(out))
    return out


def atx(bed, N, M, mave, msig, p, nthreads=1, mask4=None):
    """data::ATx (data.cpp:810-835).  dot_product (data.cpp:728-801) applies NO mask: the reference's callers hand in
    filter_pheno()'d vectors (zero at NA individuals).  mask4 != None restates that caller-side step here -- p is filtered
    with the phenotype mask first -- so that an unfiltered p (DBL_MAX at NA slots, data.cpp:147) has a defined expectation."""
    bed, mave, msig, p = _u8(bed), _f64(mave), _f64(msig), _f64(p)
    assert p.size == 4 * mbytes(N)
    if mask4 is not None:
        m4 = _u8(mask4)
        present = ((m4[np.arange(p.size) >> 2] >> (np.arange(p.size) & 3)) & 1).astype(bool)
        present &= np.arange(p.size) < N
        p = np.where(present, p, 0.0)
    out = np.empty(M)
    lib().gvo_atx(_up(bed), N, M, _dp(mave), _dp(msig), _dp(p), nthreads, _dp(out))
    return out


def g1_g1d(r, gam1, probs, vars_scaled):
    r, probs, vs = _f64(r), _f64(probs), _f64(vars_scaled)
    g1, g1d = np.empty(r.size), np.empty(r.size)
    lib().gvo_g1_g1d(_dp(r), r.size, gam1, _dp(probs), _dp(vs), probs.size, _dp(g1), _dp(g1d))
    return g1, g1d


def update_prior(r1, Mt, gam1, probs, vars_scaled, EM_max_iter=2, EM_err_thr=1e-2, learn_vars=1):
    r1 = _f64(r1)
    p, v = _f64(probs).copy(), _f64(vars_scaled).copy()
    L = lib().gvo_update_prior(_dp(r1), r1.size, Mt, gam1, _dp(p), _dp(v), p.size, EM_max_iter, EM_err_thr, learn_vars)
    return p[:L].copy(), v[:L].copy()


def cg_solve(bed, N, M, v, mu_start, tau, gam2, denoiser, CG_max_iter, nthreads=1):
    bed, v = _u8(bed), _f64(v)
    ms = _f64(mu_start) if mu_start is not None else None
    mu, rr = np.empty(M), np.empty(CG_max_iter)
    n = lib().gvo_cg_solve(_up(bed), N, M, _dp(v), _dp(ms) if ms is not None else None, tau, gam2, denoiser,
                           CG_max_iter, nthreads, _dp(mu), _dp(rr))
    return mu, rr[:n].copy()


def pvals(bed, N, M, z1, y, x1_hat, chrom=None, mask4=None, nonas=None, nthreads=1):
    """data::pvals_calc (chrom None) / pvals_calc_LOCO; z1, y length >= N (y already filtered), x1_hat length M."""
    bed, z1, y, x1_hat = _u8(bed), _f64(z1), _f64(y), _f64(x1_hat)
    out = np.empty(M)
    m4 = _u8(mask4) if mask4 is not None else None
    ch = np.ascontiguousarray(chrom, dtype=np.int32) if chrom is not None else None
    lib().gvo_pvals(_up(bed), N, M, _up(m4) if m4 is not None else None, N if nonas is None else nonas, _dp(z1), _dp(y),
                    _dp(x1_hat), ch.ctypes.data_as(C.POINTER(C.c_int)) if ch is not None else None, nthreads, _dp(out))
    return out


def probit_g(p, y, tau1, probit_var=1.0):
    """vamp::g1_bin_class / g1d_bin_class (vamp_probit.cpp:661-705) element-wise, m_cov = 0."""
    p, y = _f64(p), _f64(y)
    g, gd = np.empty(p.size), np.empty(p.size)
    lib().gvo_probit_g(_dp(p), _dp(y), p.size, tau1, probit_var, _dp(g), _dp(gd))
    return g, gd


def probit_g_cov(p, y, m_cov, tau1, probit_var=1.0):
    """g1_bin_class / g1d_bin_class with the covariate offset m_cov[i] = <Z[i], cov_eff> (vamp_probit.cpp:347,:364)."""
    p, y, m = _f64(p), _f64(y), _f64(m_cov)
    g, gd = np.empty(p.size), np.empty(p.size)
    lib().gvo_probit_g_cov(_dp(p), _dp(y), _dp(m), p.size, tau1, probit_var, _dp(g), _dp(gd))
    return g, gd


def newton_cov(covs, y, gg=None, probit_var=1.0, eta0=None):
    """vamp::Newton_method_cov (vamp_probit.cpp:936-1062): (eta, mlogL_probit(eta), grad_cov(eta))."""
    covs = np.ascontiguousarray(covs, dtype=np.float64)
    N, Cn = covs.shape
    y = _f64(y)
    gg = _f64(gg) if gg is not None else np.zeros(N)
    e0 = _f64(eta0) if eta0 is not None else np.zeros(Cn)
    out, grad, ml = np.empty(Cn), np.empty(Cn), C.c_double()
    lib().gvo_newton_cov(N, Cn, _dp(covs), _dp(y), _dp(gg), probit_var, _dp(e0), _dp(out), C.byref(ml), _dp(grad))
    return out, ml.value, grad


def lu_solve(A, b):
    A = np.ascontiguousarray(A, dtype=np.float64)
    x = np.array(b, dtype=np.float64)
    rc = lib().gvo_lu_solve(_dp(A), _dp(x), A.shape[0])
    return x if rc == 0 else None


def erfcx(x):
    return lib().gvo_erfcx(float(x))


def people_stats(bed, N, M, mask4=None, nonas=None):
    """data::compute_people_statistics: (mave_people, msig_people, numb_people), each 4*ceil(N/4) long."""
    bed = _u8(bed)
    n4 = 4 * mbytes(N)
    a, b, c = np.empty(n4), np.empty(n4), np.empty(n4)
    m4 = _u8(mask4) if mask4 is not None else None
    lib().gvo_people_stats(_up(bed), N, M, _up(m4) if m4 is not None else None, N if nonas is None else nonas,
                           _dp(a), _dp(b), _dp(c))
    return a, b, c


def student_t_two_sided(t, nu):
    return lib().gvo_student_t_two_sided(t, nu)


def sim_phen(bed, N, Mt, h2, CV, seed, nthreads=1):
    bed = _u8(bed)
    beta, y = np.empty(Mt), np.empty(N)
    lib().gvo_sim_phen(_up(bed), N, Mt, h2, CV, seed, nthreads, _dp(beta), _dp(y))
    return beta, y


def bern_vec(seed, S, M, Mt):
    out = np.empty(M)
    lib().gvo_bern_vec(seed, S, M, Mt, _dp(out))
    return out


class Run:
    """Result of one vamp::infere() restatement run."""

    def __init__(self, h, Mt):
        L = lib()
        n = L.gvo_run_niter(h)
        self.niter = n
        self.x_est = np.ctypeslib.as_array(L.gvo_run_x_est(h), (Mt,)).copy()
        self.mave = np.ctypeslib.as_array(L.gvo_run_mave(h), (Mt,)).copy()
        self.msig = np.ctypeslib.as_array(L.gvo_run_msig(h), (Mt,)).copy()
        self.x1 = [np.ctypeslib.as_array(L.gvo_run_x1(h, i), (Mt,)).copy() for i in range(n)]
        self.x2 = [np.ctypeslib.as_array(L.gvo_run_x2(h, i), (Mt,)).copy() for i in range(n)]
        self.r1 = [np.ctypeslib.as_array(L.gvo_run_r1(h, i), (Mt,)).copy() for i in range(n)]
        self.trace = []
        buf = np.empty(16)
        for i in range(n):
            L.gvo_run_trace(h, i, _dp(buf))
            t = dict(zip(TRACE_FIELDS, buf.tolist()))
            t["seconds"] = L.gvo_run_seconds(h, i)
            b3 = np.empty(3)
            L.gvo_run_trace_probit(h, i, _dp(b3))
            t["beta1"], t["tau2"], t["tau1_next"] = b3.tolist()
            self.trace.append(t)
        nl = L.gvo_run_L(h)
        self.probs = np.ctypeslib.as_array(L.gvo_run_probs(h), (nl,)).copy()
        self.vars = np.ctypeslib.as_array(L.gvo_run_vars(h), (nl,)).copy()
        self.relres = []
        rb = np.empty(4096)
        for s in range(L.gvo_run_nsolves(h)):
            k = L.gvo_run_relres(h, s, _dp(rb), rb.size)
            self.relres.append(rb[:k].copy())
        k = L.gvo_run_R2trains(h, _dp(rb), rb.size)
        self.R2trains = rb[:k].copy()
        k = L.gvo_run_cov_eff(h, _dp(rb), rb.size)
        self.cov_eff = rb[:k].copy()
        L.gvo_run_free(h)


def infere(bed_full, N, Mt, y, probs, vars_, *, nshards=1, shard_rank=-1, iterations=1, CG_max_iter=60,
           EM_max_iter=2, EM_err_thr=1e-2, stop_criteria_thr=1e-4, rho=0.15, learn_vars=1, seed=1,
           use_lmmse_damp=0, gam1=1e-8, gamw=2.0, true_signal=None, out_prefix=None, verbose=0, nthreads=1,
           alpha_scale=1.0, is_na=None, allreduce=None, use_XXT_denoiser=0, r1_init=None, x_init=None,
           model="linear", probit_var=1.0, covs=None, freeze_ind=None):
    """vamp::infere (linear) on `nshards` marker shards.  `allreduce(np_array)` is an in-place SUM callback
    used when shard_rank >= 0 (one shard per process, e.g. torch.distributed gloo)."""
    bed_full, y = _u8(bed_full), _f64(y)
    p = Params()
    p.N, p.Mt, p.nshards, p.shard_rank = N, Mt, nshards, shard_rank
    p.iterations, p.CG_max_iter, p.EM_max_iter = iterations, CG_max_iter, EM_max_iter
    p.EM_err_thr, p.stop_criteria_thr, p.rho = EM_err_thr, stop_criteria_thr, rho
    p.learn_vars, p.seed, p.use_lmmse_damp = learn_vars, seed, use_lmmse_damp
    p.gam1, p.gamw = gam1, gamw
    keep = []
    if probs is not None and len(probs):
        pr, vr = _f64(probs), _f64(vars_)
        keep += [pr, vr]
        p.L, p.probs, p.vars = pr.size, _dp(pr), _dp(vr)
    else:
        p.L = 0
    if true_signal is not None:
        ts = _f64(true_signal)
        keep.append(ts)
        p.true_signal = _dp(ts)
    p.out_prefix = out_prefix.encode() if out_prefix else None
    p.verbose, p.nthreads, p.alpha_scale = verbose, nthreads, alpha_scale
    p.use_XXT_denoiser = use_XXT_denoiser
    p.bin_class, p.probit_var = int(model == "bin_class"), probit_var
    if covs is not None:
        cz = np.ascontiguousarray(covs, dtype=np.float64)
        keep.append(cz)
        p.C, p.covs = cz.shape[1], _dp(cz)
    if freeze_ind is not None:
        fz = _f64(freeze_ind)
        keep.append(fz)
        p.freeze_ind = _dp(fz)
    if r1_init is not None:
        ri = _f64(r1_init)
        keep.append(ri)
        p.r1_init = _dp(ri)
    if x_init is not None:
        xi = _f64(x_init)
        keep.append(xi)
        p.x_init = _dp(xi)
    if is_na is not None:
        na = _u8(is_na)
        keep.append(na)
        p.phen_mode, p.is_na = 1, _up(na)
    else:
        p.phen_mode = 0
    if allreduce is not None:
        def _cb(buf, n, _user):
            a = np.ctypeslib.as_array(buf, (n,))
            allreduce(a)
        cb = ALLREDUCE_CB(_cb)
        keep.append(cb)
        p.cb = cb
    h = lib().gvo_infere(C.byref(p), _up(bed_full), _dp(y))
    return Run(h, Mt)
