"""ctypes binding of libgvamp_host.so (include/gvamp_host.h): vamp::infere() of the host-side C++ mirror, run on a
shard already resident in a capi.Shard.  Plumbing only."""
import ctypes as C
import os

import numpy as np

from . import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgvamp_host.so")
_LIB = None

ITER_FIELDS = ("gam1_denoise", "alpha1", "eta1", "gam2", "alpha2", "eta2", "gam2_reest", "gam1_next", "gamw", "rho",
               "R2_denoise", "R2_lmmse")


class Opts(C.Structure):
    _fields_ = [("iterations", C.c_int), ("CG_max_iter", C.c_int), ("EM_max_iter", C.c_int),
                ("EM_err_thr", C.c_double), ("stop_criteria_thr", C.c_double), ("rho", C.c_double),
                ("learn_vars", C.c_int), ("seed", C.c_ulong), ("use_lmmse_damp", C.c_int),
                ("gam1", C.c_double), ("gamw", C.c_double), ("L", C.c_int),
                ("probs", C.POINTER(C.c_double)), ("vars", C.POINTER(C.c_double)), ("out_prefix", C.c_char_p),
                ("verbose", C.c_int), ("diagnostics", C.c_int), ("alpha_scale", C.c_double), ("use_XXT_denoiser", C.c_int),
                ("bin_class", C.c_int), ("probit_var", C.c_double), ("fuse_solves", C.c_int),
                ("C", C.c_int), ("covs", C.POINTER(C.c_double)), ("cov_eff_out", C.POINTER(C.c_double)),
                ("freeze_index_file", C.c_char_p), ("reanchor_every", C.c_int)]


class Iter(C.Structure):
    _fields_ = [(f, C.c_double) for f in ITER_FIELDS] + \
               [("cg_iters", C.c_int), ("onsager_iters", C.c_int), ("revar_rounds", C.c_int), ("L_after", C.c_int),
                ("n_ax", C.c_long), ("n_atx", C.c_long), ("n_ax_pass", C.c_long), ("n_atx_pass", C.c_long),
                ("beta1", C.c_double), ("tau2", C.c_double),
                ("tau1_next", C.c_double), ("seconds", C.c_double), ("seconds_io", C.c_double), ("probe_product", C.c_int)]


HOST_ABI_VERSION = 2     # GVH_ABI_VERSION of include/gvamp_host.h (gvh_opts / gvh_iter below)


def load():
    global _LIB
    if _LIB is None:
        capi.load()
        if not os.path.exists(LIB_PATH):
            raise capi.GvError("libgvamp_host.so is not built: run `make -C gvamp_amd/csrc/host`")
        L = C.CDLL(LIB_PATH)
        if L.gvh_abi_version() != HOST_ABI_VERSION:
            raise capi.GvError("libgvamp_host.so speaks ABI %d, this binding %d: rebuild (make -C gvamp_amd/csrc/host)" % (L.gvh_abi_version(), HOST_ABI_VERSION))
        dp, up = C.POINTER(C.c_double), C.POINTER(C.c_ubyte)
        L.gvh_sim_phen.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_ulong, dp, dp]
        L.gvh_infere_linear.argtypes = [C.c_void_p, C.POINTER(Opts), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp, up,
                                        C.c_int, dp, dp, C.POINTER(Iter), C.c_int, C.POINTER(C.c_int), dp, dp, dp, dp, dp,
                                        C.POINTER(C.c_int)]
        L.gvh_last_error.restype = C.c_char_p
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None


def sim_phen(shard, h2, CV, seed, rank=0):
    """sim.cpp:78-79,153,183-218 on a resident shard: (beta of this rank's markers, y)."""
    beta, y = np.empty(max(shard.M, 1)), np.empty(shard.N)
    if load().gvh_sim_phen(shard.h, shard.N, shard.M, shard.Mt, shard.S, rank, h2, CV, seed, _dp(beta), _dp(y)):
        raise capi.GvError("gvh_sim_phen failed: " + load().gvh_last_error().decode())
    return beta[:shard.M].copy(), y


class Result:
    pass


def infere_linear(shard, y, probs, vars_, *, iterations=1, CG_max_iter=60, EM_max_iter=2, EM_err_thr=1e-2,
                  stop_criteria_thr=1e-4, rho=0.15, learn_vars=1, seed=1, use_lmmse_damp=0, gam1=1e-8, gamw=2.0,
                  true_signal=None, out_prefix=None, verbose=0, diagnostics=0, alpha_scale=1.0, mask4=None,
                  nonas=None, history=True, rank=0, use_XXT_denoiser=0, model="linear", probit_var=1.0, fuse_solves=1,
                  covs=None, freeze_index_file=None, reanchor_every=-1):
    """vamp::infere on the resident shard.  fuse_solves defaults to 1 HERE -- the level whose products are bit-identical to the
    reference's own sequence, which is what most parity tests want to compare against; the drivers (gvamp_sim, gvamp_main_real,
    options.hpp), the vamp class and bench.py default to 4.  reanchor_every < 0 keeps the drivers' default (10)."""
    L = load()
    y = np.ascontiguousarray(y, dtype=np.float64)
    o = Opts()
    o.iterations, o.CG_max_iter, o.EM_max_iter = iterations, CG_max_iter, EM_max_iter
    o.EM_err_thr, o.stop_criteria_thr, o.rho = EM_err_thr, stop_criteria_thr, rho
    o.learn_vars, o.seed, o.use_lmmse_damp, o.gam1, o.gamw = learn_vars, seed, use_lmmse_damp, gam1, gamw
    keep = []
    if probs is not None and len(probs):
        pr, vr = np.ascontiguousarray(probs, dtype=np.float64), np.ascontiguousarray(vars_, dtype=np.float64)
        keep += [pr, vr]
        o.L, o.probs, o.vars = pr.size, _dp(pr), _dp(vr)
    else:
        o.L = 0
    o.out_prefix = out_prefix.encode() if out_prefix else None
    o.verbose, o.diagnostics, o.alpha_scale = verbose, diagnostics, alpha_scale
    o.use_XXT_denoiser = use_XXT_denoiser
    o.bin_class, o.probit_var = int(model == "bin_class"), probit_var
    o.fuse_solves = fuse_solves
    o.reanchor_every = reanchor_every
    o.freeze_index_file = freeze_index_file.encode() if freeze_index_file else None
    cov_eff = None
    if covs is not None:
        cz = np.ascontiguousarray(covs, dtype=np.float64)
        cov_eff = np.zeros(cz.shape[1])
        keep += [cz, cov_eff]
        o.C, o.covs, o.cov_eff_out = cz.shape[1], _dp(cz), _dp(cov_eff)
    M = shard.M
    ts = np.ascontiguousarray(true_signal, dtype=np.float64) if true_signal is not None else None
    m4 = np.ascontiguousarray(mask4, dtype=np.uint8) if mask4 is not None else None
    x_est = np.zeros(max(M, 1))
    iters = (Iter * iterations)()
    n = C.c_int()
    hist = [np.zeros((iterations, max(M, 1))) if history else None for _ in range(3)]
    pout, vout, Lout = np.zeros(64), np.zeros(64), C.c_int()
    rc = L.gvh_infere_linear(shard.h, C.byref(o), shard.N, M, shard.Mt, shard.S, rank, _dp(y),
                             m4.ctypes.data_as(C.POINTER(C.c_ubyte)) if m4 is not None else None,
                             shard.N if nonas is None else nonas, _dp(ts), _dp(x_est), iters, iterations, C.byref(n),
                             _dp(hist[0]), _dp(hist[1]), _dp(hist[2]), _dp(pout), _dp(vout), C.byref(Lout))
    if rc:
        raise capi.GvError("gvh_infere_linear failed: " + L.gvh_last_error().decode())
    r = Result()
    r.niter = n.value
    r.x_est = x_est[:M].copy()
    r.trace = []
    for i in range(n.value):
        t = {f: getattr(iters[i], f) for f in ITER_FIELDS}
        for f in ("cg_iters", "onsager_iters", "revar_rounds", "L_after", "n_ax", "n_atx", "seconds", "seconds_io",
                  "beta1", "tau2", "tau1_next", "n_ax_pass", "n_atx_pass", "probe_product"):
            t[f] = getattr(iters[i], f)
        r.trace.append(t)
    if history:
        r.x1 = [hist[0][i, :M].copy() for i in range(n.value)]
        r.x2 = [hist[1][i, :M].copy() for i in range(n.value)]
        r.r1 = [hist[2][i, :M].copy() for i in range(n.value)]
    r.probs, r.vars = pout[:Lout.value].copy(), vout[:Lout.value].copy()
    r.cov_eff = cov_eff
    return r
