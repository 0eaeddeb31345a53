"""development: host wall time per call of gv_denoise / gv_prior_estep (kernel + finalisation + read-back) at the row shapes' M and prior sizes,
and the values against the oracle.   python scripts/probes/em_kernels.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gvamp_amd import capi
from oracle import gvoracle as oracle

for M, L in ((125000, 23), (200000, 16), (500000, 16), (1000000, 23)):
    with capi.Shard(256, M) as sh:
        rng = np.random.default_rng(M)
        r = rng.standard_normal(M) * np.where(rng.random(M) < 0.02, 3.0, 0.3)
        probs = np.r_[0.97, np.full(L - 1, 0.03 / (L - 1))]
        vars_ = np.r_[0.0, 1e-4 * 1.8 ** np.arange(L - 1)] * 256
        gam1 = 7.0
        r1, x1, dd = sh.vecM(r), sh.vecM(), sh.vecM()
        lam = 1 - probs[0]
        om = probs.copy(); om[1:] /= lam
        for _ in range(20):
            s_d = sh.denoise(r1, gam1, probs, vars_, x1, dd)
            s_e = sh.prior_estep(r1, gam1, lam, om, vars_)
        t = []
        for _ in range(300):
            t0 = time.perf_counter(); sh.denoise(r1, gam1, probs, vars_, x1, dd); t.append(time.perf_counter() - t0)
        td = np.median(t) * 1e6
        t = []
        for _ in range(300):
            t0 = time.perf_counter(); sh.prior_estep(r1, gam1, lam, om, vars_); t.append(time.perf_counter() - t0)
        te = np.median(t) * 1e6
        o_x, o_d = oracle.g1_g1d(r, gam1, probs, vars_)
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        print("M=%7d L=%2d: denoise %6.1f us, estep %6.1f us per call;  x1 rel %.2e  d rel %.2e  sums rel %.2e %.2e  estep sum[0] %.17g" % (
            M, L, td, te, rel(x1.download(), o_x), rel(dd.download(), o_d), abs(s_d[0] - o_d.sum()) / abs(o_d.sum()),
            abs(s_d[1] - ((o_x - r) ** 2).sum()) / ((o_x - r) ** 2).sum(), s_e[0]), flush=True)
