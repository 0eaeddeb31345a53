"""Development: mean wall time of the VAMP iterations 3.. of one long run at a given shape (A/B of a library switch across processes).
  python scripts/iter_time.py N M iterations [fuse] [xxt]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi, hostapi

N, M, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
fuse = int(sys.argv[4]) if len(sys.argv) > 4 else 4
xxt = int(sys.argv[5]) if len(sys.argv) > 5 else 0
with capi.Shard(N, M) as sh:
    sh.set_expected_passes(iters * 12)
    sh.synth_bed(4242, 5000)
    sh.compute_markers_statistics()
    beta, y = hostapi.sim_phen(sh, 0.5, max(1, M // 100), 1)
    kw = dict(iterations=iters, CG_max_iter=50, rho=0.5, seed=1, true_signal=beta, history=False, fuse_solves=fuse, use_XXT_denoiser=xxt,
              stop_criteria_thr=1e-30)
    hostapi.infere_linear(sh, y, None, None, **dict(kw, iterations=3))       # picks, clocks
    r = hostapi.infere_linear(sh, y, None, None, **kw)
t = np.array([x["seconds"] for x in r.trace[2:]])
p = np.array([x["n_ax_pass"] + x["n_atx_pass"] for x in r.trace[2:]])
print("%d iterations: mean %.4f ms, median %.4f ms per iteration; %.2f passes per iteration; %.4f ms per pass" % (
    len(t), 1e3 * t.mean(), 1e3 * np.median(t), p.mean(), 1e3 * t.sum() / p.sum()))
