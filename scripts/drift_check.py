"""Do the fuse levels drift?  The by-products of levels 3 and 4 chain from iteration to iteration (A x2_hat and A^T A x2_hat are never
re-anchored on an explicit product, docs/history/rounds1-3.md section 5).  Long runs on independent and on block-correlated genotypes at every level
against the same run issuing the reference's own sequence of products (level 0): x1_hat per iteration, gamw and step counts.
  python scripts/drift_check.py [N] [M] [iterations]        (development; run on a GPU box)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gvamp_amd import capi, hostapi

N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 500000
IT = int(sys.argv[3]) if len(sys.argv) > 3 else 40
XXT = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # 1: --use-XXT-denoiser 1 (level 4 also carries A r1 / A r2 by linearity)
REANCHOR = int(sys.argv[5]) if len(sys.argv) > 5 else -1   # --reanchor-every (-1: the drivers' default, 10; 0: never)


def rel(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


out = []
for ld in (0, 64):
    with capi.Shard(N, M) as sh:
        sh.set_kernel_mode(1)
        if ld:
            sh.synth_bed(77, 5000, ld_block=ld, ld_ppm=900000)
        else:
            sh.synth_bed(77, 5000)
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, max(1, M // 100), 1)
        kw = dict(iterations=IT, CG_max_iter=50, rho=0.5, seed=1, gam1=1e-8, gamw=2.0, true_signal=beta, history=True,
                  stop_criteria_thr=1e-14, use_XXT_denoiser=XXT, reanchor_every=REANCHOR)
        runs = {f: hostapi.infere_linear(sh, y, None, None, fuse_solves=f, **kw) for f in (0, 1, 2, 3, 4)}
    r0, r3 = runs[0], runs[4]
    row = {"genotypes": "LD blocks of %d" % ld if ld else "independent markers", "N": N, "M": M, "iterations": r3.niter,
           "use_XXT_denoiser": XXT, "reanchor_every": REANCHOR,
           "cg_iters": [t["cg_iters"] for t in r0.trace],
           # r2 = c1 x1_hat - c2 r1 with c1 - c2 = 1: what XXT level 4's A r2 by linearity amplifies rounding by (vamp.cpp: linearity_max)
           "c1_plus_c2": [float("%.3g" % ((2 * t["eta1"] - t["gam2"]) / t["gam2"])) for t in r0.trace],
           # ... and r1 = (eta2 x2_hat - gam2 r2) / gam1 the same way for the A r1 carried into the next iteration
           "r1_amplification": [float("%.3g" % ((t["eta2"] + t["gam2_reest"]) / t["gam1_next"])) for t in r0.trace],
           "passes": {f: sum(t["n_ax_pass"] + t["n_atx_pass"] for t in runs[f].trace) for f in runs},
           "seconds": {f: round(sum(t["seconds"] for t in runs[f].trace), 3) for f in runs}}
    for f in (1, 2, 3, 4):
        r = runs[f]
        per_it = [rel(a, b) for a, b in zip(r.x1[1:], r0.x1[1:])]
        row["level_%d_vs_0" % f] = {
            "x1_hat_rel_l2_by_iteration": [float("%.1e" % e) for e in per_it],
            "x_est_rel_l2": float("%.2g" % rel(r.x_est, r0.x_est)),
            "gamw_rel_max": float("%.2g" % max(abs(a["gamw"] - b["gamw"]) / abs(b["gamw"]) for a, b in zip(r.trace, r0.trace))),
            "identical_step_counts": all((a["cg_iters"], a["onsager_iters"], a["L_after"]) == (b["cg_iters"], b["onsager_iters"], b["L_after"])
                                         for a, b in zip(r.trace, r0.trace))}
    print(json.dumps(row), flush=True)
