"""Development: a long run of the shipped engine at the headline size (30 + 5 + 5 VAMP iterations on one resident shard), free HBM watched."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gvamp_amd import capi, hostapi
N, M = 400000, 1000000
free0 = torch.cuda.mem_get_info()[0]
with capi.Shard(N, M) as sh:
    sh.set_expected_passes(500); sh.synth_bed(1234, 5000); sh.compute_markers_statistics()      # nothing else configured: the shipped engine
    beta, y = hostapi.sim_phen(sh, 0.5, M // 100, 1)
    free1 = torch.cuda.mem_get_info()[0]
    for rep in range(3):
        t = time.time()
        r = hostapi.infere_linear(sh, y, None, None, iterations=30 if rep == 0 else 5, CG_max_iter=50, rho=0.5, seed=1, true_signal=beta,
                                  history=False, fuse_solves=4, stop_criteria_thr=1e-12)
        sh.synchronize()
        f = torch.cuda.mem_get_info()[0]
        print("rep", rep, "iters", r.niter, "wall %.2f s" % (time.time() - t), "free GB %.3f (delta vs after-ingest %.1f MB)" % (f / 1e9, (free1 - f) / 1e6),
              "corr %.4f" % np.corrcoef(r.x_est, beta)[0, 1], "gamw %.4f" % r.trace[-1]["gamw"], "s/iter %.4f" % np.mean([t_["seconds"] for t_ in r.trace[1:]]), flush=True)
    assert np.all(np.isfinite(r.x_est))
free2 = torch.cuda.mem_get_info()[0]
print("after close: leaked MB %.1f" % ((free0 - free2) / 1e6))
