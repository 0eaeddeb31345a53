import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from gvamp_amd import capi, hostapi
def cycle():
    with capi.Shard(100000, 200000) as sh:
        sh.set_kernel_mode(1); sh.synth_bed(1, 5000); sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, 2000, 1)
        hostapi.infere_linear(sh, y, None, None, iterations=3, CG_max_iter=20, rho=0.5, seed=1, true_signal=beta, history=False, fuse_solves=2)
        hostapi.infere_linear(sh, y, None, None, iterations=2, CG_max_iter=20, rho=0.5, seed=1, true_signal=beta, history=False, use_XXT_denoiser=1)
        sh.pvals_calc(sh.vecN(y), sh.vecN(y), sh.vecM(beta))
torch.cuda.init()
f=[torch.cuda.mem_get_info()[0]]
for i in range(4):
    cycle(); f.append(torch.cuda.mem_get_info()[0])
print([round((f[0]-x)/1e6,1) for x in f])
