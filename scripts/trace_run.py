"""A few VAMP iterations at a given shard shape, to be run under `rocprofv3 --kernel-trace` (development tool).
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 scripts/trace_run.py N M [iters] [fuse] [xxt]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from gvamp_amd import capi, hostapi

N, M = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
fuse = int(sys.argv[4]) if len(sys.argv) > 4 else 2
xxt = int(sys.argv[5]) if len(sys.argv) > 5 else 0
with capi.Shard(N, M) as sh:
    if os.environ.get("GV_LAYOUT"):
        sh.set_layout(False, int(os.environ["GV_LAYOUT"]))      # 1 two stripe sets, 2 one tile layout
    else:
        sh.set_expected_passes(iters * 12)                      # what the drivers announce: the layout a run of this length gets
    sh.synth_bed(4242, 5000)
    sh.compute_markers_statistics()
    beta, y = hostapi.sim_phen(sh, 0.5, max(1, M // 100), 1)
    prior = (None, None) if M >= 50000 else ([0.9, 0.1], [0, 0.5 / max(1, M // 100) * 0.1])
    r = hostapi.infere_linear(sh, y, prior[0], prior[1], iterations=iters, CG_max_iter=50, rho=0.5, seed=1, true_signal=beta,
                              history=False, fuse_solves=fuse, use_XXT_denoiser=xxt)
for t in r.trace:
    print(round(t["seconds"], 4), t["cg_iters"], t["n_ax_pass"] + t["n_atx_pass"])
