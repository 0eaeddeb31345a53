"""Development: is a streaming kernel as fast between other kernels as it is back to back?  python scripts/probes/alt_probe.py N M"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from gvamp_amd import capi

N, M = int(sys.argv[1]), int(sys.argv[2])
with capi.Shard(N, M) as sh:
    sh.set_layout(False, True)
    sh.synth_bed(1234, 5000)
    sh.set_kernel_mode(1)
    sh.compute_markers_statistics()
    rng = np.random.default_rng(0)
    x, x2, p, p2, w, w2 = sh.vecM(rng.standard_normal(M)), sh.vecM(rng.standard_normal(M)), sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM()
    sh.ax2_dev(x, x2, p, p2); sh.atx2_dev(p, p2, w, w2); sh.synchronize()
    sh.set_timing(1)
    for name, seq in (("Ax2 back to back", "AAAAAAAA"), ("Ax2 alternating with ATx2", "ATATATATATATATAT"), ("ATx2 back to back", "TTTTTTTT")):
        sh.counters(reset=True)
        for ch in seq:
            if ch == "A":
                sh.ax2_dev(x, x2, p, p2)
            else:
                sh.atx2_dev(p, p2, w, w2)
        c = sh.counters()
        na, nt = seq.count("A"), seq.count("T")
        print("%-28s Ax2 %.4f ms  ATx2 %.4f ms" % (name, c["ms_ax"] / max(na, 1), c["ms_atx"] / max(nt, 1)))
