"""Development: time of compute_markers_statistics on a resident shard (k_stats_stripes / k_stats_tile).  python scripts/stats_rate.py N M [layout]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gvamp_amd import capi

N, M = int(sys.argv[1]), int(sys.argv[2])
layout = int(sys.argv[3]) if len(sys.argv) > 3 else 1
with capi.Shard(N, M) as sh:
    sh.set_layout(False, layout)
    sh.synth_bed(4242, 5000)
    sh.compute_markers_statistics()
    sh.synchronize()
    ts = []
    for _ in range(5):
        t = time.perf_counter()
        sh.compute_markers_statistics()          # (synchronises)
        ts.append(time.perf_counter() - t)
    nbytes = M * ((N + 3) // 4)
    print("layout %d unroll %s: stats min %.3f ms median %.3f ms = %.0f GB/s" % (layout, os.environ.get("GV_STATS_UNROLL", "default"), min(ts) * 1e3,
          sorted(ts)[2] * 1e3, nbytes / min(ts) / 1e9))
