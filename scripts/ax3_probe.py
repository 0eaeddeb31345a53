"""Development: the three-vector Ax pass (gv_ax3_dev, tile layout) against single products (bit for bit) and against the two-vector
pass (time).   python scripts/ax3_probe.py [N M ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gvamp_amd import capi

shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)] or \
    [(2000, 3000), (1003, 517), (50000, 200000), (100000, 500000), (400000, 125000), (400000, 1000000)]
for N, M in shapes:
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, 2)
        sh.synth_bed(7, 5000)
        sh.compute_markers_statistics()
        rng = np.random.default_rng(N + M)
        xs = [sh.vecM(rng.standard_normal(M) * s) for s in (1.0, 1e-3, 40.0)]
        o3 = [sh.vecN() for _ in range(3)]
        o1 = [sh.vecN() for _ in range(3)]
        sh.ax3_dev(xs[0], xs[1], xs[2], o3[0], o3[1], o3[2])
        for k in range(3):
            sh.ax_dev(xs[k], o1[k])
        same = [bool(np.array_equal(o3[k].download(), o1[k].download())) for k in range(3)]
        t = {}
        for name, fn in (("ax", lambda: sh.ax_dev(xs[0], o1[0])), ("ax2", lambda: sh.ax2_dev(xs[0], xs[1], o1[0], o1[1])),
                         ("ax3", lambda: sh.ax3_dev(xs[0], xs[1], xs[2], o3[0], o3[1], o3[2]))):
            fn(); sh.synchronize()
            reps = 20 if N * M < 2e10 else 5
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            sh.synchronize()
            t[name] = (time.perf_counter() - t0) / reps * 1e3
        print("N=%d M=%d  bit-identical %s   ax %.3f ms  ax2 %.3f ms  ax3 %.3f ms  (ax3 / ax2 = %.3f; ax2 + ax = %.3f)  decomp %s" % (
            N, M, same, t["ax"], t["ax2"], t["ax3"], t["ax3"] / t["ax2"], t["ax2"] + t["ax"], sh.decomp()["ax2"]), flush=True)
