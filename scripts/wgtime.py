"""Development: per-workgroup start / end clocks of one streaming-kernel launch (needs the -DGV_WGTIME build of the library,
path in GV_DBG_LIB).  python scripts/wgtime.py N M which(atx|ax|atx2|ax2)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi

capi.LIB_PATH = os.environ["GV_DBG_LIB"]
N, M, which = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
with capi.Shard(N, M) as sh:
    sh.set_layout(False, True)
    sh.synth_bed(1234, 5000)
    sh.set_kernel_mode(1)
    sh.compute_markers_statistics()
    rng = np.random.default_rng(0)
    x, x2, p, p2, w, w2 = sh.vecM(rng.standard_normal(M)), sh.vecM(rng.standard_normal(M)), sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM()
    sh.ax_dev(x, p); sh.ax_dev(x2, p2)
    f = {"atx": lambda: sh.atx_dev(p, w), "ax": lambda: sh.ax_dev(x, p), "atx2": lambda: sh.atx2_dev(p, p2, w, w2),
         "ax2": lambda: sh.ax2_dev(x, x2, p, p2)}[which]
    L = capi.load()
    L.gv_debug_wgtime.argtypes = [C.c_void_p, C.c_int]
    for _ in range(3):
        f()
    sh.synchronize()
    assert L.gv_debug_wgtime(None, 0) == 0          # reset, then ONE launch
    f()
    sh.synchronize()
    n = 16384
    buf = (C.c_ulonglong * (4 * n))()
    assert L.gv_debug_wgtime(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
    a = a[a[:, 1] > 0]
    t0 = a[:, 0].min()
    st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0          # wall_clock64: 100 MHz -> us
    dur = en - st
    print("workgroups %d  kernel span %.1f us" % (len(a), en.max()))
    print("start  us: p0 %.1f p50 %.1f p90 %.1f p100 %.1f" % tuple(np.percentile(st, [0, 50, 90, 100])))
    print("end    us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f" % tuple(np.percentile(en, [0, 10, 50, 90, 100])))
    print("dur    us: p0 %.1f p10 %.1f p50 %.1f p90 %.1f p100 %.1f" % tuple(np.percentile(dur, [0, 10, 50, 90, 100])))
    first = st < 20
    print("round-1 workgroups (start < 20 us): %d, their dur p50 %.1f ; later ones: %d, dur p50 %.1f, start p50 %.1f" % (
        first.sum(), np.median(dur[first]), (~first).sum(), np.median(dur[~first]) if (~first).any() else 0,
        np.median(st[~first]) if (~first).any() else 0))
    # chip occupancy over time: resident workgroups sampled every 10 us
    step = 10.0 if en.max() < 1000 else en.max() / 60
    ts = np.arange(0, en.max(), step)
    occ = [(int(((st <= t) & (en > t)).sum())) for t in ts]
    print("resident wgs every %.0f us:" % step, " ".join(str(o) for o in occ))
    xcc = a[:, 2]
    for k in range(8):
        m = xcc == k
        if m.any():
            print("  xcc %d: %4d wgs  end p50 %.1f max %.1f  dur p50 %.1f" % (k, m.sum(), np.median(en[m]), en[m].max(), np.median(dur[m])))
