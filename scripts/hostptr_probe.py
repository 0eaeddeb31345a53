"""Where a host-pointer matvec (gv_ax / gv_atx: data::Ax / data::ATx with the reference's signatures) spends its time:
vector upload, device product, download -- at config-2 size by default.  GV_XFER_THREADS=0 shows the single-threaded staging copy."""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import numpy as np
from gvamp_amd import capi

N, M = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (100000, 500000)
mb = (N + 3) // 4
nbytes = M * mb + 24 * M + 32 * mb
with capi.Shard(N, M) as sh:
    sh.synth_bed(1234, 5000)
    sh.compute_markers_statistics()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(M)
    p = np.zeros(4 * mb)
    p[:N] = rng.standard_normal(N)
    dx, dz, dp, dw = sh.vecM(x), sh.vecN(), sh.vecN(p), sh.vecM()
    sh.Ax(x); sh.ATx(p)

    def t(f, reps=20):
        f(); sh.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                f()
            sh.synchronize()
            best = min(best, (time.perf_counter() - t0) / reps)
        return best * 1e3

    rows = [("upload M-vector (%.1f MB)" % (M * 8 / 1e6), t(lambda: dx.upload(x))),
            ("upload N-vector (%.1f MB)" % (4 * mb * 8 / 1e6), t(lambda: dp.upload(p))),
            ("download M-vector", t(lambda: dw.download())),
            ("download N-vector", t(lambda: dz.download())),
            ("Ax on handles", t(lambda: sh.ax_dev(dx, dz))),
            ("ATx on handles", t(lambda: sh.atx_dev(dp, dw))),
            ("gv_ax host pointers", t(lambda: sh.Ax(x))),
            ("gv_atx host pointers", t(lambda: sh.ATx(p)))]
    for name, ms in rows:
        print("%-32s %8.3f ms" % (name, ms))
    ax, atx = rows[6][1], rows[7][1]
    print("host-pointer rate: Ax %.0f GB/s, ATx %.0f GB/s, pair %.0f GB/s (GV_XFER_THREADS=%s, layout %d)"
          % (nbytes / ax / 1e6, nbytes / atx / 1e6, 2 * nbytes / (ax + atx) / 1e6, os.environ.get("GV_XFER_THREADS", "default"), sh.get_layout()))
