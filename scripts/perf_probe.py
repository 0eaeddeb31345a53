"""Quick Ax/ATx throughput probe on one GPU (development tool; bench.py is the contract)."""
import argparse
import sys
import time
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi
if os.environ.get("GV_DBG_LIB"):
    capi.LIB_PATH = os.environ["GV_DBG_LIB"]          # development: an experimental build of the library

ap = argparse.ArgumentParser()
ap.add_argument("--N", type=int, default=100000)
ap.add_argument("--M", type=int, default=100000)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("--stripes-only", type=int, default=0, help="1: two stripe sets, 2: one tile layout (no raw rows resident)")
a = ap.parse_args()

with capi.Shard(a.N, a.M) as sh:
    t = time.time()
    if a.stripes_only:
        sh.set_layout(False, a.stripes_only)
    sh.synth_bed(1234, 5000)
    sh.compute_markers_statistics()
    print("synth+stats s", time.time() - t, flush=True)
    if a.mode:
        sh.set_kernel_mode(a.mode)
    print("copy GB/s", sh.copy_bandwidth(1 << 30, 10), flush=True)
    rng = np.random.default_rng(0)
    x, p = sh.vecM(rng.standard_normal(a.M)), sh.vecN()
    w = sh.vecM()
    mb = (a.N + 3) // 4
    nbytes = a.M * mb + 24 * a.M + 32 * mb
    sh.ax_dev(x, p); sh.atx_dev(p, w); sh.synchronize()
    sh.set_timing(1)
    sh.counters(reset=True)
    for _ in range(a.reps):
        sh.ax_dev(x, p)
        sh.atx_dev(p, w)
    c = sh.counters()
    print(c, sh.decomp(), sh.tune_info())
    print("Ax  ms %.3f  GB/s %.1f" % (c["ms_ax"] / c["n_ax"], nbytes / (c["ms_ax"] / c["n_ax"] * 1e-3) / 1e9))
    print("ATx ms %.3f  GB/s %.1f" % (c["ms_atx"] / c["n_atx"], nbytes / (c["ms_atx"] / c["n_atx"] * 1e-3) / 1e9))
    # two-vector passes
    x2, p2, w2 = sh.vecM(rng.standard_normal(a.M)), sh.vecN(), sh.vecM()
    sh.ax2_dev(x, x2, p, p2); sh.atx2_dev(p, p2, w, w2); sh.synchronize()
    sh.counters(reset=True)
    for _ in range(a.reps):
        sh.ax2_dev(x, x2, p, p2)
        sh.atx2_dev(p, p2, w, w2)
    c = sh.counters()
    print("Ax2  ms %.3f  (x2 vectors) GB/s per pass %.1f" % (c["ms_ax"] / a.reps, nbytes / (c["ms_ax"] / a.reps * 1e-3) / 1e9))
    print("ATx2 ms %.3f  (x2 vectors) GB/s per pass %.1f" % (c["ms_atx"] / a.reps, nbytes / (c["ms_atx"] / a.reps * 1e-3) / 1e9))
