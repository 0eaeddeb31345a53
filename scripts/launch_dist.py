"""Development: the DISTRIBUTION of single-launch durations of one streaming-kernel class under candidate work decompositions, on ONE
resident shard (scripts/ab_decomp.py prints means over groups of launches; a class whose launches are bimodal -- some land 25-35 %
above the rest -- hides in a mean).  HIP events around every streaming-kernel launch (gv_set_timing(2)), one counter read per launch.
  python scripts/launch_dist.py N M cls [--layout L] [--launches K] spec ...
cls: atx | atx2 | ax | ax2;  spec: "ks=4,geo=0.5,prio=1" | "cells=995,prio=1" | "cells=218,quads=1536,prio=1" | "ks=6,geo=0.6,prio=1,occ=2" | "tuned"
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi

if os.environ.get("GV_DBG_LIB"):          # a variant build of the library (scripts/build_variant.sh)
    capi.LIB_PATH = os.environ["GV_DBG_LIB"]
ap = argparse.ArgumentParser()
ap.add_argument("N", type=int)
ap.add_argument("M", type=int)
ap.add_argument("cls")
ap.add_argument("--layout", type=int, default=0, help="0: the library's choice (default), 1 two stripe sets, 2 tile")
ap.add_argument("--launches", type=int, default=40)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("specs", nargs="+")
a = ap.parse_args()
cls = ("atx", "atx2", "ax", "ax2").index(a.cls)
with capi.Shard(a.N, a.M) as sh:
    if a.layout:
        sh.set_layout(False, a.layout)
        sh.set_kernel_mode(1)
    sh.synth_bed(1234, 5000)
    sh.compute_markers_statistics()
    rng = np.random.default_rng(0)
    x, x2, p, p2, w, w2 = sh.vecM(rng.standard_normal(a.M)), sh.vecM(rng.standard_normal(a.M)), sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM()
    sh.ax_dev(x, p); sh.ax_dev(x2, p2)          # (tunes / loads the picks)
    tuned = sh.decomp()[a.cls]
    print("N=%d M=%d %s layout %d; library pick: %s %s" % (a.N, a.M, a.cls, sh.get_layout(), tuned, sh.tune_info()), flush=True)
    f = {0: lambda: sh.atx_dev(p, w), 2: lambda: sh.ax_dev(x, p), 1: lambda: sh.atx2_dev(p, p2, w, w2), 3: lambda: sh.ax2_dev(x, x2, p, p2)}[cls]
    other = {0: lambda: sh.ax_dev(x, p), 1: lambda: sh.ax2_dev(x, x2, p, p2), 2: lambda: sh.atx_dev(p, w), 3: lambda: sh.atx2_dev(p, p2, w, w2)}[cls]
    kms, kn = ("ms_atx_kernel", "n_atx_kernel") if cls < 2 else ("ms_ax_kernel", "n_ax_kernel")

    def apply(spec):
        if spec == "tuned":
            kw = dict(ks=tuned.get("ks", 1), balanced_cells=tuned.get("balanced_cells", 0), whole_quads=tuned.get("whole_quads", 0),
                      prio=tuned["prio"], taper=tuned.get("taper", 0.0), geo=tuned.get("geo", 0.0), wgs_per_cu=tuned.get("wgs_per_cu", 0),
                      xcd_skew=tuned.get("xcd_skew", 0.0))
        else:
            d = dict(kv.split("=") for kv in spec.split(","))
            kw = dict(ks=int(d.get("ks", 1)), balanced_cells=int(d.get("cells", 0)), whole_quads=int(d.get("quads", 0)), prio=int(d.get("prio", 0)),
                      taper=float(d.get("taper", 0)), geo=float(d.get("geo", 0)), wgs_per_cu=int(d.get("occ", 0)),
                      xcd_skew=float(d.get("skew", 0)))
        sh.set_decomp(cls, **kw)

    f(); ref = w.download() if cls < 2 else p.download()
    res = {s: [] for s in a.specs}
    sh.set_timing(2)
    for r in range(a.rounds):
        for s in a.specs:
            apply(s)
            other(); f()                                      # first launch of a new grid shape untimed
            for _ in range(a.launches):
                other()
                sh.counters(reset=True)
                f()
                c = sh.counters()
                res[s].append(c[kms] / max(1, c[kn]))
            got = w.download() if cls < 2 else p.download()
            assert np.array_equal(got, ref), "decomposition changed the result: " + s
    mb = (a.N + 3) // 4
    nbytes = a.M * mb + 24 * a.M + 32 * mb
    print("%-34s %7s %7s %7s %7s %7s  %s" % ("streaming kernel alone, ms", "p10", "p50", "p90", "max", "mean", "launches > 1.1 x p50;  GB/s at p50 / mean"))
    for s in a.specs:
        v = np.array(res[s])
        p50 = np.percentile(v, 50)
        print("%-34s %7.4f %7.4f %7.4f %7.4f %7.4f  %d of %d;  %.0f / %.0f" % (s, np.percentile(v, 10), p50, np.percentile(v, 90), v.max(), v.mean(),
                                                                          int((v > 1.1 * p50).sum()), len(v), nbytes / p50 / 1e6, nbytes / v.mean() / 1e6))
