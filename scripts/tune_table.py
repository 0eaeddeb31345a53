"""Measures the work decompositions of the shapes BASELINE.json names (and the per-GPU shards of the headline job) on this GPU
and writes gvamp_amd/csrc/gv_tune_builtin.h for the current kernel sources.  Run on an MI355X after the LAST change to
gv_mfma.hip / gv_mfma.h:   gpurun -- 'python scripts/tune_table.py && cp gvamp_amd/csrc/gv_tune_builtin.h gpurun_out/'
Each shape is measured `--votes` times in fresh contexts (cache and table off); the distinct picks of a class then meet in a play-off
inside one context (kernel-alone HIP events, launches interleaved), and the lowest mean is shipped."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GV_TUNE_CACHE"] = "0"
os.environ["GV_TUNE_BUILTIN"] = "0"
import ctypes as C

import numpy as np
from gvamp_amd import build, capi

SHAPES = [(400000, 1000000), (400000, 500000), (400000, 250000), (400000, 125000), (100000, 500000), (50000, 200000),
          (2000, 10000)]
ap = argparse.ArgumentParser()
ap.add_argument("--votes", type=int, default=3)
ap.add_argument("--out", default=os.path.join(build.CSRC, "gv_tune_builtin.h"))
a = ap.parse_args()
rows = []
for N, M in SHAPES:
    for layout in (1, 2):
        if layout == 1 and 2 * M * ((N + 3) // 4) > 250e9:
            continue
        votes = [collections.Counter() for _ in range(4)]
        for _ in range(a.votes):
            with capi.Shard(N, M) as sh:
                sh.set_layout(False, layout)
                sh.synth_bed(1234, 5000)
                sh.compute_markers_statistics()
                # the first matvec measures -- on a representative operand: a constant vector has one non-zero digit plane, the
                # MFMA pipe idles and the ranking of the candidates changes (measured: 7 of 7 votes for a uniform split on
                # ones where a normal vector gives the hybrid by 3 %)
                sh.Ax(np.random.default_rng(N + M).standard_normal(M))
                assert sh.tune_info()[1] == "measured", sh.tune_info()
                d = (capi.DecompInfo * 4)()
                sh._ck(sh.L.gv_get_decomp(sh.h, d))
                for k in range(4):
                    # (xcd_skew is NOT shipped: which four XCDs finish equal shares first changed from box to box and from one allocation
                    # to the next -- the same sign gained 1.3 % on one box's 100 GB shard and lost 0.8 % on another's 12.5 GB one
                    # (profiles/r6_xcd_skew.txt) -- so it stays something autotune_ks measures on the resident data itself)
                    votes[k][(d[k].ks, int(d[k].balanced_cells), int(d[k].whole_quads), d[k].prio, round(float(d[k].taper), 2), round(float(d[k].geo), 2),
                              2 if d[k].wgs_per_cu == 2 else 0, 0.0)] += 1
        # play-off: the distinct picks of every class (votes split 1-1-1 where candidates are within the +-1 % a single 15 ms run moves
        # by) measured against each other in ONE context, kernel-alone HIP events, launches interleaved round-robin inside the Ax ->
        # ATx sequence the solvers issue; the lowest mean wins
        with capi.Shard(N, M) as sh:
            sh.set_layout(False, layout)
            sh.synth_bed(1234, 5000)
            sh.compute_markers_statistics()
            rng = np.random.default_rng(N + M)
            x, x2, p, p2, w, w2 = sh.vecM(rng.standard_normal(M)), sh.vecM(rng.standard_normal(M)), sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM()
            sh.ax_dev(x, p); sh.ax_dev(x2, p2)
            run = {0: lambda: sh.atx_dev(p, w), 1: lambda: sh.atx2_dev(p, p2, w, w2), 2: lambda: sh.ax_dev(x, p), 3: lambda: sh.ax2_dev(x, x2, p, p2)}
            other = {0: 2, 1: 3, 2: 0, 3: 1}
            picks, playoff = [], []
            for k in (2, 0, 3, 1):              # the Ax side first, then the ATx side against the chosen Ax (as autotune_ks does)
                cands = list(votes[k])
                if len(cands) == 1:
                    best, times = cands[0], {}
                else:
                    times = {c: [] for c in cands}
                    key = ("ms_atx_kernel", "n_atx_kernel") if k < 2 else ("ms_ax_kernel", "n_ax_kernel")
                    big = N * M > 4e10
                    for rnd in range(3 if big else 6):
                        for c in cands:
                            ks, skl, piv, prio, taper, geo, occ, xs = c
                            sh.set_decomp(k, ks=ks, balanced_cells=skl, whole_quads=piv, prio=prio, taper=taper, geo=geo, wgs_per_cu=occ, xcd_skew=xs)
                            run[other[k]](); run[k]()                       # first launch of a grid shape: untimed
                            sh.set_timing(2)
                            for _ in range(2 if big else 5):
                                run[other[k]]()
                                sh.counters(reset=True)
                                run[k]()
                                cn = sh.counters()
                                times[c].append(cn[key[0]] / max(1, cn[key[1]]))
                            sh.set_timing(0)
                    best = min(cands, key=lambda c: float(np.mean(times[c])))
                ks, skl, piv, prio, taper, geo, occ, xs = best
                sh.set_decomp(k, ks=ks, balanced_cells=skl, whole_quads=piv, prio=prio, taper=taper, geo=geo, wgs_per_cu=occ, xcd_skew=xs)
                picks.append((k, best))
                playoff.append({str(c): round(float(np.mean(t)), 4) for c, t in times.items()})
            picks = [b for k, b in sorted(picks)]
            playoff = [playoff[[2, 0, 3, 1].index(k)] for k in range(4)]
        print(N, M, layout, picks, [dict(v) for v in votes], "play-off (ms):", playoff, flush=True)
        rows.append((N, M, layout - 1, picks))
h = build.kernel_src_hash()[:16]
src = open(os.path.join(build.CSRC, "gv_tune_builtin.h")).read()      # (the head of the shipped file; --out may not exist yet)
head = src[:src.index("static const char* const GV_BUILTIN_FOR_HASH")]
body = 'static const char* const GV_BUILTIN_FOR_HASH = "%s";\nstatic const BuiltinPick GV_BUILTIN_PICKS[] = {\n' % h
for N, M, lay, picks in rows:
    ds = ", ".join("{%d, %d, %d, %d, %.2ff, %.2ff, %d, %.3ff}" % (ks, skl, piv, prio, taper, geo, occ, xs) for ks, skl, piv, prio, taper, geo, occ, xs in picks)
    body += "    {%d, %d, %d, {%s}},\n" % (N, M, lay, ds)
body += "};\n}  // namespace gvi\n"
open(a.out, "w").write(head + body)
print("wrote", a.out, "for kernel sources", h)
