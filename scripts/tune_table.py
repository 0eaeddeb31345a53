"""Measures the work decompositions of the shapes BASELINE.json names (and the per-GPU shards of the headline job) on this GPU
and writes gvamp_amd/csrc/gv_tune_builtin.h for the current kernel sources.  Run on an MI355X after the LAST change to
gv_mfma.hip / gv_mfma.h:   gpurun -- 'python scripts/tune_table.py && cp gvamp_amd/csrc/gv_tune_builtin.h gpurun_out/'
Each shape is measured `--votes` times in fresh contexts (cache and table off); a class keeps the pick that came up most often."""
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
                    votes[k][(d[k].ks, int(d[k].balanced_cells), int(d[k].whole_quads), d[k].prio, round(float(d[k].taper), 2), round(float(d[k].geo), 2),
                              2 if d[k].wgs_per_cu == 2 else 0)] += 1
        picks = [v.most_common(1)[0][0] for v in votes]
        print(N, M, layout, picks, [dict(v) for v in votes], flush=True)
        rows.append((N, M, layout - 1, picks))
h = build.kernel_src_hash()[:16]
src = open(os.path.join(build.CSRC, "gv_tune_builtin.h")).read()      # (the head of the shipped file; --out may not exist yet)
head = src[:src.index("static const char* const GV_BUILTIN_FOR_HASH")]
body = 'static const char* const GV_BUILTIN_FOR_HASH = "%s";\nstatic const BuiltinPick GV_BUILTIN_PICKS[] = {\n' % h
for N, M, lay, picks in rows:
    ds = ", ".join("{%d, %d, %d, %d, %.2ff, %.2ff, %d}" % (ks, skl, piv, prio, taper, geo, occ) for ks, skl, piv, prio, taper, geo, occ in picks)
    body += "    {%d, %d, %d, {%s}},\n" % (N, M, lay, ds)
body += "};\n}  // namespace gvi\n"
open(a.out, "w").write(head + body)
print("wrote", a.out, "for kernel sources", h)
