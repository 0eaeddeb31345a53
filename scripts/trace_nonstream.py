"""development: every non-streaming launch of ONE VAMP iteration (the n-th from the end, default 2) of a rocprofv3 kernel trace, in order,
with its duration and the idle gap before it, grouped by the phases of scripts/trace_phases.py -- to see which gaps are the host's.
  python scripts/trace_nonstream.py <kernel_trace.csv> [n_from_end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]) for r in rows))
is_stream = lambda e: ("k_mfma_matvec" in e[2] or "k_mfma_tile" in e[2]) and e[1] - e[0] > 20000
is_dn = lambda e: "k_denoise" in e[2] or "k_probit_denoise" in e[2]
starts, seen = [], False
for i, e in enumerate(ev):
    if is_stream(e):
        seen = False
    elif is_dn(e) and not seen:
        starts.append(i)
        seen = True
a, b = starts[-nth - 1], starts[-nth]
seg = ev[a:b]
t0 = seg[0][0]
prev_end = seg[0][0]
tot_gap = tot_small = 0.0
print("iteration of %d launches, span %.1f us" % (len(seg), (ev[b][0] - t0) / 1e3))
for e in seg:
    gap = (e[0] - prev_end) / 1e3
    dur = (e[1] - e[0]) / 1e3
    if is_stream(e):
        print("%9.1f  %8.1f  gap %6.1f  == %s" % ((e[0] - t0) / 1e3, dur, gap, e[2][:50]))
    else:
        print("%9.1f  %8.1f  gap %6.1f  %s" % ((e[0] - t0) / 1e3, dur, gap, e[2][:50]))
        tot_small += dur
    tot_gap += max(gap, 0.0)
    prev_end = max(prev_end, e[1])
print("small launches %.1f us, idle gaps %.1f us" % (tot_small, tot_gap))
