"""Every launch of the tail of a rocprofv3 kernel trace with its duration and the idle gap before it (development tool).
  python scripts/trace_tail.py <kernel_trace.csv> [n_last_launches]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 150
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))[-n:]
t0, prev = ev[0][0], None
for s, e, name in ev:
    k = re.sub(r"\(anonymous namespace\)::", "", name)
    k = re.sub(r"^void ", "", k).split("(")[0][:80]
    print("%10.1f %9.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, k))
    prev = e if prev is None else max(prev, e)
