"""Timeline summary of a rocprofv3 kernel trace: from the first to the last streaming-kernel launch, the time inside each
kernel and the idle time between kernels (development tool).   python scripts/trace_gaps.py <kernel_trace.csv> [skip | -last]"""
import csv
import re
import collections
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0     # streaming launches to skip (autotune, sim_phen, iteration 1)
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# streaming launches that streamed: a pass enqueued after every CG system had finished returns at once on the device (the GO
# instantiations) and is not a pass
idx = [i for i, e in enumerate(ev) if ("k_mfma_matvec" in e[2] or "k_mfma_tile" in e[2]) and e[1] - e[0] > 20000]
lo, hi = idx[skip if skip >= 0 else max(0, len(idx) + skip)], idx[-1]     # skip < 0: only the last -skip streaming launches
seg = ev[lo:hi + 1]
span = seg[-1][1] - seg[0][0]
by = collections.defaultdict(lambda: [0, 0])
busy_end, idle, gaps = seg[0][0], 0, []
t_stream = t_small = n_small = 0
for s, e, n in seg:
    k = re.sub(r"\(anonymous namespace\)::", "", n)
    k = re.sub(r"^void ", "", k).split("(")[0][:70]
    streaming = "k_mfma_matvec" in n or "k_mfma_tile" in n
    if streaming and e - s <= 20000:
        k += "  [dropped on the device: enqueued after every system had finished]"
    if streaming and e - s > 20000:
        t_stream += e - s
    else:
        t_small += e - s
        n_small += 1
    by[k][0] += e - s
    by[k][1] += 1
    if s > busy_end:
        idle += s - busy_end
        gaps.append(s - busy_end)
    busy_end = max(busy_end, e)
nstream = sum(1 for e in seg if ("k_mfma_matvec" in e[2] or "k_mfma_tile" in e[2]) and e[1] - e[0] > 20000)
print("span %.3f ms, %d streaming launches, idle %.3f ms (%.1f %%)" % (span / 1e6, nstream, idle / 1e6, 100.0 * idle / span))
print("streaming kernels %.3f ms (%.1f %%), %d other launches %.3f ms (%.1f %%), idle %.1f %%" % (
    t_stream / 1e6, 100.0 * t_stream / span, n_small, t_small / 1e6, 100.0 * t_small / span, 100.0 * idle / span))
for k, (t, c) in sorted(by.items(), key=lambda kv: -kv[1][0]):
    print("%8.3f ms %5.1f %% %6d x %7.1f us  %s" % (t / 1e6, 100.0 * t / span, c, t / c / 1e3, k))
gaps.sort()
if gaps:
    print("gaps: n=%d median %.1f us p90 %.1f us max %.1f us" % (len(gaps), gaps[len(gaps) // 2] / 1e3,
                                                                   gaps[int(len(gaps) * 0.9)] / 1e3, gaps[-1] / 1e3))
