"""The launch list of ONE steady-state CG step out of a rocprofv3 kernel trace: every kernel between the last two k_cgx_decide
launches whose step streamed (development tool; profiles/r5_forced_multi_gaps.txt).
  python scripts/trace_step.py <kernel_trace.csv> [which]        which: default = the step with the fewest launches (a pure CG step), -1 = the last complete step, -2 the one before ..."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
which = int(sys.argv[2]) if len(sys.argv) > 2 else -1
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))


def short(n):
    k = re.sub(r"\(anonymous namespace\)::", "", n)
    return re.sub(r"^void ", "", k).split("(")[0][:90]


dec = [i for i, e in enumerate(ev) if "k_cgx_decide" in e[2]]
# steps whose passes streamed (a step enqueued after every system had finished is dropped on the device)
steps = []
for a, b in zip(dec[:-1], dec[1:]):
    seg = ev[a + 1:b + 1]
    if any(("k_mfma_matvec" in e[2] or "k_mfma_tile" in e[2]) and e[1] - e[0] > 20000 for e in seg):
        steps.append(seg)
if len(sys.argv) <= 2:      # default: a pure CG step -- the one with the fewest launches (no denoiser / EM section inside)
    seg = min(steps, key=len)
else:
    seg = steps[which]
t0 = seg[0][0]
span = seg[-1][1] - seg[0][0]
print("one CG step: %d launches, span %.1f us" % (len(seg), span / 1e3))
print("%10s %10s %9s  kernel" % ("start us", "dur us", "gap us"))
prev_end = None
tot = {"stream": 0, "small": 0, "idle": 0}
for s, e, n in seg:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    streaming = ("k_mfma_matvec" in n or "k_mfma_tile" in n) and e - s > 20000
    tot["stream" if streaming else "small"] += e - s
    if prev_end is not None and s > prev_end:
        tot["idle"] += s - prev_end
    print("%10.1f %10.1f %9.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, short(n)))
    prev_end = e if prev_end is None else max(prev_end, e)
print("streaming %.1f us (%.1f %%), other launches %.1f us (%.1f %%), idle %.1f us (%.1f %%)" % (
    tot["stream"] / 1e3, 100.0 * tot["stream"] / span, tot["small"] / 1e3, 100.0 * tot["small"] / span, tot["idle"] / 1e3,
    100.0 * tot["idle"] / span))
