"""Where the non-streaming time of a VAMP iteration goes, from a rocprofv3 kernel trace (development tool): per iteration (cut at the
first k_denoise of each denoising section) the wall time of  A: denoiser / EM section,  B: from there to the first streaming pass of
the solves,  C: the solves minus their streaming kernels (small launches between passes, dropped steps),  D: from the last pass to
the next iteration's first k_denoise.   python scripts/trace_phases.py <kernel_trace.csv> [n_last_iterations]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
is_stream = lambda e: ("k_mfma_matvec" in e[2] or "k_mfma_tile" in e[2]) and e[1] - e[0] > 20000
is_dn = lambda e: "k_denoise" in e[2] or "k_probit_denoise" in e[2]
# iteration starts: a k_denoise not preceded (since the last streaming kernel) by another k_denoise
starts, seen_dn = [], False
for i, e in enumerate(ev):
    if is_stream(e):
        seen_dn = False
    elif is_dn(e) and not seen_dn:
        starts.append(i)
        seen_dn = True
its = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)][-nlast:]
print("%4s %9s %9s | %9s %9s %9s %9s | %7s %7s" % ("it", "span us", "stream", "A dn/EM", "B pre", "C solve-", "D post", "launch", "nonstr%"))
tot = [0.0] * 7
for a, b in its:
    seg = ev[a:b]
    t0, t1 = seg[0][0], ev[b][0]
    sidx = [i for i, e in enumerate(seg) if is_stream(e)]
    first_s, last_s = seg[sidx[0]], seg[sidx[-1]]
    stream = sum(e[1] - e[0] for e in seg if is_stream(e))
    # A ends at the end of the last denoise / E-step / finalize_pub before the first streaming kernel
    a_end = max(e[1] for e in seg[:sidx[0]] if is_dn(e) or "k_prior_estep" in e[2])
    # (the finalize_pub behind it belongs to the section: take the next kernel's end if it is one)
    for e in seg[:sidx[0]]:
        if e[0] >= a_end and "k_finalize_pub" in e[2] and e[0] - a_end < 2000:
            a_end = e[1]
            break
    A = a_end - t0
    B = first_s[0] - a_end
    C = (last_s[1] - first_s[0]) - stream
    D = t1 - last_s[1]
    span = t1 - t0
    vals = [span, stream, A, B, C, D]
    print("%4d %9.1f %9.1f | %9.1f %9.1f %9.1f %9.1f | %7d %6.1f%%" % (len(tot), *[v / 1e3 for v in vals], len(seg), 100.0 * (span - stream) / span))
    for k, v in enumerate(vals):
        tot[k] += v
n = len(its)
print("mean %9.1f %9.1f | %9.1f %9.1f %9.1f %9.1f |         %6.1f%%" % (*[t / n / 1e3 for t in tot[:6]], 100.0 * (tot[0] - tot[1]) / tot[0]))
