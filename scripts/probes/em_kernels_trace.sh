# development: kernel durations of gv_denoise / gv_prior_estep (scripts/probes/em_kernels.py under rocprofv3 --kernel-trace)
O=gpurun_out/r6_em; mkdir -p $O; rm -rf $O/trace
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 scripts/probes/em_kernels.py > $O/trace.out 2> $O/trace.err || { tail -5 $O/trace.err; exit 1; }
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" > $O/${1:-new}_kernels.txt <<'PY'
import csv, sys, collections, statistics
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    nm = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    d[(nm.split("(")[0], r.get("Grid_Size_X", r.get("Grid_Size", "")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    if any(w in k[0] for w in ("estep", "denoise", "finalize")):
        print("%-42s grid %8s  n %4d  median %6.1f us" % (k[0], k[1], len(v), statistics.median(v)))
PY
rm -rf $O/trace; cat $O/${1:-new}_kernels.txt
