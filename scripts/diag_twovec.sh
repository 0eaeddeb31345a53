# Development diagnostic: why are the two-vector streaming kernels slower per byte than the one-vector ones on 12.5 GB shards?
# Run as `gpurun -- bash scripts/diag_twovec.sh [N M]`; everything lands in gpurun_out/diag2v/.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=${1:-400000}; M=${2:-125000}
O=gpurun_out/diag2v; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tune_cache_${N}_${M}
GV_AUTOTUNE_VERBOSE=1 python3 scripts/perf_probe.py --N $N --M $M --mode 1 --stripes-only 1 --reps 10 > $O/probe_${N}_${M}.log 2>&1 || exit 1
P="python3 scripts/perf_probe.py --N $N --M $M --mode 1 --stripes-only 1 --reps 4"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/pmc_l2 -o p -- $P > $O/pmc_l2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- $P > $O/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq -o p -- $P > $O/pmc_sq.log 2>&1 || exit 1
python3 - $O <<'EOF' > $O/summary_${N}_${M}.txt
import csv, glob, sys
from collections import defaultdict
O = sys.argv[1]
for d in ("pmc_l2", "pmc_fetch", "pmc_sq"):
    for f in glob.glob(O + "/" + d + "/**/*counter_collection.csv", recursive=True):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_mfma" not in k:
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            for cn, v in sorted(acc[k].items()):
                top = max(v)
                full = [x for x in v if x >= 0.5 * top] or v
                print("%-44s %-28s n=%3d avg=%.4g" % (k[:44], cn, len(full), sum(full) / len(full)))
EOF
rm -rf $O/pmc_l2 $O/pmc_fetch $O/pmc_sq
cat $O/summary_${N}_${M}.txt
