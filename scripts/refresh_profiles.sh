# Regenerates profiles/<R>_* on a GPU box (development).  Two calls (each fits a 20-minute gpurun call):
#   gpurun --timeout 1200 -- bash scripts/refresh_profiles.sh r4 a     bench line, kernel statistics, PMC traffic
#   gpurun --timeout 1200 -- bash scripts/refresh_profiles.sh r4 b     rows, ingest, kernel timelines, mid-size counter passes
#   gpurun --timeout 600  -- bash scripts/refresh_profiles.sh r4 c     the kernel timelines only
# Outputs go to gpurun_out/prof_refresh_<part>/ ; copy them into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r4}; PART=${2:-a}
O=gpurun_out/prof_refresh_$PART; rm -rf $O; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tune_cache        # shapes outside the shipped table: the first run measures, later ones read back
if [ "$PART" = a ]; then
python3 bench.py 2>$O/bench_n1.err | tail -1 > $O/${R}_bench_n1.json
# kernel statistics of the timed region's kernels: the matvec legs only (--vamp-iterations 0) -- inside the VAMP legs a CG step
# enqueued before the host knew that both solves had converged returns at once on the device, and those ~3 us launches
# would pull the per-kernel averages down; the whole-run statistics are kept beside them (..._fullrun_kernel_stats.csv)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --vamp-iterations 0 --no-rows --no-side-leg 2>$O/stats.err | tail -1 > $O/${R}_bench_under_rocprof.json
cp $O/stats/bench_kernel_stats.csv $O/${R}_bench_kernel_stats.csv 2>/dev/null || cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${R}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>$O/stats2.err
cp $O/stats2/bench_kernel_stats.csv $O/${R}_bench_fullrun_kernel_stats.csv 2>/dev/null || cp $(ls $O/stats2/*/*kernel_stats.csv | head -1) $O/${R}_bench_fullrun_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --vamp-iterations 0 --no-rows > /dev/null 2>$O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --vamp-iterations 0 --no-rows > /dev/null 2>$O/pmc_write.err
for d in pmc_fetch pmc_write; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$O/$d/bench_counter_collection.csv" ] && cp $f $O/$d/bench_counter_collection.csv; done
python3 scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/$R > /dev/null
rm -rf $O/stats $O/stats2 $O/pmc_fetch $O/pmc_write $O/tune_cache
else
if [ "$PART" = b ]; then
python3 bench.py --rows-only > $O/${R}_rows.json 2>$O/rows.err
python3 bench.py --rows-only --rows-layout 1 > $O/${R}_rows_two_stripe_sets.json 2>$O/rows_l1.err
python3 scripts/bench_rows.py "p-values" > $O/${R}_pvals_row.json 2>$O/pvals_row.err
python3 scripts/ingest_rate.py > $O/${R}_ingest.json 2>$O/ingest.err
python3 scripts/hostptr_probe.py > $O/${R}_hostptr_probe.txt 2>&1
fi
run() {  # name N M iters fuse xxt last-streaming-launches
  rocprofv3 --kernel-trace --output-format csv -d $O/$1 -o t -- python3 scripts/trace_run.py $2 $3 $4 $5 $6 > $O/$1.out 2>$O/$1.err
  f=$(find $O/$1 -name "*kernel_trace.csv" | head -1)
  { python3 scripts/trace_gaps.py $f $7; echo; echo "== where the non-streaming time of an iteration goes (scripts/trace_phases.py)"; python3 scripts/trace_phases.py $f 4;
    echo; echo "== one steady-state CG step (scripts/trace_step.py)"; python3 scripts/trace_step.py $f; echo; cat $O/$1.out; } > $O/${R}_$1_gaps.txt 2>&1
  rm -rf $O/$1
}
run cfg5 50000 200000 6 4 1 -26        # the last three iterations (8-9 passes each): steady state, set-up excluded
run shard125k 400000 125000 6 4 0 -10
run cfg2 100000 500000 6 4 0 -18
[ "$PART" = c ] && { ls -la $O; exit 0; }      # part c: the three kernel timelines only
# counter passes of the four streaming-kernel classes on 12.5 GB shards (FETCH_SIZE against the algorithmic bytes)
bash scripts/diag_twovec.sh 400000 125000 > /dev/null 2>&1
bash scripts/diag_twovec.sh 100000 500000 > /dev/null 2>&1
cat gpurun_out/diag2v/summary_400000_125000.txt gpurun_out/diag2v/summary_100000_500000.txt > $O/${R}_midsize_pmc.txt 2>/dev/null
cp gpurun_out/diag2v/probe_*.log $O/ 2>/dev/null
rm -rf $O/tune_cache gpurun_out/diag2v
fi
ls -la $O
