# Regenerates profiles/r2_* on a GPU box (development; run as `gpurun -- bash scripts/refresh_profiles.sh`).
# Outputs go to gpurun_out/prof_refresh/ ; copy them into profiles/ afterwards.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
R=${1:-r2}
O=gpurun_out/prof_refresh; rm -rf $O; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tune_cache        # the first run measures, the profiled ones read the picks back
python3 bench.py 2>$O/bench_n1.err | tail -1 > $O/${R}_bench_n1.json
# kernel statistics of the timed region's kernels: the matvec legs only (--vamp-iterations 0) -- inside the VAMP legs a CG step
# enqueued before the host knew that both solves had converged returns at once on the device, and those ~3 us launches
# would pull the per-kernel averages down; the whole-run statistics are kept beside them (..._fullrun_kernel_stats.csv)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --vamp-iterations 0 2>$O/stats.err | tail -1 > $O/${R}_bench_under_rocprof.json
cp $O/stats/bench_kernel_stats.csv $O/${R}_bench_kernel_stats.csv 2>/dev/null || cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/${R}_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>$O/stats2.err
cp $O/stats2/bench_kernel_stats.csv $O/${R}_bench_fullrun_kernel_stats.csv 2>/dev/null || cp $(ls $O/stats2/*/*kernel_stats.csv | head -1) $O/${R}_bench_fullrun_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --vamp-iterations 0 > /dev/null 2>$O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --vamp-iterations 0 > /dev/null 2>$O/pmc_write.err
for d in pmc_fetch pmc_write; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && [ "$f" != "$O/$d/bench_counter_collection.csv" ] && cp $f $O/$d/bench_counter_collection.csv; done
python3 scripts/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/$R > /dev/null
python3 scripts/bench_rows.py > $O/${R}_rows.json 2>$O/rows.err
python3 scripts/ingest_rate.py > $O/${R}_ingest.json 2>$O/ingest.err
bash scripts/r2_trace_small.sh prof_small > /dev/null 2>&1; cp gpurun_out/prof_small/*_gaps.txt $O/ 2>/dev/null
for f in cfg5 shard125k cfg2; do mv $O/${f}_gaps.txt $O/${R}_${f}_gaps.txt 2>/dev/null; done
rm -rf $O/stats $O/stats2 $O/pmc_fetch $O/pmc_write $O/tune_cache gpurun_out/prof_small     # keep the merge small
ls -la $O
