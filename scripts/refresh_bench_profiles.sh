cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_refresh2; rm -rf $O; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tune_cache
python3 bench.py 2>$O/bench_n1.err | tail -1 > $O/r2_bench_n1.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --vamp-iterations 0 2>$O/stats.err | tail -1 > $O/r2_bench_under_rocprof.json
cp $O/stats/bench_kernel_stats.csv $O/r2_bench_kernel_stats.csv 2>/dev/null || cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/r2_bench_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats2 -o bench -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > /dev/null 2>$O/stats2.err
cp $O/stats2/bench_kernel_stats.csv $O/r2_bench_fullrun_kernel_stats.csv 2>/dev/null || cp $(ls $O/stats2/*/*kernel_stats.csv | head -1) $O/r2_bench_fullrun_kernel_stats.csv
rm -rf $O/stats $O/stats2 $O/tune_cache
ls -la $O; head -c 600 $O/r2_bench_n1.json
ls -la $O
