# round 4, call 4: fused p-value pass (correctness, headline time) and prefetch depth of the statistics kernel
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d; mkdir -p $O
python3 -m pytest tests/test_gpu_pvals.py -x -q > $O/pytest_pvals.log 2>&1; tail -5 $O/pytest_pvals.log
python3 scripts/fuzz_pvals.py 60 > $O/fuzz_pvals.log 2>&1; tail -3 $O/fuzz_pvals.log
for u in 2 3 4; do GV_STATS_UNROLL=$u python3 scripts/stats_rate.py 400000 1000000 1; done > $O/stats.txt 2>&1; cat $O/stats.txt
python3 scripts/stats_rate.py 400000 1000000 2 >> $O/stats.txt 2>&1; tail -1 $O/stats.txt
python3 scripts/bench_rows.py p-values > $O/pv_row.json 2>$O/pv_row.err; cat $O/pv_row.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pv -o t -- python3 scripts/bench_rows.py p-values > $O/pv.out 2>$O/pv.err
f=$(find $O/pv -name "*kernel_stats.csv" | head -1); cp $f $O/pv_kernel_stats.csv; rm -rf $O/pv
grep -E "pvals|prep_pv|k_quant|matvec<2" $O/pv_kernel_stats.csv | cut -c1-60,200-400
echo done
