# round 4, call 3: kernel trace of the LOO p-value call at the headline shard; baseline gap timelines of config 5 / config 2 on this round's tree
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4c; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tc
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pv -o t -- python3 scripts/bench_rows.py p-values > $O/pv.out 2>$O/pv.err
f=$(find $O/pv -name "*kernel_stats.csv" | head -1); cp $f $O/pv_kernel_stats.csv; head -20 $O/pv_kernel_stats.csv; cat $O/pv.out
rm -rf $O/pv
bash scripts/cfg5_trace.sh r4c > $O/cfg5_trace.log 2>&1; head -40 $O/cfg5_gaps.txt
rocprofv3 --kernel-trace --output-format csv -d $O/cfg2 -o t -- python3 scripts/trace_run.py 100000 500000 5 4 0 > $O/cfg2.out 2>$O/cfg2.err
f=$(find $O/cfg2 -name "*kernel_trace.csv" | head -1); python3 scripts/trace_gaps.py $f -30 > $O/cfg2_gaps.txt 2>&1; rm -rf $O/cfg2; head -40 $O/cfg2_gaps.txt
echo done
