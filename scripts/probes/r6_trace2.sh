cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_trace2; rm -rf $O; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tune_cache
run() {
  python3 scripts/trace_run.py $2 $3 2 $5 $6 > /dev/null 2>&1
  rocprofv3 --kernel-trace --output-format csv -d $O/$1 -o t -- python3 scripts/trace_run.py $2 $3 $4 $5 $6 > $O/$1.out 2>$O/$1.err
  f=$(find $O/$1 -name "*kernel_trace.csv" | head -1)
  { python3 scripts/trace_gaps.py $f $7; echo; python3 scripts/trace_phases.py $f 4; echo; python3 scripts/trace_step.py $f; echo; cat $O/$1.out; } > $O/$1_gaps.txt 2>&1
  python3 scripts/trace_tail.py $f $8 > $O/$1_tail.txt
  rm -rf $O/$1
}
run cfg5 50000 200000 6 4 1 -26 110
run shard125k 400000 125000 6 4 0 -10 80
rm -rf $O/tune_cache
