# Kernel traces + gap summaries of whole VAMP iterations at the small-shard shapes (development; run under gpurun).
#   gpurun -- bash scripts/probes/r2_trace_small.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r2}
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
run() {  # name N M iters fuse xxt
  rocprofv3 --kernel-trace --output-format csv -d $O/$1 -o t -- python3 scripts/trace_run.py $2 $3 $4 $5 $6 > $O/$1.out 2>$O/$1.err
  f=$(find $O/$1 -name "*kernel_trace.csv" | head -1)
  python3 scripts/trace_gaps.py $f -60 > $O/$1_gaps.txt 2>&1
  cp $f $O/$1_kernel_trace.csv; rm -rf $O/$1
}
run cfg5 50000 200000 5 4 1
run shard125k 400000 125000 5 4 0
run cfg2 100000 500000 5 4 0
ls -la $O
