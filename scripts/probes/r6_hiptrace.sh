cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_hiptrace; rm -rf $O; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tune_cache
python3 scripts/trace_run.py 50000 200000 2 4 1 > /dev/null 2>&1
rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/cfg5 -o t -- python3 scripts/trace_run.py 50000 200000 6 4 1 > $O/cfg5.out 2>$O/cfg5.err
ls $O/cfg5/* | head; 
f=$(find $O/cfg5 -name "*hip_api_trace.csv" | head -1); k=$(find $O/cfg5 -name "*kernel_trace.csv" | head -1)
cp $f $O/hip_api.csv; cp $k $O/kernel.csv; rm -rf $O/cfg5 $O/tune_cache
head -3 $O/hip_api.csv
