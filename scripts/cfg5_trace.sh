# development: kernel timeline of config 5 (N=50k x M=200k, --use-XXT-denoiser 1, fuse 4) -> gpurun_out/$1/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-cfg5t}; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/cfg5 -o t -- python3 scripts/trace_run.py 50000 200000 6 4 1 > $O/cfg5.out 2>$O/cfg5.err
f=$(find $O/cfg5 -name "*kernel_trace.csv" | head -1); cp $f $O/cfg5_kernel_trace.csv
python3 scripts/trace_gaps.py $f -27 > $O/cfg5_gaps.txt 2>&1; head -14 $O/cfg5_gaps.txt; cat $O/cfg5.out
rm -rf $O/cfg5
