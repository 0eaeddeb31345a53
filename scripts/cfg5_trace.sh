cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5e; mkdir -p $O
GV_LAYOUT=2 rocprofv3 --kernel-trace --output-format csv -d $O/cfg5 -o t -- python3 scripts/trace_run.py 50000 200000 6 4 1 > $O/cfg5.out 2>$O/cfg5.err
f=$(find $O/cfg5 -name "*kernel_trace.csv" | head -1); cp $f $O/cfg5_kernel_trace.csv
python3 scripts/trace_gaps.py $f -27 > $O/cfg5_gaps.txt 2>&1; head -14 $O/cfg5_gaps.txt; cat $O/cfg5.out
python3 scripts/trace_tail.py $f 130 > $O/cfg5_tail.txt
rm -rf $O/cfg5
python3 scripts/trace_phases.py $O/cfg5_kernel_trace.csv 4 > $O/cfg5_phases.txt; cat $O/cfg5_phases.txt
