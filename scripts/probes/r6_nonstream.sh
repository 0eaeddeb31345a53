# development: the non-streaming launches of one iteration, with gaps (scripts/trace_nonstream.py), shard shape and config 5
export TMPDIR=/tmp
O=gpurun_out/r6_nonstream; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/sh -o t -- python3 scripts/trace_run.py 400000 125000 8 4 0 > $O/sh.out 2>$O/sh.err || exit 1
python3 scripts/trace_nonstream.py $(find $O/sh -name "*kernel_trace.csv" | head -1) 2 > $O/shard.txt; rm -rf $O/sh
rocprofv3 --kernel-trace --output-format csv -d $O/c5 -o t -- python3 scripts/trace_run.py 50000 200000 8 4 1 > $O/c5.out 2>$O/c5.err || exit 1
python3 scripts/trace_nonstream.py $(find $O/c5 -name "*kernel_trace.csv" | head -1) 2 > $O/cfg5.txt; rm -rf $O/c5
wc -l $O/*.txt
