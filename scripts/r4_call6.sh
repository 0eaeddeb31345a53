# round 4, call 6: solver bit-identity tests on the new epilogues / copy kernel, config-5 and config-2 timelines, reader threads of the file ingest, rows
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4f; mkdir -p $O
python3 -m pytest tests/test_gpu_matvec.py tests/test_gpu_tile.py tests/test_gpu_xxt.py tests/test_gpu_cgdevice.py tests/test_gpu_dual.py tests/test_gpu_warm.py tests/test_gpu_fuzz.py tests/test_gpu_pvals.py tests/test_gpu_hardening.py -x -q -m gpu > $O/pytest_sub.log 2>&1; tail -4 $O/pytest_sub.log
export GV_TUNE_CACHE_DIR=$PWD/$O/tc
bash scripts/cfg5_trace.sh r4f > $O/cfg5_trace.log 2>&1; head -36 $O/cfg5_gaps.txt
rocprofv3 --kernel-trace --output-format csv -d $O/cfg2 -o t -- python3 scripts/trace_run.py 100000 500000 5 4 0 > $O/cfg2.out 2>$O/cfg2.err
f=$(find $O/cfg2 -name "*kernel_trace.csv" | head -1); python3 scripts/trace_gaps.py $f -30 > $O/cfg2_gaps.txt 2>&1; rm -rf $O/cfg2; head -12 $O/cfg2_gaps.txt
GV_IO_THREADS=8 python3 scripts/ingest_rate.py > $O/ingest_io8.json 2>$O/ingest_io8.err; cat $O/ingest_io8.json
GV_IO_THREADS=12 python3 scripts/ingest_rate.py > $O/ingest_io12.json 2>$O/ingest_io12.err; cat $O/ingest_io12.json
python3 scripts/bench_rows.py > $O/rows.json 2>$O/rows.err; cat $O/rows.json
echo done
