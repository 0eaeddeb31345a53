# round 4, call 7: scheduling variants of the tile-layout ATx step (headline shard), tile statistics kernel, config-5 timeline after the epilogue change
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tc
P="python3 scripts/perf_probe.py --N 400000 --M 1000000 --mode 1 --stripes-only 2 --reps 6"
$P > $O/warm.txt 2>&1     # (measures and caches the decompositions of the tile layout)
for rep in 1 2 3; do
  for v in base ts1 ts2; do
    if [ $v = base ]; then unset GV_DBG_LIB; else export GV_DBG_LIB=$GRAFT_REPO_ROOT/gpurun_${v}_libgvamp.so; fi
    echo "== $v rep $rep" >> $O/tile_sched.txt
    $P 2>&1 | grep -E "^(Ax|ATx)" >> $O/tile_sched.txt
  done
done
unset GV_DBG_LIB
cat $O/tile_sched.txt
python3 scripts/stats_rate.py 400000 1000000 2 > $O/stats_tile.txt 2>&1; cat $O/stats_tile.txt
python3 -m pytest tests/test_gpu_tile.py tests/test_gpu_xxt.py tests/test_gpu_cgdevice.py tests/test_gpu_dual.py -x -q -m gpu > $O/pytest_sub.log 2>&1; tail -3 $O/pytest_sub.log
bash scripts/cfg5_trace.sh r4g > $O/cfg5_trace.log 2>&1; head -12 $O/cfg5_gaps.txt
echo done
