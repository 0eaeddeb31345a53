# round 4, call 1: grid-barrier cost; per-workgroup clocks, decomposition sweep and SQ counters of the two-vector ATx on 12.5 GB shards
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4a; mkdir -p $O
echo skip barrier
export GV_DBG_LIB=$GRAFT_REPO_ROOT/gpurun_dbg_libgvamp.so
for shape in "100000 500000" "400000 125000"; do
  set -- $shape
  for w in atx2 atx ax2; do
    python3 scripts/wgtime.py $1 $2 $w > $O/wg_${1}_${2}_$w.txt 2>&1 || { tail -5 $O/wg_${1}_${2}_$w.txt; exit 1; }
  done
done
unset GV_DBG_LIB
echo "wgtime done"
for shape in "100000 500000" "400000 125000"; do
  set -- $shape
  for cfg in "" "GV_KS_M=1 GV_PRIO=0" "GV_KS_M=1 GV_PRIO=1" "GV_KS_M=2 GV_PRIO=1" "GV_KS_M=2 GV_PRIO=1 GV_TAPER=0.5" "GV_KS_M=3 GV_PRIO=1 GV_TAPER=0.5" "GV_KS_M=3 GV_PRIO=0" \
             "GV_SK_M=768" "GV_SK_M=1536" "GV_SK_M=2304" "GV_HY_M=0:768" "GV_HY_M=0:1536" "GV_HY_M=256:768" "GV_HY_M=384:768" "GV_HY_M=256:512"; do
    echo "== $1 $2 [$cfg]" >> $O/sweep.txt
    env $cfg python3 scripts/perf_probe.py --N $1 --M $2 --mode 1 --stripes-only 1 --reps 10 2>&1 | grep -E "^(Ax|ATx)" >> $O/sweep.txt
  done
done
cat $O/sweep.txt
bash scripts/diag_twovec.sh 100000 500000 > $O/diag_100k.log 2>&1
bash scripts/diag_twovec.sh 400000 125000 > $O/diag_400k.log 2>&1
cp gpurun_out/diag2v/summary_*.txt $O/ 2>/dev/null
echo done
