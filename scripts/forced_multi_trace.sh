# development: the forced multi-rank branches at the per-GPU shard of the 8-GPU headline job (N=400k x M=125k) on ONE GPU
# (gv_debug_force_multi: 1-rank RCCL all-reduce followed by the loop-back) -> gpurun_out/$1/ ; see profiles/r<N>_forced_multi_gaps.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-fm}; mkdir -p $O
for leg in plain forced forced_ov4; do
  case $leg in plain) FM=0; OV=0;; forced) FM=3; OV=0;; forced_ov4) FM=3; OV=4;; esac
  GV_LAYOUT=2 GVAMP_FORCE_MULTI=$FM GV_OVERLAP=$OV rocprofv3 --kernel-trace --output-format csv -d $O/$leg -o t -- python3 scripts/trace_run.py 400000 125000 6 4 0 > $O/$leg.out 2>$O/$leg.err || exit 1
  f=$(find $O/$leg -name "*kernel_trace.csv" | head -1); cp $f $O/${leg}_kernel_trace.csv
  { echo "== $leg (GVAMP_FORCE_MULTI=$FM GV_OVERLAP=$OV): iterations 2.."; python3 scripts/trace_gaps.py $f -20; echo; python3 scripts/trace_step.py $f -2; echo; cat $O/$leg.out; } > $O/${leg}_gaps.txt 2>&1
  rm -rf $O/$leg
done
for leg in 0 2 3; do
  GVAMP_FORCE_MULTI=$leg python3 bench.py --N 400000 --Mt 125000 --no-cpu-baseline --no-side-leg --no-rows --ld-block 0 > $O/bench_fm$leg.json 2>$O/bench_fm$leg.err || exit 1
done
GVAMP_FORCE_MULTI=3 GV_OVERLAP=4 python3 bench.py --N 400000 --Mt 125000 --no-cpu-baseline --no-side-leg --no-rows --ld-block 0 > $O/bench_fm3_ov4.json 2>$O/bench_fm3_ov4.err
