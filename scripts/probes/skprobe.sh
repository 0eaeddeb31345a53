# development: kernel-level durations of the uniform vs balanced decomposition (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
N=$1; M=$2; shift; shift
for sk in "$@"; do
  if [ $sk = 0 ]; then unset GV_SK_M GV_SK_N; else export GV_SK_M=$sk GV_SK_N=$sk; fi
  rm -rf gpurun_out/trace_sk
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_sk -- python3 scripts/perf_probe.py --N $N --M $M --mode 1 --stripes-only 1 --reps 10 > /dev/null 2>&1
  echo "== N=$N M=$M sk=$sk"
  f=$(ls gpurun_out/trace_sk/*/*kernel_trace.csv | head -1)
  python3 scripts/trace_gaps.py $f 14 | grep -E "span|k_mfma|k_fin|k_prep|k_quant"
done
rm -rf gpurun_out/trace_sk
