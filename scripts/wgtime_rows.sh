# development: per-workgroup clocks of the two-vector launches of the mid-size rows on the engine that ships (layout auto -> tile).
# needs `bash scripts/build_variant.sh wgtime -DGV_WGTIME` first; writes gpurun_out/r6_wgtime/*.txt
O=gpurun_out/r6_wgtime; mkdir -p $O
export GV_DBG_LIB=$PWD/gpurun_wgtime_libgvamp.so
for shape in "400000 125000" "100000 500000" "50000 200000"; do
  set -- $shape
  for which in ax2 atx2; do
    python3 scripts/wgtime.py $1 $2 $which > $O/wg_$1x$2_$which.txt 2>&1 || { tail -5 $O/wg_$1x$2_$which.txt; exit 1; }
  done
done
tail -n +1 $O/wg_400000x125000_*.txt
