# development: per-workgroup clocks of the two-vector launches of the mid-size rows on the engine that ships (layout auto -> tile): round 5's
# picks pinned beside what the library picks now.  Needs `bash scripts/build_variant.sh wgtime -DGV_WGTIME` first; writes gpurun_out/r6_wgtime/*.txt
O=gpurun_out/r6_wgtime; mkdir -p $O
export GV_DBG_LIB=$PWD/gpurun_wgtime_libgvamp.so
run() { python3 scripts/wgtime.py "$@" > $O/tmp.txt 2>&1 || { tail -5 $O/tmp.txt; exit 1; }; cat $O/tmp.txt; echo; }
{
echo "## the 8-GPU shard shape, N=400k x M=125k"
run 400000 125000 atx2 0 ks=1,prio=1
run 400000 125000 atx2
run 400000 125000 ax2 0 ks=8,geo=0.65,prio=1
run 400000 125000 ax2
echo "## config 2, N=100k x M=500k"
run 100000 500000 atx2
run 100000 500000 ax2 0 ks=5
run 100000 500000 ax2
echo "## config 5, N=50k x M=200k"
run 50000 200000 atx2 0 cells=8,whole=768,prio=1
run 50000 200000 atx2
run 50000 200000 ax2
} > $O/all.txt
rm -f $O/tmp.txt
