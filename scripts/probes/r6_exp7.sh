O=gpurun_out/r6_dist; mkdir -p $O
export GV_TUNE_CACHE=0
GV_DBG_LIB=$PWD/gpurun_wgtime_libgvamp.so python3 scripts/wgtime.py 400000 125000 ax2 2>&1 | grep -E "block index mod 8"
python scripts/launch_dist.py 400000 1000000 atx --launches 8 --rounds 2 tuned ks=4,geo=0.5,prio=1,skew=0.025 ks=8,geo=0.65,prio=1,skew=0.025 ks=6,geo=0.6,prio=1,skew=0.025 ks=3,geo=0.5,prio=1,skew=0.025 ks=8,geo=0.65,prio=1 > $O/head_atx_skew.txt 2>&1; tail -7 $O/head_atx_skew.txt
python scripts/launch_dist.py 400000 1000000 ax --launches 8 --rounds 2 tuned ks=8,geo=0.65,prio=1,skew=0.02 ks=8,geo=0.65,prio=1,skew=0.03 ks=6,geo=0.6,prio=1,skew=0.025 ks=4,geo=0.5,prio=1,skew=0.025 > $O/head_ax_skew3.txt 2>&1; tail -6 $O/head_ax_skew3.txt
