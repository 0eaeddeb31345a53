O=gpurun_out/r6_dist; mkdir -p $O
export GV_TUNE_CACHE=0
GV_DBG_LIB=$PWD/gpurun_wgtime_libgvamp.so python3 scripts/wgtime.py 400000 125000 ax2 2>&1 | grep -E "XCC b mod 8|xcc [0-7]:"
python scripts/launch_dist.py 400000 1000000 ax --launches 8 --rounds 2 tuned ks=8,geo=0.65,prio=1,skew=-0.015 ks=8,geo=0.65,prio=1,skew=-0.025 ks=8,geo=0.65,prio=1,skew=-0.04 > $O/head_ax_skew2.txt 2>&1; tail -5 $O/head_ax_skew2.txt
python scripts/launch_dist.py 400000 125000 ax2 --launches 25 --rounds 2 tuned ks=6,geo=0.6,prio=1 ks=6,geo=0.6,prio=1,skew=-0.015 ks=6,geo=0.6,prio=1,skew=-0.03 > $O/shard_ax2_skew2.txt 2>&1; tail -5 $O/shard_ax2_skew2.txt
python scripts/launch_dist.py 400000 125000 atx2 --launches 25 --rounds 2 tuned ks=2,geo=0.5,prio=1,skew=-0.015 ks=2,geo=0.5,prio=1,skew=-0.03 > $O/shard_atx2_skew2.txt 2>&1; tail -4 $O/shard_atx2_skew2.txt
python scripts/launch_dist.py 100000 500000 ax2 --launches 25 --rounds 2 tuned ks=5,prio=1 ks=5,prio=1,skew=-0.015 ks=5,prio=1,skew=-0.03 > $O/cfg2_ax2_skew2.txt 2>&1; tail -5 $O/cfg2_ax2_skew2.txt
