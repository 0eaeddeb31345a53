O=gpurun_out/r6_dist; mkdir -p $O
export GV_TUNE_CACHE=0
python scripts/launch_dist.py 400000 125000 ax2 --launches 30 tuned ks=6,geo=0.6,prio=1 ks=3,geo=0.5,prio=1 ks=4,geo=0.5,prio=1 ks=4,geo=0.7,prio=1 ks=5,geo=0.6,prio=1 ks=12,geo=0.8,prio=1 ks=2,geo=0.35,prio=1 > $O/shard_ax2_b.txt 2>&1; cat $O/shard_ax2_b.txt
GV_LDS_PAD=60000 python scripts/launch_dist.py 400000 125000 ax2 --launches 30 tuned ks=6,geo=0.6,prio=1 ks=3,geo=0.5,prio=1 ks=4,geo=0.5,prio=1 ks=2,prio=1 ks=2,geo=0.5,prio=1 > $O/shard_ax2_pad.txt 2>&1; cat $O/shard_ax2_pad.txt
GV_LDS_PAD=60000 python scripts/launch_dist.py 400000 125000 atx2 --launches 30 tuned ks=1,prio=1 ks=2,geo=0.5,prio=1 ks=2,geo=0.35,prio=1 > $O/shard_atx2_pad.txt 2>&1; cat $O/shard_atx2_pad.txt
python scripts/launch_dist.py 100000 500000 ax2 --launches 30 tuned ks=5,prio=1 ks=6,geo=0.6,prio=1 ks=8,prio=1 ks=10,geo=0.8,prio=1 ks=7,geo=0.5,prio=1 > $O/cfg2_ax2.txt 2>&1; cat $O/cfg2_ax2.txt
