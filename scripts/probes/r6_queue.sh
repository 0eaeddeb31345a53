# development: the queue prototype (docs/history/attic/queue_prototype.patch applied, then `bash scripts/build_variant.sh queue`) beside the plain launches
set -e
export GV_DBG_LIB=$PWD/gpurun_queue_libgvamp.so
export GV_PARTIAL_KS=24
O=gpurun_out/r6_queue; mkdir -p $O
timeout -k 10 300 python3 scripts/launch_dist.py 400000 125000 ax2 --launches 30 tuned ks=8,geo=0.65,prio=1 ks=8,geo=0.65,prio=1,queue=1 ks=12,geo=0.75,prio=1,queue=1 ks=16,geo=0.8,prio=1,queue=1 ks=8,taper=0.5,prio=1,queue=1 ks=8,geo=0.65,prio=0,queue=1 > $O/shard_ax2.txt 2>&1 &&
timeout -k 10 300 python3 scripts/launch_dist.py 400000 125000 atx2 --launches 30 tuned ks=2,geo=0.5,prio=1 ks=4,geo=0.6,prio=1,queue=1 ks=6,geo=0.7,prio=1,queue=1 ks=8,geo=0.75,prio=1,queue=1 ks=4,geo=0.6,prio=0,queue=1 > $O/shard_atx2.txt 2>&1
tail -n 12 $O/shard_ax2.txt; tail -n 12 $O/shard_atx2.txt
