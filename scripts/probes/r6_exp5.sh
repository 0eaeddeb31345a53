O=gpurun_out/r6_dist; mkdir -p $O
export GV_TUNE_CACHE=0
python scripts/launch_dist.py 400000 1000000 ax --launches 8 --rounds 2 tuned ks=8,geo=0.65,prio=1,skew=0.02 ks=8,geo=0.65,prio=1,skew=0.03 ks=8,geo=0.65,prio=1,skew=0.045 > $O/head_ax_skew.txt 2>&1; tail -5 $O/head_ax_skew.txt
python scripts/launch_dist.py 400000 125000 ax2 --launches 25 --rounds 2 tuned ks=6,geo=0.6,prio=1 ks=6,geo=0.6,prio=1,skew=0.02 ks=6,geo=0.6,prio=1,skew=0.035 ks=6,geo=0.6,prio=1,occ=2,skew=0.02 > $O/shard_ax2_skew.txt 2>&1; tail -6 $O/shard_ax2_skew.txt
python scripts/launch_dist.py 400000 125000 atx2 --launches 25 --rounds 2 tuned ks=2,geo=0.5,prio=1,skew=0.02 ks=2,geo=0.5,prio=1,skew=0.035 > $O/shard_atx2_skew.txt 2>&1; tail -4 $O/shard_atx2_skew.txt
python scripts/launch_dist.py 100000 500000 ax2 --launches 25 --rounds 2 tuned ks=5,prio=1 ks=5,prio=1,skew=0.02 ks=5,prio=1,skew=0.035 > $O/cfg2_ax2_skew.txt 2>&1; tail -5 $O/cfg2_ax2_skew.txt
python -m pytest tests/test_gpu_tile.py tests/test_gpu_matvec.py -x -q 2>&1 | tail -3
