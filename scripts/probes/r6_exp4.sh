O=gpurun_out/r6_dist; mkdir -p $O
export GV_TUNE_CACHE=0
python scripts/launch_dist.py 100000 500000 atx2 --launches 25 --rounds 2 tuned ks=1,prio=1 ks=2,taper=0.5,prio=1 ks=3,geo=0.5,prio=1 ks=4,geo=0.5,prio=1 cells=213,quads=1536,prio=1 cells=163,quads=1024,prio=1 cells=498,prio=1 cells=249,prio=1 ks=1,prio=1,occ=2 > $O/cfg2_atx2_b.txt 2>&1; tail -11 $O/cfg2_atx2_b.txt
python scripts/launch_dist.py 50000 200000 atx2 --launches 40 --rounds 2 tuned ks=4,geo=0.5,prio=1 ks=3,geo=0.5,prio=1 ks=5,geo=0.6,prio=1 ks=6,geo=0.6,prio=1 ks=4,geo=0.7,prio=1 ks=2,geo=0.5,prio=1 cells=8,quads=768,prio=1 > $O/cfg5_atx2_b.txt 2>&1; tail -9 $O/cfg5_atx2_b.txt
python scripts/launch_dist.py 50000 200000 ax2 --launches 40 --rounds 2 tuned ks=10,prio=1 ks=12,prio=1 ks=14,prio=1 ks=15,prio=1 ks=16,geo=0.9,prio=1 ks=12,taper=0.5,prio=1 ks=20,prio=1 > $O/cfg5_ax2_b.txt 2>&1; tail -9 $O/cfg5_ax2_b.txt
