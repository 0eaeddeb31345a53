O=gpurun_out/r6_dist; mkdir -p $O
export GV_TUNE_CACHE=0
for pad in 0 60000; do
  for cls in ax ax2 atx atx2; do
    GV_LDS_PAD=$pad python scripts/launch_dist.py 400000 1000000 $cls --launches 12 --rounds 1 tuned > $O/head_${cls}_pad$pad.txt 2>&1; tail -2 $O/head_${cls}_pad$pad.txt
  done
done
for pad in 0 60000; do
  for cls in ax2 atx2; do
    GV_LDS_PAD=$pad python scripts/launch_dist.py 50000 200000 $cls --launches 30 --rounds 1 tuned ks=8,geo=0.8,prio=1 ks=4,geo=0.5,prio=1 ks=1,prio=1 > $O/cfg5_${cls}_pad$pad.txt 2>&1; tail -5 $O/cfg5_${cls}_pad$pad.txt
  done
done
GV_LDS_PAD=60000 python scripts/launch_dist.py 100000 500000 ax2 --launches 30 --rounds 1 tuned ks=5,prio=1 ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 > $O/cfg2_ax2_pad.txt 2>&1; tail -5 $O/cfg2_ax2_pad.txt
GV_LDS_PAD=60000 python scripts/launch_dist.py 100000 500000 atx2 --launches 30 --rounds 1 tuned ks=1,prio=1 ks=2,geo=0.5,prio=1 > $O/cfg2_atx2_pad.txt 2>&1; tail -4 $O/cfg2_atx2_pad.txt
