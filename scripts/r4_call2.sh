# round 4, call 2: in-process A/B of geometric (big-first) uniform splits against the tuner's picks
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4b; mkdir -p $O
export GV_TUNE_CACHE=0
A="python3 scripts/ab_decomp.py"
$A 400000 125000 atx2 tuned ks=1,prio=1 ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 ks=4,geo=0.5 ks=3,geo=0.4,prio=1 ks=5,geo=0.5,prio=1 cells=995,prio=1 > $O/ab_400k_atx2.txt 2>&1; cat $O/ab_400k_atx2.txt
$A 400000 125000 atx tuned ks=1,prio=1 ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 ks=3,geo=0.4,prio=1 > $O/ab_400k_atx.txt 2>&1; cat $O/ab_400k_atx.txt
$A 400000 125000 ax2 tuned ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 ks=3,geo=0.4,prio=1 ks=2,geo=0.4,prio=1 > $O/ab_400k_ax2.txt 2>&1; cat $O/ab_400k_ax2.txt
$A 400000 125000 ax tuned ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 ks=3,geo=0.4,prio=1 > $O/ab_400k_ax.txt 2>&1; cat $O/ab_400k_ax.txt
$A 100000 500000 atx2 tuned ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=3,geo=0.5,prio=1 ks=2,geo=0.5,prio=1 ks=4,geo=0.5 ks=1,prio=1 > $O/ab_100k_atx2.txt 2>&1; cat $O/ab_100k_atx2.txt
$A 100000 500000 atx tuned ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=3,geo=0.5,prio=1 ks=2,geo=0.5,prio=1 > $O/ab_100k_atx.txt 2>&1; cat $O/ab_100k_atx.txt
$A 100000 500000 ax2 tuned ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 ks=12,geo=0.75,prio=1 > $O/ab_100k_ax2.txt 2>&1; cat $O/ab_100k_ax2.txt
$A 100000 500000 ax tuned ks=4,geo=0.5,prio=1 ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 ks=12,geo=0.75,prio=1 > $O/ab_100k_ax.txt 2>&1; cat $O/ab_100k_ax.txt
$A 50000 200000 atx2 --reps 12 tuned ks=4,geo=0.5,prio=1 ks=3,geo=0.5,prio=1 ks=2,geo=0.4,prio=1 ks=1,prio=1 > $O/ab_50k_atx2.txt 2>&1; cat $O/ab_50k_atx2.txt
$A 50000 200000 ax2 --reps 12 tuned ks=6,geo=0.6,prio=1 ks=8,geo=0.65,prio=1 ks=4,geo=0.5,prio=1 ks=8,geo=0.8,prio=1 > $O/ab_50k_ax2.txt 2>&1; cat $O/ab_50k_ax2.txt
echo done
