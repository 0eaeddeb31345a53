O=gpurun_out/r6_dist; mkdir -p $O
export GV_TUNE_CACHE=0
for lib in base occ4; do
  if [ $lib = occ4 ]; then export GV_DBG_LIB=$PWD/gpurun_occ4_libgvamp.so; else unset GV_DBG_LIB; fi
  echo "== $lib"
  python scripts/launch_dist.py 400000 1000000 atx --launches 8 --rounds 2 tuned cells=261,quads=3840,prio=1 cells=1275,quads=3072,prio=1 cells=638,quads=3072,prio=1 cells=320,quads=3584,prio=1 2>&1 | tail -6
  python scripts/launch_dist.py 100000 500000 atx2 --launches 20 --rounds 2 tuned cells=213,quads=1536,prio=1 cells=400,quads=1024,prio=1 cells=712,quads=1024,prio=1 2>&1 | tail -5
done
