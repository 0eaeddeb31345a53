# development: whole-op GB/s of uniform (autotuned) vs balanced decompositions; LIBS="main dbg" compares two builds
for shape in "$@"; do
  N=${shape%x*}; M=${shape#*x}
  for lib in ${LIBS:-main}; do
  for sk in ${SKS:-0 768 1536}; do
    unset GV_SK_M GV_SK_N GV_DBG_LIB
    if [ $sk != 0 ]; then export GV_SK_M=$sk GV_SK_N=$sk; fi
    if [ $lib = dbg ]; then export GV_DBG_LIB=$GRAFT_REPO_ROOT/gpurun_dbg_libgvamp.so; fi
    echo "== N=$N M=$M sk=$sk lib=$lib"
    python scripts/perf_probe.py --N $N --M $M --mode 1 --stripes-only 1 --reps 10 2>&1 | grep -E "^(Ax|ATx|Ax2|ATx2) " | sed 's/(x2 vectors) GB.s per pass/GB\/s/' | tr '\n' ';'; echo
  done; done
done
