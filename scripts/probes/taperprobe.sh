# development: tapered uniform splits at one shape.  usage: taperprobe.sh N M "km:kn pairs" "tapers" "prios"
N=$1; M=$2
for prio in ${5:-0 1}; do for taper in ${4:-0 0.5 0.9}; do
  for pair in $3; do
    km=${pair%:*}; kn=${pair#*:}
    echo -n "prio $prio taper $taper ks_m $km ks_n $kn : "
    GV_PRIO=$prio GV_TAPER=$taper GV_KS_M=$km GV_KS_N=$kn python scripts/perf_probe.py --N $N --M $M --mode 1 --stripes-only 1 --reps 8 2>&1 | grep -E "^(Ax|ATx|Ax2|ATx2) " | sed 's/(x2 vectors) GB.s per pass/GB\/s/; s/ GB.s.*//' | tr '\n' ';'; echo
  done
done; done
