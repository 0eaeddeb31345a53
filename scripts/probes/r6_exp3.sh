O=gpurun_out/r6_tune; mkdir -p $O
export GV_TUNE_CACHE=0 GV_AUTOTUNE_VERBOSE=1
for shape in "400000 125000" "100000 500000" "50000 200000"; do
  set -- $shape
  python scripts/launch_dist.py $1 $2 ax2 --launches 20 --rounds 1 tuned > $O/tune_$1x$2.txt 2>&1; grep -- "->" $O/tune_$1x$2.txt; tail -1 $O/tune_$1x$2.txt
  python scripts/launch_dist.py $1 $2 atx2 --launches 20 --rounds 1 tuned 2>/dev/null | tail -1
done
unset GV_AUTOTUNE_VERBOSE
python bench.py --rows-only > $O/rows.json 2> $O/rows.err; python -c "
import json
d=json.load(open('$O/rows.json'))
for k,v in d.items(): print(k, json.dumps(v)[:400])
"
