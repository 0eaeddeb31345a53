O=gpurun_out/r6_ab; mkdir -p $O
export GV_TUNE_CACHE_DIR=$PWD/$O/tune
python bench.py --rows-only > /dev/null 2>&1      # fills the tune cache: every leg below runs on the same picks
for rep in 1 2 3; do
  for leg in defer nodefer; do
    if [ $leg = nodefer ]; then export GV_NO_DEFER=1; else unset GV_NO_DEFER; fi
    python bench.py --rows-only > $O/rows_${leg}_$rep.json 2>/dev/null
    python -c "
import json
d=json.load(open('$O/rows_${leg}_$rep.json'))
print('$leg $rep', ' '.join('%s %.2f it/s %.4f' % (r['row'], r['fuse_4']['iters_per_s'], r['fuse_4']['frac']) for r in d['rows']))
"
  done
done
rm -rf $O/tune
