cd $GRAFT_REPO_ROOT
for dev in 1 0; do
  export GV_CG_DEVICE=$dev
  echo "=== GV_CG_DEVICE=$dev"
  python scripts/bench_rows.py 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('  ', d['row'][:50], d.get('iters_per_s'), d.get('pass_GBps'), d.get('seconds_per_iter'))"
  python scripts/trace_run.py 400000 125000 6 2 0
  python bench.py --no-cpu-baseline --steps 4 --warmup 2 --ld-block 0 --no-tile-leg 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); v=d['vamp']; print('  headline vamp', v['iters_per_s'], v['seconds_per_iter'], v['n_ax_pass'], v['n_atx_pass'])"
done
