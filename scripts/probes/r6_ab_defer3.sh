export GV_TUNE_CACHE_DIR=$PWD/gpurun_out/r6_ab/tune3
shape="50000 200000 60 4 1"
python scripts/iter_time.py $shape > /dev/null 2>&1
for rep in 1 2 3 4 5 6 7 8; do
  for leg in nodefer bound unbounded; do
    unset GV_NO_DEFER GV_DEFER_ELEMS
    [ $leg = nodefer ] && export GV_NO_DEFER=1
    [ $leg = unbounded ] && export GV_DEFER_ELEMS=1000000000
    echo "$leg $(python scripts/iter_time.py $shape 2>/dev/null | tail -1 | cut -c1-70)"
  done
done
rm -rf gpurun_out/r6_ab/tune3
