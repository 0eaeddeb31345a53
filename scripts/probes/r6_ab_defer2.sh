export GV_TUNE_CACHE_DIR=$PWD/gpurun_out/r6_ab/tune2
for shape in "50000 200000 40 4 1" "400000 125000 25 4 0" "100000 500000 20 4 0"; do
  python scripts/iter_time.py $shape > /dev/null 2>&1
  for rep in 1 2 3 4; do
    for leg in defer nodefer; do
      if [ $leg = nodefer ]; then export GV_NO_DEFER=1; else unset GV_NO_DEFER; fi
      echo "$shape $leg: $(python scripts/iter_time.py $shape 2>/dev/null | tail -1)"
    done
  done
done
rm -rf gpurun_out/r6_ab/tune2
