# development: A/B of the denoiser / E-step kernels across processes on one box -- the in-tree library swapped between the current build and
# gpurun_oldem_libgvamp.so (the previous gv_kernels.hip linked with the same other objects); bench.py --rows-only, three alternations
O=gpurun_out/r6_em_ab; mkdir -p $O
cp gvamp_amd/libgvamp.so /tmp/new_libgvamp.so
for i in 1 2 3; do
  cp gpurun_oldem_libgvamp.so gvamp_amd/libgvamp.so
  timeout -k 10 200 python3 bench.py --rows-only > $O/old_$i.json 2> $O/old_$i.err || exit 1
  cp /tmp/new_libgvamp.so gvamp_amd/libgvamp.so
  timeout -k 10 200 python3 bench.py --rows-only > $O/new_$i.json 2> $O/new_$i.err || exit 1
  echo "alternation $i done"
done
python3 - <<'PY'
import json, glob
for tag in ("old", "new"):
    for f in sorted(glob.glob("gpurun_out/r6_em_ab/%s_*.json" % tag)):
        rows = json.load(open(f))["rows"]
        print(tag, " ".join("%s %.4f (%.1f it/s)" % (r["row"], r["fuse_4"]["frac"], r["fuse_4"]["iters_per_s"]) for r in rows))
PY
