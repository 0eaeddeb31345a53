#!/bin/bash
# First contact with an 8-GPU MI355X node: everything this build has never been able to run (RCCL with nranks > 1), in the order
# that pays, each leg a fresh process tree and ONE JSON file.  Usage: scripts/first_8gpu.sh [outdir]   (from the repo root)
#
#   1. bench.py --gpus 1 / 2 / 4 / 8        the strong-scaling line of BASELINE.json; every line must carry
#                                            multi_gpu.rccl_nranks == N, and the N = 1 value must sit within 3 % of the last
#                                            committed 1-GPU bench line (profiles/r*_bench_n1.json)
#   2. GV_OVERLAP = 0 / 2 / 4 at 8 ranks     the exchange of data::Ax overlapped with the decode (docs/history/rounds1-3.md section 6): measured, not assumed
#   3. GV_CG_DEVICE = 0 / 1 at 8 ranks       device-resident CG scalars against the host-driven loop (its case is the sharded job)
#   4. gvamp_sim as 8 RCCL ranks             the reference's command line under `mpirun -np 8`, against the files the real
#                                            reference wrote (tests/golden/survey_probe/sim_np8_*)
#   5. the RCCL tests of the suite           tests/test_gpu_multiproc.py (skipped on 1-GPU boxes)
# scripts/first_8gpu_check.py turns the JSON files into one verdict (exit code 0 = every assertion held).
set -u
OUT=${1:-gpurun_out/first8}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python -c "import torch; print(torch.cuda.device_count())")
echo "GPUs visible: $NGPU" | tee "$OUT/node.txt"
(rocm-smi --showtopo 2>/dev/null || true) >> "$OUT/node.txt"

run_bench() {   # name, gpus, extra env..., then bench flags after --
    local name=$1 n=$2; shift 2
    local envs=()
    while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
    [ $# -gt 0 ] && shift
    echo "== $name (gpus $n ${envs[*]:-})"
    if [ "$n" -gt "$NGPU" ]; then echo "{\"skipped\": \"needs $n GPUs, have $NGPU\"}" > "$OUT/$name.json"; return; fi
    env "${envs[@]}" timeout -k 10 1500 python bench.py --gpus "$n" "$@" > "$OUT/$name.json" 2> "$OUT/$name.err" \
        || echo "{\"failed\": \"exit $?\"}" > "$OUT/$name.json"
}

for n in 1 2 4 8; do run_bench "bench_n$n" $n -- ; done
for t in 0 2 4; do run_bench "overlap_$t" 8 GV_OVERLAP=$t -- --no-cpu-baseline --no-tile-leg --ld-block 0; done
for d in 0 1; do run_bench "cgdevice_$d" 8 GV_CG_DEVICE=$d -- --no-cpu-baseline --no-tile-leg --ld-block 0; done

if [ "$NGPU" -ge 8 ]; then
    echo "== gvamp_sim as 8 RCCL ranks"
    T=$(mktemp -d)
    xz -dc tests/golden/survey_probe/toy.bed.xz > "$T/toy.bed"
    mkdir -p "$T/out"
    timeout -k 10 900 python scripts/run_sharded.py -n 8 -- gvamp_amd/gvamp_sim --bed-file "$T/toy.bed" --N 2000 --Mt 10000 \
        --out-dir "$T/out/" --out-name toy --iterations 3 --num-mix-comp 3 --probs 0.90,0.07,0.03 --vars 0,0.001,0.01 --CV 500 \
        --h2 0.5 --rho 0.5 --CG-max-iter 20 --model linear --seed 7 --store-pvals 0 > "$OUT/sim_np8.log" 2>&1
    python - "$T/out" "$OUT/sim_np8.json" <<'PY'
import json, sys, numpy as np
out, dst = sys.argv[1], sys.argv[2]
G = "tests/golden/survey_probe/"
res = {}
for name in ("it_1_x2_hat", "it_3", "it_3_x2_hat"):
    try:
        a, b = np.fromfile("%s/toy_%s.bin" % (out, name)), np.fromfile(G + "sim_np8_%s.bin" % name)
        res[name] = float(np.linalg.norm(a - b) / np.linalg.norm(b))
    except Exception as e:   # noqa: BLE001
        res[name] = repr(e)
res["ok"] = all(isinstance(v, float) and v < 1e-7 for v in res.values())
json.dump(res, open(dst, "w"))
print(res)
PY
    echo "== RCCL tests of the suite"
    timeout -k 10 1800 python -m pytest tests/test_gpu_multiproc.py -q -m gpu > "$OUT/pytest_multiproc.log" 2>&1
    tail -3 "$OUT/pytest_multiproc.log"
fi
python scripts/first_8gpu_check.py "$OUT"
