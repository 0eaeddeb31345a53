"""scripts/first_8gpu_check.py -- the verdict over the legs of scripts/first_8gpu.sh -- on synthetic legs built from the committed
1-GPU bench line: a consistent 2-GPU leg passes; shards that do not tile the marker range, a wrong communicator size, an x_hat
that left the tolerance, or different CG step counts are each caught.  (No 8-GPU node has been available: the kit's own logic is
what can be tested.)"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _legs(tmp_path, mutate=None):
    ref = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))[-1]))
    one = dict(ref, n_gpus=1)
    two = dict(ref, n_gpus=2)
    half = 500000
    two["multi_gpu"] = {"rccl_nranks": 2, "ms_allreduce_per_ax": 0.1,
                        "per_rank": [{"rank": 0, "markers": half, "first_marker": 0}, {"rank": 1, "markers": half, "first_marker": half}]}
    two["vamp"] = dict(ref["vamp"])
    if mutate:
        mutate(two)
    json.dump(one, open(tmp_path / "bench_n1.json", "w"))
    json.dump(two, open(tmp_path / "bench_n2.json", "w"))
    for name in ("bench_n4", "bench_n8", "overlap_0", "overlap_2", "overlap_4", "cgdevice_0", "cgdevice_1"):
        json.dump({"skipped": "needs 8 GPUs, have 2"}, open(tmp_path / (name + ".json"), "w"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "first_8gpu_check.py"), str(tmp_path)], capture_output=True, text=True)
    return r.returncode, r.stdout


def test_consistent_legs_pass_and_print_the_scaling_table(tmp_path):
    rc, out = _legs(tmp_path)
    assert rc == 0, out
    assert "efficiency" in out and "FAILED" not in out


def test_inconsistent_legs_are_caught(tmp_path):
    def shards(d):
        d["multi_gpu"]["per_rank"][1]["first_marker"] -= 1
    def nranks(d):
        d["multi_gpu"]["rccl_nranks"] = 1
    def xhat(d):
        d["vamp"]["x_hat_rel_l2"] = 3e-6
    def steps(d):
        d["vamp"]["cg_iters"] = [c + 1 for c in d["vamp"]["cg_iters"]]
    for mut, needle in ((shards, "starts at marker"), (nranks, "rccl_nranks"), (xhat, "x_hat_rel_l2"), (steps, "CG steps")):
        rc, out = _legs(tmp_path, mut)
        assert rc == 1 and needle in out, (needle, out)
