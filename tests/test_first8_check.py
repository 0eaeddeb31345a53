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


def _eight(tmp_path, **over):
    """a consistent 8-GPU leg beside the committed 1-GPU line; `over` replaces top-level / vamp / multi_gpu entries"""
    ref = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))[-1]))
    Mt = ref["config"].get("Mt", 1000000)
    size, modu = divmod(Mt, 8)
    per, at = [], 0
    for r in range(8):
        m = size + 1 if r < modu else size
        per.append({"rank": r, "markers": m, "first_marker": at})
        at += m
    eight = dict(ref, n_gpus=8, ms_per_step=4.0, value=50000.0)
    eight["multi_gpu"] = {"rccl_nranks": 8, "ms_allreduce_per_ax": 0.06, "per_rank": per}
    eight["vamp"] = dict(ref["vamp"], iters_per_s=50.0)
    for k, v in over.items():
        if k in ("iters_per_s",):
            eight["vamp"][k] = v
        elif k in ("ms_allreduce_per_ax", "rccl_nranks"):
            eight["multi_gpu"][k] = v
        else:
            eight[k] = v
    json.dump(dict(ref, n_gpus=1), open(tmp_path / "bench_n1.json", "w"))
    json.dump(eight, open(tmp_path / "bench_n8.json", "w"))
    for name in ("bench_n2", "bench_n4", "overlap_0", "overlap_2", "overlap_4", "cgdevice_0", "cgdevice_1"):
        json.dump({"skipped": "not run"}, open(tmp_path / (name + ".json"), "w"))


def test_predicted_band_of_the_first_8gpu_line(tmp_path):
    """The bands DESIGN.md section 6 predicts for the first 8-GPU bench line: reported always, a failure under --strict-band (what
    tests/test_gpu_multiproc.py::test_first_contact_eight_gpus asserts on an 8-GPU node)."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import first_8gpu_check as chk
    for k, (lo, hi) in chk.PRED.items():
        assert 0 < lo < hi
    # the bands are consistent with one another and with the committed 1-GPU line: step time <-> aggregate rate over the same bytes
    ref = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_n1.json")))[-1]))
    gb_per_step = ref["value"] * ref["ms_per_step"] / 1e3                         # GB one lmmse_mult streams at the headline
    assert chk.PRED["value"][0] <= gb_per_step / chk.PRED["ms_per_step"][1] * 1e3 * 1.02
    assert chk.PRED["value"][1] >= gb_per_step / chk.PRED["ms_per_step"][0] * 1e3 * 0.98
    assert chk.PRED["value"][1] < 8 * ref["value"] * 1.05                         # nobody predicted super-linear scaling
    _eight(tmp_path)
    script = os.path.join(ROOT, "scripts", "first_8gpu_check.py")
    r = subprocess.run([sys.executable, script, str(tmp_path), "--strict-band"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("inside") == 4 and "OUTSIDE" not in r.stdout, r.stdout
    for over, needle in ((dict(ms_per_step=5.2, value=38000.0), "ms_per_step"), (dict(iters_per_s=30.0), "vamp_iters_per_s"),
                         (dict(ms_allreduce_per_ax=0.9), "ms_allreduce_per_ax")):
        _eight(tmp_path, **over)
        loose = subprocess.run([sys.executable, script, str(tmp_path)], capture_output=True, text=True)
        assert loose.returncode == 0 and "OUTSIDE" in loose.stdout, loose.stdout          # reported, not failed
        strict = subprocess.run([sys.executable, script, str(tmp_path), "--strict-band"], capture_output=True, text=True)
        assert strict.returncode == 1 and needle in strict.stdout and "FAILED" in strict.stdout, strict.stdout
    got = {k: w for k, lo, hi, g, w in chk.prediction_verdict({"ms_per_step": 4.0, "value": 1.0, "vamp": {}, "multi_gpu": {}})}
    assert got == {"ms_per_step": "inside", "value": "OUTSIDE", "vamp_iters_per_s": "n/a", "ms_allreduce_per_ax": "n/a"}
