"""Verdict over the JSON files scripts/first_8gpu.sh leaves: scaling table, the assertions each leg must hold, A/B deltas.
Exit code 0 = all held (legs skipped for want of GPUs are reported, not failed).  --strict-band: an 8-GPU line outside the band
DESIGN.md section 6 predicted for it fails too."""
import glob
import json
import os
import sys


def load(path):
    try:
        txt = [ln for ln in open(path).read().splitlines() if ln.strip().startswith("{")]
        return json.loads(txt[-1]) if txt else {"failed": "no JSON line"}
    except (OSError, ValueError) as e:
        return {"failed": repr(e)}


# what the first 8-GPU line should read (how each band was derived)
PRED = {"ms_per_step": (3.85, 4.25), "value": (47000.0, 52000.0), "vamp_iters_per_s": (46.0, 58.0), "ms_allreduce_per_ax": (0.02, 0.15)}


def prediction_verdict(d8, pred=None):
    """[(key, lo, hi, measured, 'inside' | 'OUTSIDE' | 'n/a')] of an 8-GPU bench line against the predicted bands"""
    pred = pred or PRED
    got = {"ms_per_step": d8.get("ms_per_step"), "value": d8.get("value"), "vamp_iters_per_s": d8.get("vamp", {}).get("iters_per_s"),
           "ms_allreduce_per_ax": d8.get("multi_gpu", {}).get("ms_allreduce_per_ax")}
    out = []
    for k, (lo, hi) in pred.items():
        g = got[k]
        out.append((k, lo, hi, g, "n/a" if g is None else ("inside" if lo <= g <= hi else "OUTSIDE")))
    return out


def main(out, strict_band=False):
    bad, rows = [], {}
    last = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r*_bench_n1.json")))
    ref = json.load(open(last[-1]))["value"] if last else None
    for n in (1, 2, 4, 8):
        d = load(os.path.join(out, "bench_n%d.json" % n))
        if "skipped" in d:
            print("bench --gpus %d: %s" % (n, d["skipped"]))
            continue
        if "failed" in d:
            bad.append("bench --gpus %d: %s" % (n, d["failed"]))
            continue
        rows[n] = d
        if d.get("n_gpus") != n:
            bad.append("bench --gpus %d reported n_gpus %r" % (n, d.get("n_gpus")))
        if n > 1 and d.get("multi_gpu", {}).get("rccl_nranks") != n:
            bad.append("bench --gpus %d: multi_gpu.rccl_nranks = %r" % (n, d.get("multi_gpu", {}).get("rccl_nranks")))
        if n == 1 and ref and abs(d["value"] / ref - 1) > 0.03:
            bad.append("1-GPU value %.0f GB/s is more than 3 %% from the committed line %.0f (%s)" % (d["value"], ref, last[-1]))
        # every leg: the shards cover the marker range exactly once, in rank order (divide_work, utilities.cpp:259-291)
        pr = d.get("multi_gpu", {}).get("per_rank") if n > 1 else None
        if n > 1:
            if not pr or len(pr) != n:
                bad.append("bench --gpus %d: multi_gpu.per_rank has %s entries" % (n, len(pr) if pr else None))
            else:
                pr = sorted(pr, key=lambda r: r["rank"])
                at = 0
                for r in pr:
                    if r["first_marker"] != at:
                        bad.append("bench --gpus %d: rank %d starts at marker %d, expected %d" % (n, r["rank"], r["first_marker"], at))
                    at += r["markers"]
                if 1 in rows and "config" in rows[1] and at != rows[1]["config"].get("Mt", at):
                    bad.append("bench --gpus %d: the shards hold %d markers in all" % (n, at))
        # the VAMP leg of every leg agrees with its own reference sequence (x_hat 1e-7; north star 1e-5) and -- the recursion being the
        # same arithmetic whatever the sharding, up to the order of the cross-rank sums -- takes the CG steps of the 1-GPU run
        v = d.get("vamp", {})
        if v and v.get("x_hat_rel_l2") is not None and not v["x_hat_rel_l2"] < 1e-7:
            bad.append("bench --gpus %d: vamp.x_hat_rel_l2 = %r" % (n, v["x_hat_rel_l2"]))
        if n > 1 and 1 in rows and v.get("cg_iters") and rows[1].get("vamp", {}).get("cg_iters") and v["cg_iters"] != rows[1]["vamp"]["cg_iters"]:
            bad.append("bench --gpus %d: CG steps per iteration %s differ from the 1-GPU run's %s" % (n, v["cg_iters"], rows[1]["vamp"]["cg_iters"]))
    if 1 in rows:
        print("%-6s %12s %10s %12s %14s" % ("GPUs", "GB/s", "x 1 GPU", "efficiency", "VAMP it/s"))
        for n, d in sorted(rows.items()):
            s = d["value"] / rows[1]["value"]
            print("%-6d %12.0f %10.2f %12.3f %14s" % (n, d["value"], s, s / n, d.get("vamp", {}).get("iters_per_s")))
    # DESIGN.md section 6 wrote down, before any multi-GPU run existed, what the 8-GPU line should read (the sharded branches measured
    # on one GPU over an in-stream loop-back exchange, profiles/r6_forced_multi_gaps.txt, plus an assumed 40-100 us for a 3.2 MB
    # all-reduce over xGMI).  Outside the band is not a failure of the run -- it is the first thing to explain -- unless the caller
    # asks for it to be one (strict_band: tests/test_gpu_multiproc.py on an 8-GPU node).
    if 8 in rows:
        for k, lo, hi, g, where in prediction_verdict(rows[8]):
            print("PREDICTION 8 GPUs %-22s predicted %.4g .. %.4g   measured %s   %s" % (k, lo, hi, g, where))
            if strict_band and where != "inside":
                bad.append("8-GPU %s = %s is %s the predicted band %.4g .. %.4g (DESIGN.md section 6)" % (k, g, "not reported: cannot be held against" if g is None else "outside", lo, hi))
    for fam, keys in (("overlap", (0, 2, 4)), ("cgdevice", (0, 1))):
        vals = {k: load(os.path.join(out, "%s_%d.json" % (fam, k))) for k in keys}
        if all("value" in v for v in vals.values()):
            print(fam + ": " + ", ".join("%d -> %.0f GB/s, %s it/s, all-reduce %.3f ms/Ax" % (
                k, v["value"], v.get("vamp", {}).get("iters_per_s"), v.get("multi_gpu", {}).get("ms_allreduce_per_ax", float("nan")))
                for k, v in vals.items()))
        else:
            print(fam + ": " + ", ".join("%d: %s" % (k, v.get("skipped") or v.get("failed")) for k, v in vals.items() if "value" not in v))
            bad += ["%s_%d: %s" % (fam, k, v["failed"]) for k, v in vals.items() if "failed" in v]
    sim = os.path.join(out, "sim_np8.json")
    if os.path.exists(sim):
        s = json.load(open(sim))
        print("gvamp_sim np=8 vs the reference's files:", s)
        if not s.get("ok"):
            bad.append("gvamp_sim np=8 does not reproduce tests/golden/survey_probe/sim_np8_*")
    for b in bad:
        print("FAILED:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    args = [x for x in sys.argv[1:] if x != "--strict-band"]
    sys.exit(main(args[0] if args else "gpurun_out/first8", strict_band="--strict-band" in sys.argv[1:]))
