"""Measurement of the widened rows (SURVEY 8f / docs/history/rounds1-3.md section 8) at the sizes BASELINE.json names for them.

One JSON object per row on stdout (development tool: bench.py stays the contract for the headline metric).
  python scripts/bench_rows.py > gpurun_out/rows.json
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (first: libamdhip64 / librccl are shared with libgvamp)

from gvamp_amd import capi, hostapi


def shard_bytes(N, M):
    mb = (N + 3) // 4
    return M * mb + 24 * M + 32 * mb


def vamp_row(name, N, M, iterations, **kw):
    raw = kw.pop("raw_rows", False)
    with capi.Shard(N, M) as sh:
        sh.set_layout(raw, int(os.environ.get("GV_LAYOUT", "1")))
        sh.set_kernel_mode(1)
        t = time.perf_counter()
        sh.synth_bed(4242, 5000)
        sh.compute_markers_statistics()
        sh.synchronize()
        ingest = time.perf_counter() - t
        beta, y = hostapi.sim_phen(sh, 0.5, max(1, M // 100), 1)
        if kw.get("model") == "bin_class":
            y = (y > 0).astype(float)                      # case / control labels from the simulated liability
        r = hostapi.infere_linear(sh, y, None, None, iterations=iterations, CG_max_iter=50, rho=0.5, seed=1, true_signal=beta,
                                  history=False, **kw)
    its = r.trace
    tail = its[1:] if len(its) > 1 else its
    secs = sum(t["seconds"] for t in tail)
    return {"row": name, "N": N, "M": M, "iters_per_s": round(len(tail) / secs, 3),
            "seconds_per_iter": [round(t["seconds"], 4) for t in its], "cg_iters": [t["cg_iters"] for t in its],
            "n_ax_pass": [t["n_ax_pass"] for t in its], "n_atx_pass": [t["n_atx_pass"] for t in its],
            "pass_GBps": round(sum(t["n_ax_pass"] + t["n_atx_pass"] for t in tail) * shard_bytes(N, M) / secs / 1e9, 1),
            # the same time priced per vector PRODUCT (what a one-product-per-pass engine would have had to stream): not a
            # bandwidth -- a two-vector pass reads the shard once for two products
            "n_ax": [t["n_ax"] for t in its], "n_atx": [t["n_atx"] for t in its],
            "product_equiv_GBps": round(sum(t["n_ax"] + t["n_atx"] for t in tail) * shard_bytes(N, M) / secs / 1e9, 1),
            "corr_with_truth": round(float(np.corrcoef(r.x_est, beta)[0, 1]), 4), "ingest_s": round(ingest, 2)}


def pvals_row(N, M):
    rng = np.random.default_rng(0)
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, int(os.environ.get("GV_LAYOUT", "1")))
        sh.set_kernel_mode(1)
        sh.synth_bed(4242, 5000)
        sh.compute_markers_statistics()
        x = sh.vecM(rng.standard_normal(M) * (rng.random(M) < 0.01))
        z1, y = sh.vecN(), sh.vecN(rng.standard_normal(N))
        sh.ax_dev(x, z1)
        sh.pvals_calc(z1, y, x)                              # warm-up
        sh.synchronize()
        t = time.perf_counter()
        reps = 3
        for _ in range(reps):
            pv = sh.pvals_calc(z1, y, x)
        dt = (time.perf_counter() - t) / reps
    return {"row": "p-values LOO (gv_pvals_loo: one two-vector pass + per-marker t-test kernel)", "N": N, "M": M,
            "seconds": round(dt, 4), "GBps_whole_call": round(shard_bytes(N, M) / dt / 1e9, 1),
            "finite": bool(np.all(np.isfinite(pv)))}


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""          # e.g. "config 5": the rows whose name contains it
    specs = [
        ("config 2: linear N=100k x M=500k, CG-max-iter 50, fuse-solves 4", lambda n: vamp_row(n, 100000, 500000, 5, fuse_solves=4)),
        ("one shard of the 8-GPU headline job on its own: N=400k x M=125k (no exchange), fuse-solves 4",
         lambda n: vamp_row(n, 400000, 125000, 6, fuse_solves=4)),
        ("config 4: probit N=100k x M=500k, fuse-solves 4",
         lambda n: vamp_row(n, 100000, 500000, 5, fuse_solves=4, model="bin_class", gam1=1e-8, gamw=1.0)),
        ("config 5: --use-XXT-denoiser 1 (matrix-free N-space CG) N=50k x M=200k, fuse-solves 4",
         lambda n: vamp_row(n, 50000, 200000, 4, use_XXT_denoiser=1, raw_rows=True, fuse_solves=4)),
        ("config 5 as the reference sequences it (fuse-solves 0)",
         lambda n: vamp_row(n, 50000, 200000, 4, use_XXT_denoiser=1, raw_rows=True, fuse_solves=0)),
        ("p-values LOO", lambda n: pvals_row(400000, 1000000)),
    ]
    for name, fn in specs:
        if only in name:
            print(json.dumps(fn(name)), flush=True)


if __name__ == "__main__":
    main()
