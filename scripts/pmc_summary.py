"""Summarise the rocprofv3 PMC passes of bench.py into profiles/ (development tool).

usage: python scripts/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r1 [N Mt]

Reads <dir>/bench_counter_collection.csv of the two separate passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE), averages
the counters per kernel and writes <prefix>_pmc_summary.csv and <prefix>_pmc_traffic.json.  HBM bytes per launch follow
MI355X_MICROARCH.md (HBM / rocprofv3 section): both counters are in KB on gfx950 and FETCH_SIZE counts wide streaming reads at
one half, so hbm_bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024.
"""
import csv
import json
import sys
from collections import defaultdict


def kernel_sources_sha256():
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("gv_mfma.hip", "gv_mfma.h", "gv_pval_dev.h"):      # (gv_pval_dev.h is compiled into k_fin_pvals)
        with open(os.path.join(root, "gvamp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def per_kernel(path, counter):
    """average of `counter` per kernel over its FULL launches: a CG step enqueued before the host knew that every system had
    converged returns at once on the device (gv_solvers.hip: cg_run_device) and would pull the averages down"""
    vals = defaultdict(list)
    with open(path + "/bench_counter_collection.csv") as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            vals[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    out = {}
    for k, v in vals.items():
        top = max(v)
        full = [x for x in v if x >= 0.5 * top] if top > 0 else v
        out[k] = (len(full), sum(full) / len(full))
    return out


def main():
    fdir, wdir, prefix = sys.argv[1:4]
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 400000
    Mt = int(sys.argv[5]) if len(sys.argv) > 5 else 1000000
    fetch = per_kernel(fdir, "FETCH_SIZE")
    write = per_kernel(wdir, "WRITE_SIZE")
    with open(prefix + "_pmc_summary.csv", "w") as f:
        f.write("counter,kernel,dispatches,avg_value_KB\n")
        for name, tab in (("FETCH_SIZE", fetch), ("WRITE_SIZE", write)):
            for k, (n, v) in sorted(tab.items(), key=lambda kv: -kv[1][1] * kv[1][0]):
                f.write('%s,"%s",%d,%.3f\n' % (name, k[:110].replace('"', "'"), n, v))

    def pick(tab, frag):
        for k, v in tab.items():
            if frag in k:
                return v[1]
        return None

    out = {
        "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 3 --warmup 1 "
                "--no-cpu-baseline --vamp-iterations 0`, work decompositions read from the tuning cache (no candidate launches); hbm_bytes = 2*FETCH_SIZE_KB*1024 + WRITE_SIZE_KB*1024 (gfx950: "
                "FETCH_SIZE counts wide streaming reads at half, MI355X_MICROARCH.md section HBM)",
        "N": N, "Mt": Mt, "n_gpus": 1, "kernel_mode": 1,
        # identity of the kernel sources these counters belong to: bench.py quotes `roofline.traffic` from this file only
        # while its own gv_mfma.hip / gv_mfma.h hash to the same value
        "kernel_sources_sha256": kernel_sources_sha256(),
    }
    for key, frag in (("ax", "k_mfma_matvec<1,"), ("atx", "k_mfma_matvec<0,"), ("ax2", "k_mfma_matvec<3,"),
                      ("atx2", "k_mfma_matvec<2,"), ("tile_ax", "k_mfma_tile<1, 3,"), ("tile_atx", "k_mfma_tile<0, 0,"),
                      ("tile_atx2", "k_mfma_tile<0, 2,")):
        fk, wk = pick(fetch, frag), pick(write, frag)
        if fk is None or wk is None:
            continue
        out[key] = {"fetch_KB": round(fk, 3), "write_KB": round(wk, 3), "hbm_bytes": int(2 * fk * 1024 + wk * 1024)}
    with open(prefix + "_pmc_traffic.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
