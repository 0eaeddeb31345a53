#!/usr/bin/env python3
"""bench.py -- headline benchmark of the gVAMP hot path on MI355X.

Metric (BASELINE.json): genotype matvec GB/s at N=400k x Mt=1M, 1/2/4/8 GPUs (strong scaling: the marker
dimension is sharded exactly as the reference shards over MPI ranks, utilities.cpp:259-291).

A "step" is one application of the LMMSE operator of the CG loop (vamp.cpp:1074-1118):
    d = tau * A^T (A p) + gam2 * p   =  data::Ax (+ N-vector all-reduce across ranks) + data::ATx + axpy
i.e. two streams over the rank's whole 2-bit genotype shard.  `value` = algorithmic bytes of the WHOLE job per
step (2 x (Mt*ceil(N/4) + 24*Mt + 32*ceil(N/4))) x steps / wall time, inputs resident in HBM.

One process per GPU:  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=400000, help="individuals")
    ap.add_argument("--Mt", type=int, default=1000000, help="total markers")
    ap.add_argument("--mode", type=int, default=1, help="0 = fp64 VALU kernels, 1 = i8 MFMA fixed point")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-markers", type=int, default=0, help="markers of the CPU-baseline sample (0 = auto)")
    ap.add_argument("--vamp-iterations", type=int, default=5, help="VAMP iterations of the iters/s leg (0 = skip)")
    ap.add_argument("--CG-max-iter", type=int, default=50)
    ap.add_argument("--ld-block", type=int, default=64, help="second VAMP setting: markers per LD block (0 = skip that leg)")
    ap.add_argument("--ld-ppm", type=int, default=900000, help="second VAMP setting: within-block copy probability, 1e-6")
    ap.add_argument("--no-side-leg", "--no-tile-leg", dest="no_side_leg", action="store_true",
                    help="skip the closing measurement of the same step on the OTHER resident layout")
    ap.add_argument("--no-rows", action="store_true", help="skip the mid-size rows (config 2 / one 8-GPU shard / config 4 / config 5)")
    ap.add_argument("--rows-only", action="store_true", help="only the mid-size rows: prints {\"rows\": [...]} (profiles/r*_rows.json)")
    ap.add_argument("--rows-layout", type=int, default=0, help="resident layout of the rows: 0 = what the drivers get for a run of that "
                    "length (gv_set_expected_passes(iterations x 12) -> one tile layout), 1 / 2 = fixed")
    ap.add_argument("--layout", type=int, default=0, help="resident re-encoding of kernel mode 1: 0 = leave the library's default "
                    "(auto: one tile layout unless the run is announced as long and two stripe sets fit), 1 = two stripe sets "
                    "(2 x M*N/4 bytes), 2 = one tile layout (M*N/4 bytes)")
    ap.add_argument("--fuse-solves", type=int, default=4,
                    help="0 = the reference's sequence of matvecs, 1 = LMMSE and Onsager CG share passes (bit-identical), "
                         "2 = also z1 rides in a free slot and A x2_hat / A^T A invQ u come out of the CG recurrences, "
                         "3 = also the warm start's initial residual comes from the previous solve (no Ax + ATx for it), "
                         "4 = also the Onsager solve's first step comes from A^T A u of the probe, computed once")
    return ap.parse_args()


def divide_work(Mt, nranks, rank):
    """utilities.cpp:259-291"""
    size, modu = divmod(Mt, nranks)
    lens = [size + 1 if i < modu else size for i in range(nranks)]
    return lens[rank], sum(lens[:rank])


def alg_bytes(N, M):
    mb = (N + 3) // 4
    return M * mb + 24 * M + 32 * mb


def cpu_baseline(N, seed, want_markers, device):
    """Oracle (CPU restatement, OpenMP) timed on the GPU box's host cores, per SURVEY 8(d): a timing-only build of the port
    (-O3 -march=native -fopenmp, compiled on this host: oracle/Makefile `timing`; the parity library keeps its own flags), on a
    bounded sample -- the first `m` markers of the same synthetic matrix:
      * one Ax + one ATx per repetition at several thread counts (all hardware threads, half, a quarter): the best is `value`;
      * REAL VAMP iterations of the port: config 1 whole (N=2000 x M=10000) and the m-marker slice at full N -- the measured
        second iteration beside what the matvec model (n_ax * t_ax + n_atx * t_atx) says for that same iteration, so that the
        extrapolation to the headline size rests on a checked model.
    A port, not the reference (unbuildable here): no credit either way is claimed from the GPU / CPU ratio -- the roofline
    fraction is what measures the kernels."""
    from gvamp_amd import capi
    from oracle import gvoracle as go
    flags = go.use_timing_build()
    hw = os.cpu_count() or 1
    mb = (N + 3) // 4
    m = want_markers or max(256, min(20000, int(2.0e9 // mb)))
    with capi.Shard(N, m, Mt=m, S=0, device=device) as sh:
        sh.set_layout(True, False)
        sh.synth_bed(seed, 5000)
        bed = sh.download_bed()
    rng = np.random.default_rng(0)
    mave, msig = go.marker_stats(bed, N, m, nthreads=hw)
    x = rng.standard_normal(m)
    tried = []
    for nt in sorted({hw, max(1, hw // 2), max(1, hw // 4)}, reverse=True):
        reps, t_ax, t_atx = 0, 0.0, 0.0
        while reps < 1 or (t_ax + t_atx < 3.0 and reps < 30):
            t1 = time.time()
            z = go.ax(bed, N, m, mave, msig, x, nthreads=nt)
            t2 = time.time()
            go.atx(bed, N, m, mave, msig, z, nthreads=nt)
            t3 = time.time()
            t_ax += t2 - t1
            t_atx += t3 - t2
            reps += 1
        tried.append({"threads": nt, "GBps": round(2 * alg_bytes(N, m) * reps / (t_ax + t_atx) / 1e9, 3),
                      "ax_s": t_ax / reps, "atx_s": t_atx / reps, "reps": reps})
    best = max(tried, key=lambda t: t["GBps"])
    out = {"value": best["GBps"], "unit": "GB/s", "cores": best["threads"], "kind": "port", "flags": flags, "sample_markers": m,
           "ax_s": best["ax_s"], "atx_s": best["atx_s"], "hardware_threads": hw,
           "thread_sweep": [{k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()} for t in tried],
           "sample": "oracle/ (OpenMP, %s) Ax+ATx on the first %d markers x N=%d of the same synthetic matrix; best of %s threads: "
                     "%d threads, Ax %.3f s, ATx %.3f s per call (scalar port; the reference itself cannot be built here)"
                     % (flags.split(" (")[0], m, N, "/".join(str(t["threads"]) for t in tried), best["threads"], best["ax_s"], best["atx_s"]),
           "cpu_model": _cpu_model()}
    # ---- real VAMP iterations of the port, and the matvec model checked against them
    nt = best["threads"]
    measured = {}
    try:
        kw = dict(iterations=2, CG_max_iter=50, rho=0.5, seed=1, gam1=1e-8, gamw=2.0, nthreads=nt)
        beta, y = go.sim_phen(bed, N, m, 0.5, max(1, m // 100), 1, nthreads=nt)
        # (the default 23-component prior needs Mt >= 50 000 -- utilities.cpp:91-140 -- so the slice takes config 1's three components)
        r = go.infere(bed, N, m, y, [0.90, 0.07, 0.03], [0, 0.001, 0.01], true_signal=beta, **kw)
        t2 = r.trace[-1]
        model = t2["n_ax"] * best["ax_s"] + t2["n_atx"] * best["atx_s"]
        measured["slice_N%d_M%d" % (N, m)] = {"seconds": round(t2["seconds"], 3), "n_ax": int(t2["n_ax"]), "n_atx": int(t2["n_atx"]),
                                              "matvec_model_s": round(model, 3), "measured_over_model": round(t2["seconds"] / model, 3)}
        out["measured_over_model"] = round(t2["seconds"] / model, 3)
        with capi.Shard(2000, 10000, device=device) as s1:          # config 1 whole (sim.cpp N=2000 M=10000, 3 mixture components)
            s1.set_layout(True, False)
            s1.synth_bed(seed, 5000)
            bed1 = s1.download_bed()
        b1, y1 = go.sim_phen(bed1, 2000, 10000, 0.5, 100, 1, nthreads=nt)
        r1 = go.infere(bed1, 2000, 10000, y1, [0.90, 0.07, 0.03], [0, 0.001, 0.01], true_signal=b1, **dict(kw, iterations=3))
        measured["config1_N2000_M10000"] = {"seconds": round(float(np.mean([t["seconds"] for t in r1.trace[1:]])), 4),
                                            "n_ax": int(r1.trace[-1]["n_ax"]), "n_atx": int(r1.trace[-1]["n_atx"])}
    except Exception as e:      # noqa: BLE001  (a baseline leg must never take the bench line down)
        measured["error"] = repr(e)
    out["vamp_iter_s_measured"] = measured
    return out


def kernel_sources_sha256():
    """identity of the streaming-kernel sources a PMC profile belongs to (scripts/pmc_summary.py stamps the same hash)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("gv_mfma.hip", "gv_mfma.h", "gv_pval_dev.h"):      # (gv_pval_dev.h is compiled into k_fin_pvals)
        with open(os.path.join(ROOT, "gvamp_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


ROWS = [
    # (key, what, N, M, iterations, extra arguments of infere)
    ("config2", "config 2: linear N=100k x M=500k, CG-max-iter 50", 100000, 500000, 5, {}),
    ("shard_8gpu", "one shard of the 8-GPU headline job on its own: N=400k x M=125k (no exchange)", 400000, 125000, 6, {}),
    ("config4", "config 4: probit (--model bin_class) N=100k x M=500k", 100000, 500000, 5, dict(model="bin_class", gam1=1e-8, gamw=1.0)),
    ("config5", "config 5: --use-XXT-denoiser 1 (matrix-free N-space CG) N=50k x M=200k", 50000, 200000, 4, dict(use_XXT_denoiser=1)),
]


def run_rows(a, device):
    """The mid-size rows of SURVEY 8(d) / BASELINE configs 2, 4, 5 and the per-GPU shard of config 3: whole VAMP iterations of the
    host loop on a fresh shard each, --fuse-solves 4 (the drivers' default) and 0 (the reference's own sequence of products) side
    by side on the SAME resident shard.  it/s over iterations 2..; pass_GBps = passes over the shard x algorithmic bytes / time;
    frac = that over the 8 TB/s peak.  The layout is what a driver run of that length gets (gv_set_expected_passes(iterations x 12))
    unless --rows-layout fixes it."""
    from gvamp_amd import capi, hostapi
    rows = []
    for key, what, N, M, iterations, kw in ROWS:
        with capi.Shard(N, M, device=device) as sh:
            if a.rows_layout:
                sh.set_layout(False, a.rows_layout)
            else:
                sh.set_expected_passes(iterations * 12)
            t = time.perf_counter()
            sh.synth_bed(4242, 5000)
            sh.compute_markers_statistics()
            sh.synchronize()
            ingest = time.perf_counter() - t
            beta, y = hostapi.sim_phen(sh, 0.5, max(1, M // 100), 1)
            if kw.get("model") == "bin_class":
                y = (y > 0).astype(float)                      # case / control labels from the simulated liability
            row = {"row": key, "what": what, "N": N, "M": M, "layout": sh.get_layout(), "ingest_s": round(ingest, 3),
                   "alg_GB_per_pass": round(alg_bytes(N, M) / 1e9, 3)}
            xs = {}
            for fuse in (4, 0):
                r = hostapi.infere_linear(sh, y, None, None, iterations=iterations, CG_max_iter=a.CG_max_iter, rho=0.5, seed=1,
                                          true_signal=beta, history=False, fuse_solves=fuse, **kw)
                its = r.trace
                tail = its[1:] if len(its) > 1 else its
                secs = sum(t["seconds"] for t in tail)
                npass = sum(t["n_ax_pass"] + t["n_atx_pass"] for t in tail)
                gbps = npass * alg_bytes(N, M) / secs / 1e9
                xs[fuse] = r.x_est
                row["fuse_%d" % fuse] = {
                    "iters_per_s": round(len(tail) / secs, 2), "passes": npass, "pass_GBps": round(gbps, 1), "frac": round(gbps / 8000.0, 4),
                    "seconds_per_iter": [round(t["seconds"], 5) for t in its], "cg_iters": [t["cg_iters"] for t in its],
                    "n_ax_pass": [t["n_ax_pass"] for t in its], "n_atx_pass": [t["n_atx_pass"] for t in its]}
            den = float(np.linalg.norm(xs[0]))
            row["x_hat_rel_l2_fuse4_vs_0"] = float(np.linalg.norm(xs[4] - xs[0]) / den) if den > 0 else None
            row["corr_with_truth"] = round(float(np.corrcoef(xs[4], beta)[0, 1]), 4)
        rows.append(row)
    return rows


def spawn_ranks(a):
    """`python bench.py --gpus N` with no launcher: start the N per-GPU ranks ourselves, as plain child processes, BEFORE
    anything in this process touches the GPU (torch.cuda.device_count() does not initialise it on this image), and leave
    with the first non-zero exit code of a rank.  Rank 0 inherits stdout: the one JSON line is its."""
    import socket
    import subprocess
    import torch
    have = torch.cuda.device_count()
    if have < a.gpus:
        print("bench.py: --gpus %d but this box has %d GPU(s); refusing to report a smaller job under that name"
              % (a.gpus, have), file=sys.stderr)
        sys.exit(3)
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(a.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("NCCL_SOCKET_IFNAME", "lo")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc, live = 0, set(range(a.gpus))
    while live:
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.discard(r)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    procs[q].terminate()
        time.sleep(0.05)
    sys.exit(rc)


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        spawn_ranks(a)                       # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        # never downgrade: a harness that asked for N GPUs must not get another job's number under that name
        if rank == 0:
            print("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d; start it as `python -m torch.distributed.run "
                  "--nproc-per-node %d ... bench.py --gpus %d`, or bare (it then spawns its own ranks)"
                  % (a.gpus, world, a.gpus, a.gpus), file=sys.stderr)
        sys.exit(2)

    # torch first: it brings its own ROCm runtime libraries, and libgvamp.so then binds to the same loaded
    # libamdhip64 / librccl (same SONAMEs) -- the other order leaves torch without a device.
    import torch
    import torch.distributed as dist
    from gvamp_amd import capi      # the HIP library: raises if it is not built / no GPU
    capi.load()

    dev_ok = torch.cuda.is_available()
    if not dev_ok:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    # one process per GPU: each rank on the CPUs of the NUMA node its GPU hangs off (no-op on a single-node host)
    numa_node = capi.bind_host_numa(local_rank) if (world > 1 or os.environ.get("GVAMP_NUMA_BIND") == "1") else -1
    force_dist = os.environ.get("GVAMP_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ   # exercise the
    # N > 1 plumbing (rendezvous, id broadcast, RCCL communicator) on a single-GPU box under torchrun --nproc-per-node 1
    if world > 1 or force_dist:
        # all ranks of this bench live on ONE node: keep RCCL's out-of-band bootstrap on the loopback interface unless the
        # launcher says otherwise (the container's hostname / first NIC need not be reachable from inside it)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # rendezvous + timing reductions only;
        # the data path's collectives are RCCL calls inside libgvamp (gv_comm_init below)

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()

    if a.rows_only:
        if rank == 0:
            print(json.dumps({"rows": run_rows(a, local_rank), "rows_layout": a.rows_layout or "as the drivers (expected passes)"}), flush=True)
        return
    N, Mt = a.N, a.Mt
    M, S = divide_work(Mt, world, rank)
    sh = capi.Shard(N, M, Mt=Mt, S=S, device=local_rank)
    # The default run configures NOTHING: what is measured is what a binding that only calls gv_create / gv_set_dims /
    # gv_upload_bed / gv_ax / gv_atx gets -- kernel mode 1, no raw rows, layout picked at ingest.
    engine_defaults = a.mode == 1 and a.layout == 0
    if a.mode == 1 and a.layout != 0:
        sh.set_layout(False, a.layout)     # no raw rows resident: 2 x M*N/4 bytes (two stripe sets) or M*N/4 (tile layout)
    elif a.mode == 0:
        sh.set_layout(True, False)
        sh.set_kernel_mode(0)
    assert sh.get_kernel_mode() == a.mode
    # what this run is about to do with the shard (gv_set_expected_passes, as the drivers announce iterations x 12): the timed steps
    # and the VAMP legs (levels 4 / 0 / 3, then the LD genotypes at 4 / 0), ~12 ATx passes per iteration
    planned_passes = a.warmup + a.steps + a.vamp_iterations * 12 * (3 + (2 if a.ld_block > 0 else 0))
    sh.set_expected_passes(planned_passes)
    t0 = time.time()
    sh.synth_bed(a.seed, 5000)
    sh.compute_markers_statistics()
    t_ingest = time.time() - t0
    t_alloc, t_fill = sh.ingest_info()      # hipMalloc of the resident layouts (driver: page mapping / wipe) vs generating them
    layout = sh.get_layout()                # 1 two stripe sets, 2 tile layout, 0 none (kernel mode 0)
    ingest_stats = sh.ingest_stats()
    if world > 1 or force_dist:
        uid = [capi.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(uid, src=0)
        sh.comm_init(world, rank, uid[0])
    # the automatic layout is picked per rank from its own free HBM: a leg every rank must enter together (the closing tile-layout
    # leg starts with a barrier) is gated on what ALL ranks hold, not on this rank's pick
    layout_everywhere = layout
    if world > 1 or force_dist:
        t = torch.tensor([float(layout), -float(layout)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        layout_everywhere = layout if t[0].item() == -t[1].item() else -1      # -1: the ranks hold different layouts

    rng = np.random.default_rng(7)          # same p on every rank's own slice
    rng_p = rng.standard_normal(Mt)[S:S + M]
    p = sh.vecM(rng_p)
    d = sh.vecM()
    tau, gam2 = 2.0, 1.35

    def step():
        sh.lmmse_mult(p, tau, gam2, d)

    for _ in range(a.warmup):
        step()
    sh.synchronize()
    tune_s, tune_src = sh.tune_info()       # the first matvec picked the work decompositions (measured, or from the cache)
    sh.set_timing(2)
    sh.counters(reset=True)
    sh.synchronize()
    torch.cuda.synchronize()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sh.synchronize()
    torch.cuda.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    cnt = sh.counters()
    sh.set_timing(0)
    d_host = d.download()            # (after the timed region) the operator's output, compared with the tile-layout leg below
    # PCIe-inclusive rate through the host-pointer entry points gv_ax / gv_atx (data::Ax / data::ATx as the reference's vamp.cpp
    # calls them: the M- or N-vector crosses PCIe each way).  Outside the timed region; never `value`.
    hx = np.ascontiguousarray(rng_p)
    hp = np.zeros(4 * ((N + 3) // 4))
    hp[:N] = np.random.default_rng(8).standard_normal(N)
    sh.Ax(hx); sh.ATx(hp)
    t1 = time.perf_counter()
    for _ in range(5):
        sh.Ax(hx)
    t_hax = (time.perf_counter() - t1) / 5
    t1 = time.perf_counter()
    for _ in range(5):
        sh.ATx(hp)
    t_hatx = (time.perf_counter() - t1) / 5
    sh.counters(reset=True)

    job_bytes = 2 * alg_bytes(N, Mt)
    value = job_bytes * a.steps / dt / 1e9
    # roofline of the dominant kernel on THIS rank's shard (Ax: the 2-bit-transposed stream)
    shard_bytes = alg_bytes(N, M)
    ms_ax = cnt["ms_ax_kernel"] / max(cnt["n_ax_kernel"], 1)
    ms_atx = cnt["ms_atx_kernel"] / max(cnt["n_atx_kernel"], 1)
    ax_gbps = shard_bytes / (ms_ax * 1e-3) / 1e9 if ms_ax > 0 else 0.0
    atx_gbps = shard_bytes / (ms_atx * 1e-3) / 1e9 if ms_atx > 0 else 0.0
    dec = sh.decomp()
    if a.mode == 1 and layout == 2:     # template arguments: <DIR, MODE, balanced decomposition, device-CG instantiation>
        kname = "k_mfma_tile<1, 3, %s, false> (Ax)" % ("true" if "balanced_cells" in dec["ax"] else "false")
    elif a.mode == 1:                     # <MODE, balanced decomposition, device-CG instantiation>
        kname = "k_mfma_matvec<1, %s, false> (Ax)" % ("true" if "balanced_cells" in dec["ax"] else "false")
    else:
        kname = "k_ax_f64"
    # HBM traffic of that kernel comes from separate rocprofv3 --pmc passes of this same command (counters cannot be
    # read from inside the process).  The newest committed summary is quoted ONLY when it was taken on this configuration
    # AND on these very kernel sources (sha256 of gv_mfma.hip recorded by scripts/pmc_summary.py); otherwise null.
    traffic, traffic_src = None, None
    ksha = kernel_sources_sha256()
    for name in sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json")), reverse=True):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", name)))
            if (pm["N"], pm["Mt"], pm["n_gpus"], pm["kernel_mode"]) == (N, Mt, world, a.mode) and \
                    pm.get("kernel_sources_sha256") == ksha:
                # (the counters of the kernel this run's roofline names: the tile layout's Ax kernel or the stripe sets')
                traffic, traffic_src = pm["tile_ax" if layout == 2 else "ax"]["hbm_bytes"], "profiles/" + name
                break
        except (OSError, KeyError, ValueError):
            continue
    out = {
        "metric": "genotype_matvec_GBps", "value": round(value, 2), "unit": "GB/s", "n_gpus": world,
        "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "i8 MFMA / i32 accumulate, 56-bit fixed point (fp64 in/out)" if a.mode == 1 else "f64",
        "data": "synthetic",
        "config": {"workload": "N=%d x Mt=%d 2-bit genotype matrix, step = lmmse_mult = Ax + N-vector all-reduce + ATx "
                               "(vamp.cpp:1074-1118)" % (N, Mt),
                   "markers_per_gpu": M, "kernel_mode": a.mode, "parallelism": "marker-sharded x%d" % world,
                   "resident_layout": ("fp64 raw rows" if a.mode == 0 else "two stripe sets, 2 x M*N/4 bytes" if layout == 1
                                       else "one tile layout, M*N/4 bytes"),
                   "engine": ("library defaults (no gv_set_kernel_mode / gv_set_layout call; gv_set_expected_passes(%d) = what this "
                              "run makes)" % planned_passes if engine_defaults else "--mode %d --layout %d" % (a.mode, a.layout)),
                   "ingest_s": round(t_ingest, 2), "ingest_alloc_s": round(t_alloc, 2), "ingest_fill_s": round(t_fill, 2),
                   "tune_s": round(tune_s, 3), "tune_source": tune_src},
        "roofline": {"bound": "hbm", "kernel": kname, "achieved": round(ax_gbps, 1), "peak": 8000.0, "unit": "GB/s",
                     "frac": round(ax_gbps / 8000.0, 4), "traffic": traffic, "traffic_source": traffic_src,
                     "alg_bytes_per_launch": shard_bytes, "avg_kernel_ms": round(ms_ax, 4),
                     # how avg_kernel_ms is measured: gv_set_timing(2) = HIP events recorded on the context's stream around EVERY
                     # streaming-kernel launch of the timed region (the events stay inside `value`'s clock: conservative by their
                     # ~2 us per launch); the rocprofv3 --kernel-trace --stats average of the same command is in profiles/
                     "timing": "HIP events around every streaming-kernel launch inside the timed region (gv_set_timing(2))",
                     # context ("also report vs measured copy bandwidth"): what plain streaming kernels reach
                     # on this box, measured now -- a read-only stream over the same resident stripes, and a copy
                     "read_stream_GBps": round(sh.read_bandwidth(1 << 30, 3), 1),
                     "copy_GBps_read_plus_write": round(sh.copy_bandwidth(1 << 30, 10), 1)},
        "kernels": {"ax": {"avg_ms": round(ms_ax, 4), "GBps": round(ax_gbps, 1), "launches": cnt["n_ax_kernel"]},
                    "atx": {"avg_ms": round(ms_atx, 4), "GBps": round(atx_gbps, 1), "launches": cnt["n_atx_kernel"]}},
        # the same two products through gv_ax / gv_atx on HOST pointers (this rank's shard; pageable numpy buffers, vectors over
        # PCIe both ways, for sharded jobs the all-reduce inside): algorithmic bytes / wall time per call
        "hostptr_GBps": round(2 * shard_bytes / (t_hax + t_hatx) / 1e9, 1),
        "hostptr": {"ax_ms": round(t_hax * 1e3, 3), "atx_ms": round(t_hatx * 1e3, 3),
                    "ax_GBps": round(shard_bytes / t_hax / 1e9, 1), "atx_GBps": round(shard_bytes / t_hatx / 1e9, 1)},
    }
    # what each rank ran: shard, picked decompositions, kernel and exchange times of the timed region (HIP events)
    mine = {"rank": rank, "markers": M, "first_marker": S, "ms_ax_kernel": round(ms_ax, 4), "ms_atx_kernel": round(ms_atx, 4),
            "ms_allreduce_per_ax": round(cnt["ms_allreduce"] / max(cnt["n_allreduce"], 1), 4), "decomposition": dec}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    if world > 1 or force_dist:
        out["multi_gpu"] = {"rccl_nranks": sh.L.gv_comm_size(sh.h), "rank0_numa_node": numa_node, "exchange": "ncclAllReduce(double, sum, %d) per Ax on the "
                            "context stream (data.cpp:928/:995)" % (4 * ((N + 255) // 256) * 64),
                            "ms_allreduce_per_ax": max(r["ms_allreduce_per_ax"] for r in per_rank), "per_rank": per_rank}
    else:
        out["decomposition"] = mine["decomposition"]
    if os.environ.get("GVAMP_FORCE_MULTI"):
        # gv_debug_force_multi (test hook): this one-rank run took the sharded branches over an in-stream exchange -- NOT a benchmark
        # configuration; what it shows is the cost of those branches at one rank (the exchange itself moves nothing between GPUs)
        out["forced_multi"] = {"GVAMP_FORCE_MULTI": os.environ["GVAMP_FORCE_MULTI"], "GV_OVERLAP": os.environ.get("GV_OVERLAP", "0"),
                               "ms_allreduce_per_ax": mine["ms_allreduce_per_ax"], "n_allreduce": cnt["n_allreduce"]}
    # ---- second half of the metric: VAMP iterations/s (vamp::infere of the host C++ mirror on the same shard) ----
    if a.vamp_iterations > 0:
        from gvamp_amd import hostapi
        p.free(); d.free()
        CV = max(1, Mt // 100)
        beta, y = hostapi.sim_phen(sh, 0.5, CV, 1, rank=rank)                 # sim.cpp recipe, h2 = 0.5, seed 1

        def vamp_leg(fuse):
            barrier()
            t1 = time.perf_counter()
            r = hostapi.infere_linear(sh, y, None, None, iterations=a.vamp_iterations, CG_max_iter=a.CG_max_iter, rho=0.5,
                                      seed=1, gam1=1e-8, gamw=2.0, true_signal=beta, history=False, rank=rank,
                                      fuse_solves=fuse)
            sh.synchronize()
            t_total = time.perf_counter() - t1
            its = r.trace
            tail = its[1:] if len(its) > 1 else its                           # iteration 1 has a cold CG
            secs = [t["seconds"] for t in tail]
            if world > 1:
                tt = torch.tensor([sum(secs)], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tot = float(tt.item())
            else:
                tot = sum(secs)
            return r, {
                "iters_per_s": round(len(tail) / tot, 4) if tot > 0 else None, "fuse_solves": fuse,
                "seconds_per_iter": [round(t["seconds"], 4) for t in its], "wall_s_all_iterations": round(t_total, 3),
                "n_ax": [t["n_ax"] for t in its], "n_atx": [t["n_atx"] for t in its],
                "n_ax_pass": [t["n_ax_pass"] for t in its], "n_atx_pass": [t["n_atx_pass"] for t in its],
                "cg_iters": [t["cg_iters"] for t in its], "onsager_iters": [t["onsager_iters"] for t in its],
                "R2_lmmse": [round(t["R2_lmmse"], 5) for t in its], "L_after": [t["L_after"] for t in its],
            }

        r, out["vamp"] = vamp_leg(a.fuse_solves)
        # what a user waits for: shard ingest (synthetic here: generated on the device) + picking the decompositions (0 when an
        # earlier run on this shape left them in the cache) + every VAMP iteration, the cold first one included
        out["vamp"]["time_to_solution_s"] = round(t_ingest + tune_s + out["vamp"]["wall_s_all_iterations"], 3)
        out["vamp"]["time_to_solution_parts"] = {"ingest_s": round(t_ingest, 3), "of_which_hipMalloc_s": round(t_alloc, 3),
                                                 "of_which_hidden_behind_source_prep_s": round(ingest_stats["overlap_s"], 3),
                                                 "resident_GB": round(ingest_stats["resident_GB"], 1),
                                                 "tune_s": round(tune_s, 3), "tune_source": tune_src,
                                                 "iterations_s": out["vamp"]["wall_s_all_iterations"]}
        out["vamp"]["config"] = (
            "sim.cpp phenotype (h2 0.5, CV %d, seed 1), default 23-component prior, rho 0.5, CG-max-iter %d, %d iterations; "
            "iters/s over iterations 2.., file output off; n_ax / n_atx = explicit vector products, n_*_pass = passes over "
            "the genotype shard.  fuse_solves %d (docs/history/rounds1-3.md section 5); `reference_sequence` = the same run issuing the "
            "reference's own sequence of products (fuse_solves 0), x_hat agreement between the two in `x_hat_rel_l2`"
            % (CV, a.CG_max_iter, len(r.trace), a.fuse_solves))
        if a.fuse_solves != 0:
            r0, v0 = vamp_leg(0)
            keep = ("iters_per_s", "seconds_per_iter", "n_ax", "n_atx", "n_ax_pass", "n_atx_pass", "cg_iters", "onsager_iters")
            out["vamp"]["reference_sequence"] = {k: v0[k] for k in keep}
            num = float(np.linalg.norm(r.x_est - r0.x_est)) ** 2
            den = float(np.linalg.norm(r0.x_est)) ** 2
            if world > 1:
                tt = torch.tensor([num, den], dtype=torch.float64)
                dist.all_reduce(tt)
                num, den = float(tt[0]), float(tt[1])
            out["vamp"]["x_hat_rel_l2"] = float(np.sqrt(num / den)) if den > 0 else None
            if a.fuse_solves >= 4:
                # level 4 (the default of the drivers and of this bench) is the only level that touches alpha2 (docs/history/rounds1-3.md section
                # 5): the level below it is measured beside it, with its own distance from the reference sequence
                r3, v3 = vamp_leg(3)
                num3 = float(np.linalg.norm(r3.x_est - r0.x_est)) ** 2
                if world > 1:
                    tt = torch.tensor([num3], dtype=torch.float64)
                    dist.all_reduce(tt)
                    num3 = float(tt[0])
                out["vamp"]["level_3"] = {k: v3[k] for k in ("iters_per_s", "seconds_per_iter", "n_ax_pass", "n_atx_pass")}
                out["vamp"]["level_3"]["x_hat_rel_l2"] = float(np.sqrt(num3 / den)) if den > 0 else None
    # ---- a harder setting beside it: block-correlated genotypes (LD), where the CG of the LMMSE step runs tens of steps -------
    if a.vamp_iterations > 0 and a.ld_block > 0:
        from gvamp_amd import hostapi
        t1 = time.perf_counter()
        sh.synth_bed(a.seed + 1, 5000, ld_block=a.ld_block, ld_ppm=a.ld_ppm)      # same shapes, same picks
        sh.compute_markers_statistics()
        sh.synchronize()
        t_ld_ingest = time.perf_counter() - t1
        CV = max(1, Mt // 100)
        beta, y = hostapi.sim_phen(sh, 0.5, CV, 1, rank=rank)
        def ld_run(fuse):
            barrier()
            t1 = time.perf_counter()
            r = hostapi.infere_linear(sh, y, None, None, iterations=a.vamp_iterations, CG_max_iter=a.CG_max_iter, rho=0.5, seed=1,
                                      gam1=1e-8, gamw=2.0, true_signal=beta, history=False, rank=rank, fuse_solves=fuse)
            sh.synchronize()
            return r, time.perf_counter() - t1

        r, t_total = ld_run(a.fuse_solves)
        its = r.trace
        tail = its[1:] if len(its) > 1 else its
        tot = sum(t["seconds"] for t in tail)
        if world > 1:
            tt = torch.tensor([tot], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tot = float(tt.item())
        npass = sum(t["n_ax_pass"] + t["n_atx_pass"] for t in tail)
        out["vamp_ld"] = {
            "config": "as `vamp`, on block-correlated genotypes: LD blocks of %d markers, within-block copy probability %.2f "
                      "(gv_synth_bed_ld; mean within-block genotype correlation ~0.6), so the CG is matvec-bound for tens of "
                      "steps as on real data" % (a.ld_block, a.ld_ppm / 1e6),
            "iters_per_s": round(len(tail) / tot, 4) if tot > 0 else None, "fuse_solves": a.fuse_solves,
            "seconds_per_iter": [round(t["seconds"], 4) for t in its], "wall_s_all_iterations": round(t_total, 3),
            "cg_iters": [t["cg_iters"] for t in its], "onsager_iters": [t["onsager_iters"] for t in its],
            "n_ax_pass": [t["n_ax_pass"] for t in its], "n_atx_pass": [t["n_atx_pass"] for t in its],
            "pass_GBps": round(npass * alg_bytes(N, Mt) / tot / 1e9, 1) if tot > 0 else None,
            "R2_lmmse": [round(t["R2_lmmse"], 5) for t in its], "ingest_s": round(t_ld_ingest, 2)}
        if a.fuse_solves != 0:
            # the same run issuing the reference's own sequence of products (level 0): the by-products of levels 2-4 must leave
            # the CG / Onsager / merge counts where they were and x_hat within rounding (tests/test_gpu_ld.py holds the same
            # against the oracle at sizes it can run)
            r0, t0_total = ld_run(0)
            num = float(np.linalg.norm(r.x_est - r0.x_est)) ** 2
            den = float(np.linalg.norm(r0.x_est)) ** 2
            if world > 1:
                tt = torch.tensor([num, den], dtype=torch.float64)
                dist.all_reduce(tt)
                num, den = float(tt[0]), float(tt[1])
            tail0 = r0.trace[1:] if len(r0.trace) > 1 else r0.trace
            tot0 = sum(t["seconds"] for t in tail0)
            out["vamp_ld"]["x_hat_rel_l2"] = float(np.sqrt(num / den)) if den > 0 else None
            out["vamp_ld"]["counts_equal_reference_sequence"] = bool(
                len(r.trace) == len(r0.trace) and all(t["cg_iters"] == u["cg_iters"] and t["onsager_iters"] == u["onsager_iters"]
                                                      and t["L_after"] == u["L_after"] and t["revar_rounds"] == u["revar_rounds"]
                                                      for t, u in zip(r.trace, r0.trace)))
            out["vamp_ld"]["reference_sequence"] = {
                "iters_per_s": round(len(tail0) / tot0, 4) if tot0 > 0 else None,
                "n_ax_pass": [t["n_ax_pass"] for t in r0.trace], "n_atx_pass": [t["n_atx_pass"] for t in r0.trace],
                "cg_iters": [t["cg_iters"] for t in r0.trace], "onsager_iters": [t["onsager_iters"] for t in r0.trace]}
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cb = cpu_baseline(N, a.seed, a.cpu_markers, local_rank)
        if "vamp" in out and len(out["vamp"]["n_ax"]) > 1:
            f = Mt / cb["sample_markers"]       # CPU matvec time is linear in the number of markers at fixed N
            v = out["vamp"].get("reference_sequence", out["vamp"])             # the reference's own matvec counts
            est = [(v["n_ax"][i] * cb["ax_s"] + v["n_atx"][i] * cb["atx_s"]) * f for i in range(1, len(v["n_ax"]))]
            cb["vamp_iter_s_extrapolated"] = round(sum(est) / len(est), 1)
            if cb.get("measured_over_model"):
                cb["vamp_iter_s_extrapolated_x_measured_over_model"] = round(sum(est) / len(est) * cb["measured_over_model"], 1)
            cb["vamp_note"] = ("seconds per VAMP iteration the CPU port would need for the matvec counts of the reference "
                               "sequence (vamp.reference_sequence): (n_ax*t_ax + n_atx*t_atx) measured on the sample, scaled "
                               "linearly by Mt/sample_markers; `vamp_iter_s_measured` holds REAL iterations of the port in this run "
                               "(config 1 whole; the sample slice at full N) and `measured_over_model` what that model missed on "
                               "the slice (denoiser, EM, vector updates)")
        out["cpu_baseline"] = cb
    sh.close()
    # ---- the same operator on the OTHER resident layout (main leg on the one tile layout: two stripe sets, 2 x M*N/4 bytes, announced
    # as a long run would announce itself; main leg on two stripe sets: the tile layout), measured in the same process on the same box,
    # and checked bit for bit against the main leg's result --------------------------------------------------------------------
    other = 1 if layout_everywhere == 2 else (2 if layout_everywhere == 1 else 0)
    if other == 1:
        free_b, total_b = torch.cuda.mem_get_info(local_rank)
        if 2.0 * M * ((N + 3) // 4) + 4e9 > 0.92 * free_b:
            other = 0                       # two stripe sets do not fit this GPU: no side leg
        if world > 1:
            t = torch.tensor([float(other)], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            other = int(t.item())
    if a.mode == 1 and other != 0 and not a.no_side_leg:
        barrier()
        # this leg always MEASURES its decompositions (no cache, no shipped table): its tune_s is what a cold first contact with a
        # shape costs on this box, whatever the main leg found in the cache or in gv_tune_builtin.h
        cold_env = {k: os.environ.get(k) for k in ("GV_TUNE_CACHE", "GV_TUNE_BUILTIN")}
        os.environ["GV_TUNE_CACHE"] = "0"
        os.environ["GV_TUNE_BUILTIN"] = "0"
        with capi.Shard(N, M, Mt=Mt, S=S, device=local_rank) as st:
            st.set_layout(False, other)
            st.set_kernel_mode(1)
            t1 = time.time()
            st.synth_bed(a.seed, 5000)
            st.compute_markers_statistics()
            t_in = time.time() - t1
            side_ingest = st.ingest_stats()
            if world > 1 or force_dist:
                uid = [capi.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                st.comm_init(world, rank, uid[0])
            p2, d2 = st.vecM(rng_p), st.vecM()
            for _ in range(2):
                st.lmmse_mult(p2, tau, gam2, d2)
            st.synchronize()
            tt_s, tt_src = st.tune_info()
            st.set_timing(2)
            st.counters(reset=True)
            barrier()
            t1 = time.perf_counter()
            nst = max(2, a.steps // 2)
            for _ in range(nst):
                st.lmmse_mult(p2, tau, gam2, d2)
            st.synchronize()
            barrier()
            dt2 = time.perf_counter() - t1
            if world > 1:
                t = torch.tensor([dt2], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt2 = float(t.item())
            c2 = st.counters()
            same = bool(np.array_equal(d2.download(), d_host))
            if world > 1:
                t = torch.tensor([1.0 if same else 0.0], dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                same = bool(t.item() == 1.0)
            m_ax = c2["ms_ax_kernel"] / max(c2["n_ax_kernel"], 1)
            m_atx = c2["ms_atx_kernel"] / max(c2["n_atx_kernel"], 1)
            one_gb, two_gb = M * ((N + 3) // 4) / 1e9, 2 * M * ((N + 3) // 4) / 1e9
            out["two_stripe_sets" if other == 1 else "tile_layout"] = {
                "what": ("the same step on TWO resident re-encodings, one per product (gv_set_layout(.., 1); what auto builds for a run "
                         "announced as >= 1000 ATx passes): resident genotype bytes per GPU %.1f GB instead of %.1f GB" % (two_gb, one_gb))
                        if other == 1 else
                        ("the same step on ONE resident re-encoding that serves Ax and ATx (docs/history/rounds1-3.md section 3): resident "
                         "genotype bytes per GPU %.1f GB instead of %.1f GB" % (one_gb, two_gb)),
                "value_GBps": round(job_bytes * nst / dt2 / 1e9, 2), "ms_per_step": round(dt2 / nst * 1e3, 4), "steps": nst,
                "ax": {"avg_ms": round(m_ax, 4), "GBps": round(shard_bytes / (m_ax * 1e-3) / 1e9, 1) if m_ax > 0 else None},
                "atx": {"avg_ms": round(m_atx, 4), "GBps": round(shard_bytes / (m_atx * 1e-3) / 1e9, 1) if m_atx > 0 else None},
                "bit_identical_to_main_leg": same, "ingest_s": round(t_in, 2), "ingest_alloc_s": round(side_ingest["alloc_s"], 2),
                "ingest_fill_s": round(side_ingest["fill_s"], 2), "resident_GB": round(side_ingest["resident_GB"], 1),
                "tune_s": round(tt_s, 3), "tune_source": tt_src,
                "decomposition": st.decomp()}
        for k, v in cold_env.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if rank == 0 and world == 1 and not a.no_rows:
        out["rows"] = run_rows(a, local_rank)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
