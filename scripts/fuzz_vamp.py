"""Randomised full VAMP runs against the oracle (development; run on a GPU box):   python scripts/fuzz_vamp.py [cases] [seed]
Per case: a random small shard (N, M around tile / block boundaries), missing genotypes, a random prior (2-5 components),
h2, rho, CG cap, 3-5 iterations; model linear / linear with --use-XXT-denoiser 1 / probit; kernel family (fp64 on raw rows, or
fixed point on two stripe sets / the tile layout, or kernel mode 2 -- the two-level fixed point -- at levels 0 and 4); --fuse-solves
0 ... 4.  Against the oracle's run of the same
configuration: x_hat to the north-star tolerance 1e-5 (relative l2), per-iteration CG and Onsager step counts, prior after
EM.  Between fuse levels and layouts of the product: fixed-point layouts bit-identical, fuse levels to 1e-9."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi, hostapi, synth
from oracle import gvoracle as oracle

oracle.lib()
TOL = 1e-5        # north-star tolerance on x_hat (relative l2); the fixed-shape tests hold 1e-7
EDGE_N = [200, 255, 256, 257, 511, 513, 1000, 1023, 1025]
EDGE_M = [63, 64, 65, 127, 129, 255, 257, 1000, 1023, 1025, 2049]


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def pick(rng, edges, lo, hi):
    return int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(lo, hi))


_group = [int(time.time()) % 100000 * 10 + 500000]


def run_sharded(N, M, bed, cuts, layout, y, beta, probs, vars_, kw, extra, overlap):
    """the same run on in-process marker shards (one thread per rank, host communicator)"""
    nr = len(cuts) - 1
    out, errors = [None] * nr, []
    _group[0] += 1
    group = _group[0]
    mb = (N + 3) // 4

    def work(rank):
        try:
            S, Ms = cuts[rank], cuts[rank + 1] - cuts[rank]
            with capi.Shard(N, Ms, Mt=M, S=S) as sh:
                sh.set_layout(False, layout)
                sh.set_kernel_mode(1)
                sh.upload_bed(bed[S * mb:(S + Ms) * mb])
                sh.comm_init_local(group, nr, rank)
                sh.set_overlap(overlap)
                out[rank] = hostapi.infere_linear(sh, y, probs, vars_, true_signal=beta[S:S + Ms], fuse_solves=2 + (group % 3), rank=rank,
                                                  **kw, **extra)
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nr)]
    for t in th:
        t.start()
    t_end = time.time() + 60
    for t in th:
        t.join(timeout=max(0.1, t_end - time.time()))
    if any(t.is_alive() for t in th):     # its peers left the sequence of collectives: nothing after this is trustworthy
        raise RuntimeError("a rank is stuck in a collective; errors so far: %r" % (errors,))
    assert not errors, ("rank failed", errors)
    return out


def amplification(trace):
    amp = 1.0
    for o in trace:
        for big, small in ((o["eta2"], o["gam1_next"]), (o["eta1"], o["gam2"])):
            if small > 0 and np.isfinite(big / small):
                amp = max(amp, abs(big / small))
    return amp


def run_case(seed0, k):
    rng = np.random.default_rng(seed0 * 100003 + k)
    N, M = pick(rng, EDGE_N, 200, 2500), pick(rng, EDGE_M, 60, 4000)
    miss = int(rng.choice([0, 5000, 30000]))
    bed = synth.synth_bed(N, M, seed=int(rng.integers(1 << 30)), miss_ppm=miss)
    L = int(rng.integers(2, 6))
    p0 = float(rng.uniform(0.5, 0.95))
    w = rng.random(L - 1) + 0.1
    probs = [p0] + list((1 - p0) * w / w.sum())
    vars_ = [0.0] + sorted(float(10.0 ** rng.uniform(-4, -1.5)) for _ in range(L - 1))
    model = str(rng.choice(["linear", "linear", "xxt", "probit"]))
    kw = dict(iterations=int(rng.integers(3, 6)), CG_max_iter=int(rng.choice([5, 20, 40])), rho=float(rng.choice([0.15, 0.5, 0.9])),
              seed=int(rng.integers(1, 50)), gam1=float(rng.choice([1e-8, 1e-6, 1e-3])), gamw=float(rng.choice([1.0, 2.0])),
              learn_vars=int(rng.integers(2)), EM_max_iter=int(rng.choice([1, 2, 5])))
    h2, CV = float(rng.uniform(0.2, 0.8)), max(1, int(M * rng.uniform(0.01, 0.2)))
    sseed = int(rng.integers(1, 100))
    info = dict(N=N, M=M, miss=miss, L=L, model=model, h2=round(h2, 3), CV=CV, **kw)
    extra = {}
    if model == "xxt":
        extra["use_XXT_denoiser"] = 1
    if model == "probit":
        extra["model"] = "bin_class"
    runs = {}
    y = None
    for mode, layout in ((1, 1), (1, 2), (0, 0), (2, 2)):        # (2, 2): kernel mode 2, the two-level fixed point, on the tile layout
        with capi.Shard(N, M) as sh:
            if mode != 0:
                sh.set_layout(False, layout)
            else:
                sh.set_layout(True, False)
            sh.set_kernel_mode(mode)
            sh.upload_bed(bed)
            if y is None:
                beta, yy = hostapi.sim_phen(sh, h2, CV, sseed)
                y = (yy > 0).astype(float) if model == "probit" else yy
            for fuse in ((0, 1, 2, 3, 4) if mode == 1 and layout == 1 else (2, 4) if mode == 1 else (0, 4) if mode == 2 else (0,)):
                runs[(mode, layout, fuse)] = hostapi.infere_linear(sh, y, probs, vars_, true_signal=beta, fuse_solves=fuse,
                                                                   **kw, **extra)
    ref = oracle.infere(bed, N, M, y, probs, vars_, true_signal=beta, **kw, **extra)
    base = runs[(1, 1, 0)]
    assert np.all(np.isfinite(ref.x_est)), ("oracle not finite", info)
    # How much does VAMP itself amplify a rounding difference?  gam1_next = eta2 - gam2 (vamp.cpp:702) and gam2 = eta1 - gam1 (:472)
    # are differences of nearly equal numbers when alpha2 / alpha1 sit next to 1 (gam1 = 1e-8 in iteration 1; no signal; N << M):
    # a relative rounding difference eps in alpha comes out as eps * eta / (eta - gam).  Two correct implementations that add
    # in another order (the oracle and the product: eps ~ 1e-13 on sums of thousands of terms) differ by that much, so the
    # tolerance follows it -- at the north-star 1e-5 for every run whose amplification stays below 1e8.
    amp = amplification(ref.trace)
    info["amp"] = float("%.2g" % amp)
    tol_run = max(TOL, 1e-13 * amp)
    for key, r in runs.items():
        assert r.niter == len(ref.trace), ("iterations run", key, info, r.niter, len(ref.trace))
        e = rel(r.x_est, ref.x_est)
        assert e < tol_run, ("x_hat vs oracle", key, info, e, tol_run)
        for it in range(r.niter):
            t, o = r.trace[it], ref.trace[it]
            assert (t["cg_iters"], t["onsager_iters"], t["L_after"]) == (o["cg_iters"], o["onsager_iters"], o["L_after"]), \
                ("step counts", key, it, info, (t["cg_iters"], t["onsager_iters"], t["L_after"]),
                 (o["cg_iters"], o["onsager_iters"], o["L_after"]))
            assert np.isclose(t["gamw"], o["gamw"], rtol=1e-5), ("gamw", key, it, info, t["gamw"], o["gamw"])
    assert np.array_equal(runs[(1, 1, 2)].x_est, runs[(1, 2, 2)].x_est), ("layouts differ", info)
    assert np.array_equal(runs[(1, 1, 4)].x_est, runs[(1, 2, 4)].x_est), ("layouts differ at fuse 4", info)
    for fuse in (1, 2, 3, 4):
        # levels 1 to 3 leave the Onsager solve bit-identical (alpha2 to the last bit); level 4 takes its first operator
        # application from A^T A u of the probe, a rounding-level change of alpha2 that the run amplifies as above
        tol_f = 1e-8 if fuse < 4 else max(1e-8, 1e-15 * amp)
        e = rel(runs[(1, 1, fuse)].x_est, base.x_est)
        assert e < tol_f, ("fuse level", fuse, info, e, tol_f)
    if rng.random() < 0.5:     # marker shards as divide_work cuts them (the Onsager probe is seeded per shard: utilities.cpp:259-291)
        nr = int(rng.integers(2, 4))
        cuts = [0]
        for r in range(nr):
            cuts.append(cuts[-1] + oracle.divide_work(M, nr, r)[0])
        info["cuts"] = cuts
        ref_s = oracle.infere(bed, N, M, y, probs, vars_, true_signal=beta, nshards=nr, **kw, **extra)
        sh_runs = run_sharded(N, M, bed, cuts, int(rng.integers(1, 3)), y, beta, probs, vars_, kw, extra, int(rng.choice([0, 3])))
        xs = np.concatenate([r.x_est for r in sh_runs])
        tol_s = max(TOL, 1e-13 * amplification(ref_s.trace))
        assert rel(xs, ref_s.x_est) < tol_s, ("sharded x_hat vs oracle", info, rel(xs, ref_s.x_est), tol_s)
        for it in range(len(ref_s.trace)):
            o = ref_s.trace[it]
            for r in sh_runs:
                t = r.trace[it]
                assert (t["cg_iters"], t["onsager_iters"], t["L_after"]) == (o["cg_iters"], o["onsager_iters"], o["L_after"]), \
                    ("sharded step counts", it, info)
    return info


def main(ncases, seed):
    t0 = time.time()
    bad = []
    for k in range(ncases):
        try:
            info = run_case(seed, k)
        except AssertionError as e:
            bad.append((k, str(e)))
            print("CASE %d FAILED: %s" % (k, e), flush=True)
            continue
        if k % 5 == 0:
            print("case %d ok %s  (%.0f s)" % (k, info, time.time() - t0), flush=True)
    print("%d cases, %d failed, %.0f s" % (ncases, len(bad), time.time() - t0))
    return bad


if __name__ == "__main__":
    if len(sys.argv) > 4 and sys.argv[3] == "only":      # python scripts/fuzz_vamp.py <n> <seed> only <case>: replay one case
        print(run_case(int(sys.argv[2]), int(sys.argv[4])))
        sys.exit(0)
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
