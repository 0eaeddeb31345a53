"""Parity on block-correlated genotypes -- the regime real data lives in.  gv_synth_bed_ld (LD blocks of 64 markers, within-block
copy probability 0.9, mean within-block correlation ~0.6) makes the LMMSE CG (vamp.cpp:1130-1229) run tens of steps instead of
4-5 and gives A^T A the spread of eigenvalues on which the by-products of --fuse-solves 2-4 (products taken from CG recurrences
and residuals, vamp.cpp:871-889 / :892-927 / :1142-1145) and the capture rule of level 4 (vamp::probe_product_is_usable) have
to hold.  Everything here is product vs ORACLE (the oracle issues the reference's own sequence of explicit products): identical
CG / Onsager / re-estimation / merge counts and x_hat within 1e-7 iteration by iteration, at every fuse level, in both resident
layouts, on one shard and on 2-3 in-process marker shards."""
import threading

import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
LD = dict(ld_block=64, ld_ppm=900000)
PROBS, VARS = [0.90, 0.07, 0.03], [0, 0.001, 0.01]
TIGHT = 1e-7


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def test_ld_generator_device_matches_host_bit_for_bit():
    """gv_synth_bed_ld on the device against gvamp_amd.synth.synth_bed(ld_block=...) on the host, shard offsets included (the
    blocks are cut on GLOBAL marker indices: a shard that starts inside a block continues it)."""
    for N, M, S, blk, ppm in ((2000, 300, 0, 64, 900000), (1003, 200, 37, 64, 900000), (515, 129, 1000, 8, 500000), (64, 70, 5, 64, 1000000)):
        host = synth.synth_bed(N, M, seed=4321, miss_ppm=7000, S=S, ld_block=blk, ld_ppm=ppm)
        with capi.Shard(N, M, Mt=S + M + 5, S=S) as sh:
            sh.set_layout(True, 1)
            sh.synth_bed(4321, 7000, ld_block=blk, ld_ppm=ppm)
            assert np.array_equal(sh.download_bed(), host), (N, M, S)
    # and the columns really are correlated inside a block, not across blocks
    N, M = 4000, 256
    bed = synth.synth_bed(N, M, seed=1, miss_ppm=0, **LD).reshape(M, N // 4)
    codes = np.stack([(bed >> (2 * k)) & 3 for k in range(4)], axis=2).reshape(M, N)
    a = np.where(codes == 0, 2.0, np.where(codes == 2, 1.0, 0.0))
    c = np.corrcoef(a)
    inside = np.mean([abs(c[i, j]) for i in range(0, 64) for j in range(i + 1, 64)])
    across = np.mean([abs(c[i, j]) for i in range(0, 64) for j in range(64, 128)])
    assert inside > 0.3 and across < 0.05, (inside, across)


def _check_run(r, ref, niter, what):
    assert r.niter == ref.niter == niter, what
    for it in range(niter):
        t, o = r.trace[it], ref.trace[it]
        assert (t["cg_iters"], t["onsager_iters"], t["L_after"], t["revar_rounds"]) == \
               (o["cg_iters"], o["onsager_iters"], o["L_after"], o["revar_rounds"]), (what, it, t, o)
        assert np.isclose(t["gamw"], o["gamw"], rtol=1e-6) and np.isclose(t["alpha2"], o["alpha2"], rtol=1e-6), (what, it)
        assert rel(r.x1[it], ref.x1[it]) < TIGHT and rel(r.x2[it], ref.x2[it]) < TIGHT, (what, it, rel(r.x1[it], ref.x1[it]))
    assert rel(r.x_est, ref.x_est) < TIGHT, what


@pytest.mark.parametrize("N,M,layout", [(2000, 5000, 1), (2000, 5000, 2), (3001, 2500, 1), (1500, 9984, 2)])
def test_ld_runs_follow_the_oracle_at_every_fuse_level(oracle, N, M, layout):
    """six iterations on LD genotypes (CG 15-40 steps per iteration): levels 0-4 against the oracle"""
    bed = synth.synth_bed(N, M, seed=77, miss_ppm=5000, **LD)
    kw = dict(iterations=6, CG_max_iter=50, rho=0.5, seed=9, gam1=1e-8, gamw=2.0, stop_criteria_thr=1e-12)
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, layout)
        sh.upload_bed(bed)
        beta, y = hostapi.sim_phen(sh, 0.5, max(1, M // 50), 9)
        runs = {f: hostapi.infere_linear(sh, y, PROBS, VARS, true_signal=beta, fuse_solves=f, **kw) for f in range(5)}
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, true_signal=beta, **kw)
    assert max(t["cg_iters"] for t in ref.trace) >= 12, [t["cg_iters"] for t in ref.trace]     # the regime the test is about
    for f, r in runs.items():
        _check_run(r, ref, 6, "fuse %d" % f)
    # levels 0 and 1 issue the same arithmetic (shared passes, exact integer accumulation): bit for bit
    assert all(np.array_equal(runs[1].x1[it], runs[0].x1[it]) for it in range(6))
    # passes: each level needs no more than the one below it, and level 4 needs fewer than level 0 by a wide margin
    passes = {f: sum(t["n_ax_pass"] + t["n_atx_pass"] for t in r.trace) for f, r in runs.items()}
    assert passes[4] <= passes[3] <= passes[2] <= passes[1] < passes[0], passes
    # level 4's capture rule went both ways in this very run: the product of iteration 1 (gam2 / tau in the thousands at
    # gam1 = 1e-8) was dropped, a later one kept, and from then on the kept product was used
    states = [t["probe_product"] for t in runs[4].trace]
    assert states[0] in (1, 2) and 1 in states and states[-1] == 3, states       # captured once, then used
    assert all(t["probe_product"] == 0 for t in runs[3].trace)


def test_level_4_capture_rule_goes_both_ways_inside_a_run(oracle):
    """A prior whose only sizeable slab has a tiny weight makes gam2 / tau of the first iteration (gam1 = 1e-8) tens of thousands
    (gam2 ~ 1 / E[v]) and of order one from the second on -- the situation of the headline run with the default 23-component
    prior: the product captured in iteration 1 is dropped (state 2), iteration 2 captures again and keeps it (1), later
    iterations use it (3); counts and x_hat still follow the oracle."""
    N, M = 2000, 5000
    bed = synth.synth_bed(N, M, seed=77, miss_ppm=5000, **LD)
    probs, vars_ = [0.9, 0.099999, 1e-6], [0, 1e-9, 5e-3]
    kw = dict(iterations=5, CG_max_iter=50, rho=0.5, seed=9, gam1=1e-8, gamw=2.0, stop_criteria_thr=1e-12)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        beta, y = hostapi.sim_phen(sh, 0.5, 100, 9)
        r4 = hostapi.infere_linear(sh, y, probs, vars_, true_signal=beta, fuse_solves=4, **kw)
    ref = oracle.infere(bed, N, M, y, probs, vars_, true_signal=beta, **kw)
    states = [t["probe_product"] for t in r4.trace]
    assert states[0] == 2 and states[1] == 1 and states[2:] == [3, 3, 3], (states, [t["gam2"] for t in r4.trace])
    assert r4.trace[0]["gam2"] > 5e4 and r4.trace[1]["gam2"] < 100.0
    _check_run(r4, ref, 5, "fuse 4, tiny-variance prior")


@pytest.mark.parametrize("nshards,layouts", [(2, (1, 1)), (3, (2, 2, 2)), (2, (1, 2)), (3, (2, 1, 2))])
@pytest.mark.parametrize("fuse", [0, 2, 4])
def test_ld_sharded_runs_follow_the_oracle(oracle, nshards, layouts, fuse):
    """the same on 2-3 in-process marker shards (divide_work, probe seeded seed + S, one N-vector sum per Ax), the ranks of one job
    holding the same or DIFFERENT resident layouts -- what gv_set_layout(.., 3) may decide per rank -- against the oracle's
    run on the same shards"""
    N, Mt = 2000, 4097
    bed = synth.synth_bed(N, Mt, seed=78, miss_ppm=5000, **LD)
    mb = (N + 3) // 4
    kw = dict(iterations=6, CG_max_iter=50, rho=0.5, seed=9, gam1=1e-8, gamw=2.0, stop_criteria_thr=1e-12)
    beta, y = oracle.sim_phen(bed, N, Mt, 0.5, 80, 9)
    ref = oracle.infere(bed, N, Mt, y, PROBS, VARS, true_signal=beta, nshards=nshards, **kw)
    out, errors = [None] * nshards, []
    group = 9100 + 10 * nshards + fuse + 100 * sum(layouts)

    def work(rank):
        try:
            size, modu = divmod(Mt, nshards)
            M = size + 1 if rank < modu else size
            S = sum(size + 1 if r < modu else size for r in range(rank))
            with capi.Shard(N, M, Mt=Mt, S=S) as sh:
                sh.set_layout(False, layouts[rank])
                sh.upload_bed(bed[S * mb:(S + M) * mb])
                sh.comm_init_local(group, nshards, rank)
                out[rank] = (S, M, hostapi.infere_linear(sh, y, PROBS, VARS, true_signal=beta[S:S + M], rank=rank,
                                                         fuse_solves=fuse, **kw))
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nshards)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not any(t.is_alive() for t in th), "a rank is stuck in a collective"
    assert not errors, errors
    for it in range(6):
        x1 = np.concatenate([o[2].x1[it] for o in out])
        assert rel(x1, ref.x1[it]) < TIGHT, (it, rel(x1, ref.x1[it]))
        for o in out:
            t, q = o[2].trace[it], ref.trace[it]
            assert (t["cg_iters"], t["onsager_iters"], t["L_after"]) == (q["cg_iters"], q["onsager_iters"], q["L_after"]), it


def test_ld_overlapped_exchange_with_mixed_layouts_is_bit_identical():
    """gv_set_overlap on ranks that hold DIFFERENT resident layouts (two stripe sets on one, the tile layout on the other -- what
    the per-rank auto decision can produce): the slices of the N-vector are cut from N alone (units of 1024 individuals), so
    both ranks exchange the same ranges; results equal the one-message form bit for bit.  N = 10300 is the shape on which slices
    derived from the layout's own row groups (64 vs 256 rows) disagreed: 3328 vs 3072 individuals in the first of three."""
    N, Mt = 10300, 3000
    bed = synth.synth_bed(N, Mt, seed=5, miss_ppm=8000, **LD)
    mb = (N + 3) // 4
    rng = np.random.default_rng(3)
    x, x2 = rng.standard_normal(Mt), rng.standard_normal(Mt)

    def run(tiles):
        out, errors = [None] * 2, []

        def work(rank):
            try:
                M = Mt // 2
                S = rank * M
                with capi.Shard(N, M, Mt=Mt, S=S) as sh:
                    sh.set_layout(False, 1 + rank)                  # rank 0: two stripe sets, rank 1: the tile layout
                    sh.upload_bed(bed[S * mb:(S + M) * mb])
                    sh.comm_init_local(9500 + tiles, 2, rank)
                    sh.set_overlap(tiles)
                    sh.compute_markers_statistics()
                    z = sh.Ax(x[S:S + M])
                    xa, xb, za, zb = sh.vecM(x[S:S + M]), sh.vecM(x2[S:S + M]), sh.vecN(), sh.vecN()
                    sh.ax2_dev(xa, xb, za, zb)
                    mu = sh.vecM()
                    st, rr = sh.cg_solve(xa, None, 2.0, 0.8, 1, 40, mu)
                    out[rank] = (z, za.download(), zb.download(), mu.download(), rr)
            except Exception as e:   # noqa: BLE001
                errors.append((rank, repr(e)))

        th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(2)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=300)
        assert not any(t.is_alive() for t in th), "a rank is stuck in a collective (mismatched slices?)"
        assert not errors, errors
        return out

    plain = run(0)
    for tiles in (3, 7):
        ovl = run(tiles)
        for rank in range(2):
            for a, b in zip(plain[rank], ovl[rank]):
                assert np.array_equal(a, b), (tiles, rank)
    assert np.array_equal(plain[0][0], plain[1][0])


def test_capture_rule_separates_accurate_from_cancelled_products(oracle):
    """What vamp::probe_product_is_usable decides on, measured on LD genotypes through the C ABI: A^T A u captured from the first
    application of the zero-started solve is (diag / tau) d - (gam2 / tau) u.  With gam2 / tau of order |A^T A u| it agrees with
    the explicit product (oracle: Ax then ATx) to 1e-12; with gam2 / tau thousands of times larger -- the first VAMP iteration
    at gam1 = 1e-8 -- it has lost those digits.  The rule's threshold (gam2 <= 1e3 tau |A^T A u|) keeps the first and drops the
    second."""
    N, M = 2000, 3000
    rng = np.random.default_rng(11)
    bed = synth.synth_bed(N, M, seed=79, miss_ppm=5000, **LD)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    o_mave, o_msig = oracle.marker_stats(bed, N, M)
    explicit = oracle.atx(bed, N, M, o_mave, o_msig, oracle.ax(bed, N, M, o_mave, o_msig, u))
    nrm = float(np.linalg.norm(explicit))
    errs = {}
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        du, dv = sh.vecM(u), sh.vecM(rng.standard_normal(M))
        for tau, gam2 in ((2.0, 1.3), (2.0, 2.0 * nrm * 5e5)):
            cap, mu_a, mu_b = sh.vecM(), sh.vecM(), sh.vecM()
            sh.cg_solve2x(dv, None, du, tau, gam2, 30, mu_a, mu_b, ata_v_b=cap, have_ata_v_b=False)
            errs[gam2 <= 1e3 * tau * nrm] = rel(cap.download(), explicit)
    assert errs[True] < 1e-12, errs            # kept by the rule: as good as the explicit product
    assert errs[False] > 1e-11, errs           # dropped by the rule: five to six digits gone
