"""--use-XXT-denoiser 1 (denoiserXXT.cpp; BASELINE config 5, matrix-free form): people statistics, the N-space CG and a
full run of the product against the CPU oracle."""
import os

import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
PROBS, VARS = [0.9, 0.1], [0, 0.01]


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_people_statistics_vs_oracle(oracle):
    N, M = 1003, 700
    rng = np.random.default_rng(1)
    bed = synth.synth_bed(N, M, seed=71, miss_ppm=15000)
    present = rng.random(N) >= 0.02
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(present.sum())
    o = oracle.people_stats(bed, N, M, mask4=m4, nonas=nonas)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_mask(m4, nonas)
        sh.compute_markers_statistics()
        g = sh.compute_people_statistics()
    for a, b in zip(g, o):
        assert np.allclose(a[:N], b[:N], rtol=1e-10, atol=1e-13)
    assert np.all(g[0][:N][~present] == 0) and np.all(g[1][:N][~present] == 0)


@pytest.mark.parametrize("N,M,miss", [(1003, 700, 15000), (257, 3001, 0), (4099, 1500, 200000), (64, 256, 5000)])
def test_people_statistics_from_stripes(oracle, N, M, miss):
    """Kernel mode 1 / stripes-only layout: the three per-individual sums come from four fixed-point passes over
    stripes_n (one of them on the a^2 plane of the codes, MODE 4) instead of three fp64 passes over the raw rows."""
    rng = np.random.default_rng(N)
    bed = synth.synth_bed(N, M, seed=75, miss_ppm=miss)
    present = rng.random(N) >= 0.03
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(present.sum())
    o = oracle.people_stats(bed, N, M, mask4=m4, nonas=nonas)
    with capi.Shard(N, M, anchor=True) as sh:          # both layouts: mode 0 reads the raw rows, mode 1 the stripes
        sh.upload_bed(bed)
        sh.set_mask(m4, nonas)
        sh.compute_markers_statistics()
        g0 = sh.compute_people_statistics()
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        g1 = sh.compute_people_statistics()
    with capi.Shard(N, M) as sh:          # stripes only
        sh.set_layout(False, True)
        sh.upload_bed(bed)
        sh.set_mask(m4, nonas)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        g2 = sh.compute_people_statistics()
    for a, b, c, d in zip(g1, g2, g0, o):
        assert np.array_equal(a, b)
        assert np.allclose(a[:N], c[:N], rtol=1e-10, atol=1e-13)
        assert np.allclose(a[:N], d[:N], rtol=1e-10, atol=1e-13)
    assert np.array_equal(g1[2][:N][present], g0[2][:N][present])          # the counts are integers: exact
    assert np.all(g1[0][:N][~present] == 0) and np.all(g1[1][:N][~present] == 0)


@pytest.mark.parametrize("mode,fuse", [(0, 1), (1, 0), (1, 1), (1, 2)])
def test_xxt_run_vs_oracle(oracle, mode, fuse):
    N, M = 600, 2000
    bed = synth.synth_bed(N, M, seed=72, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 60, 3)
    ref = oracle.infere(bed, N, M, y, PROBS, VARS, iterations=3, CG_max_iter=40, rho=0.5, seed=3, true_signal=beta,
                        use_XXT_denoiser=1)
    std = oracle.infere(bed, N, M, y, PROBS, VARS, iterations=3, CG_max_iter=40, rho=0.5, seed=3, true_signal=beta)
    with capi.Shard(N, M, anchor=(mode == 0)) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(mode)
        r = hostapi.infere_linear(sh, y, PROBS, VARS, iterations=3, CG_max_iter=40, rho=0.5, seed=3, true_signal=beta,
                                  use_XXT_denoiser=1, fuse_solves=fuse)
    assert [t["cg_iters"] for t in r.trace] == [int(t["cg_iters"]) for t in ref.trace]
    assert [t["onsager_iters"] for t in r.trace] == [int(t["onsager_iters"]) for t in ref.trace]
    for a, b in zip(r.trace, ref.trace):
        for f in ("alpha2", "gamw", "gam1_next"):
            assert np.isclose(a[f], b[f], rtol=1e-6), f
    assert rel(r.x_est, ref.x_est) < 1e-7
    assert rel(r.x2[2], ref.x2[2]) < 1e-7
    # Woodbury: the N-space solve gives the M-space LMMSE estimate up to the two CG tolerances (1e-4 / 1e-5)
    assert rel(ref.x_est, std.x_est) < 5e-3


@pytest.mark.parametrize("warm", [False, True])
def test_joint_nspace_and_onsager_solves_equal_the_separate_ones(warm):
    """gv_cg_solve_aat2: the N-space solve and the M-space Onsager solve run half an application out of phase on shared
    passes.  Bit-identical to gv_cg_solve_aat + gv_cg_solve + an ATx; about half the passes; by-products to rounding."""
    N, M = 1200, 2600
    rng = np.random.default_rng(6)
    bed = synth.synth_bed(N, M, seed=73, miss_ppm=5000)
    npad = 4 * ((N + 3) // 4)
    v = np.zeros(npad)
    v[:N] = rng.standard_normal(N)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    mu0 = np.zeros(npad)
    mu0[:N] = 0.05 * rng.standard_normal(N)
    tau, gam2 = 2.0, 0.8
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        sh.compute_people_statistics()
        dv, du = sh.vecN(v), sh.vecM(u)
        dm0 = sh.vecN(mu0) if warm else None
        a1, b1, a2, b2, at1, at2 = sh.vecN(), sh.vecM(), sh.vecN(), sh.vecM(), sh.vecM(), sh.vecM()
        aat, ata = sh.vecN(), sh.vecM()
        sh.counters(reset=True)
        sa, ra = sh.cg_solve_aat(dv, dm0, tau, gam2, 30, a1)
        sb, rb = sh.cg_solve(du, None, tau, gam2, 0, 30, b1)
        sh.atx_dev(a1, at1)
        c1 = sh.counters(reset=True)
        (s2a, r2a), (s2b, r2b) = sh.cg_solve_aat2(dv, dm0, du, tau, gam2, 30, a2, at2, b2, aat_mu_a=aat, ata_mu_b=ata)
        c2 = sh.counters()
        assert (s2a.iters, s2b.iters) == (sa.iters, sb.iters) and sa.iters > 1 and sb.iters > 1
        assert np.array_equal(r2a, ra) and np.array_equal(r2b, rb)
        assert np.array_equal(a2.download(), a1.download()) and np.array_equal(b2.download(), b1.download())
        assert np.array_equal(at2.download(), at1.download())
        assert c2["n_ax"] == c1["n_ax"] and c2["n_atx"] == c1["n_atx"]            # the same products ...
        p1, p2 = c1["n_ax_pass"] + c1["n_atx_pass"], c2["n_ax_pass"] + c2["n_atx_pass"]
        napp = max(2 * (sa.iters + (1 if warm else 0)) + 1, 2 * sb.iters) + 1
        assert p2 <= napp and p2 < 0.7 * p1, (p1, p2)                              # ... in about half the passes
        mu_a, mu_b = a2.download(), b2.download()
        assert rel(aat.download(), sh.Ax(sh.ATx(mu_a))) < 1e-10
        assert rel(ata.download(), sh.ATx(sh.Ax(mu_b))) < 1e-10


@pytest.mark.parametrize("fuse", [0, 2, 4])
def test_xxt_sharded_run_vs_oracle(oracle, fuse):
    """--use-XXT-denoiser 1 on two marker shards (in-process communicator): people statistics all-reduced over the
    shards (data.cpp:604-606), N-space vectors replicated, M-space Onsager solve sharded."""
    import threading
    N, Mt, nshards = 500, 1800, 2
    bed = synth.synth_bed(N, Mt, seed=74, miss_ppm=5000)
    mb = (N + 3) // 4
    beta, y = oracle.sim_phen(bed, N, Mt, 0.5, 50, 4)
    kw = dict(iterations=3, CG_max_iter=40, rho=0.5, seed=4, use_XXT_denoiser=1)
    ref = oracle.infere(bed, N, Mt, y, PROBS, VARS, nshards=nshards, true_signal=beta, **kw)
    results, errors = [None] * nshards, []

    def work(rank):
        try:
            size, modu = divmod(Mt, nshards)
            M = size + 1 if rank < modu else size
            S = sum(size + 1 if r < modu else size for r in range(rank))
            with capi.Shard(N, M, Mt=Mt, S=S) as sh:
                sh.upload_bed(bed[S * mb:(S + M) * mb])
                sh.set_kernel_mode(1)
                sh.comm_init_local(4200 + fuse, nshards, rank)
                results[rank] = hostapi.infere_linear(sh, y, PROBS, VARS, true_signal=beta[S:S + M], rank=rank,
                                                      fuse_solves=fuse, **kw)
        except Exception as e:   # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,), daemon=True) for r in range(nshards)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    assert not errors, errors
    x = np.concatenate([r.x_est for r in results])
    assert rel(x, ref.x_est) < 1e-7
    assert [t["cg_iters"] for t in results[0].trace] == [int(t["cg_iters"]) for t in ref.trace]
    assert all([t["gamw"] for t in r.trace] == [t["gamw"] for t in results[0].trace] for r in results)


@pytest.mark.parametrize("N,M", [(300032, 6000), (3000, 9000)])
def test_pipelined_joint_solver_vs_its_unfused_launches_and_the_host_paced_loop(N, M):
    """gv_cg_solve_aat2w in the form the VAMP loop uses (A^T mu accumulated, right-hand side completed inside, a rider): its steady
    state is enqueued ahead of its statuses, a step's reductions are added up by their consumers and the Ax epilogue takes the
    first update of the N-space step along.  Against (a) the same pipeline with every fused piece as a launch of its own -- what a
    sharded job runs: k_finalize + exchange between a reduction and its consumer, k_aat_dq behind the exchanged product, the
    rider copied out of its slot (gv_debug_force_multi) -- every output and every counter BIT FOR BIT; (b) the host-paced loop with
    host-side scalars (GV_CG_DEVICE=0): the same counts, outputs to rounding.  N = 300 032 is above the 262 144 entries a reduction
    covers with one entry per thread: there the block partials run over 1024 blocks and the epilogue fusion is off by construction."""
    rng = np.random.default_rng(N + M)
    npad = 4 * ((N + 3) // 4)
    v = np.zeros(npad)
    v[:N] = rng.standard_normal(N)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    x1, r2 = rng.standard_normal(M) * 0.1, rng.standard_normal(M) * 0.1
    tau, gam2 = 1.5, 0.6
    keys = ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass")
    outs = []
    with capi.Shard(N, M) as sh:
        sh.synth_bed(91, 5000)
        sh.compute_markers_statistics()
        sh.compute_people_statistics()
        du, dx1, dr2 = sh.vecM(u), sh.vecM(x1), sh.vecM(r2)
        for mode in ("pipelined", "unfused", "host"):
            env = {"GV_CG_DEVICE": "0"} if mode == "host" else {}
            for k, val in env.items():
                os.environ[k] = val
            if mode == "unfused":
                sh.force_multi(1)
            try:
                dv = sh.vecN(v)
                mu, at, mb, aat, ata, ro, po = sh.vecN(), sh.vecM(), sh.vecM(), sh.vecN(), sh.vecM(), sh.vecN(), sh.vecN()
                sh.counters(reset=True)
                (sa, ra), (sb, rb) = sh.cg_solve_aat2(dv, None, du, tau, gam2, 12, mu, at, mb, aat_mu_a=aat, ata_mu_b=ata,
                                                      accumulate_at_mu_a=True, pre_x=dr2, pre_out=po, ride_x=dx1, ride_out=ro)
                c = sh.counters()
                outs.append(((sa.iters, sa.converged, sb.iters, sb.converged) + tuple(c[k] for k in keys),
                             [ra, rb] + [q.download() for q in (dv, mu, at, mb, aat, ata, ro, po)]))
            finally:
                for k in env:
                    os.environ.pop(k, None)
                sh.force_multi(0)
        assert outs[0][0][0] >= 2 and outs[0][0][2] >= 2, outs[0][0]
        assert outs[1][0] == outs[0][0] and outs[2][0] == outs[0][0], [o[0] for o in outs]
        for a_, b_ in zip(outs[0][1], outs[1][1]):
            assert np.array_equal(a_, b_, equal_nan=True)
        for a_, b_ in zip(outs[0][1], outs[2][1]):
            assert rel(np.asarray(b_), np.asarray(a_)) < 1e-11
        # and the products themselves: the rider and the completed right-hand side against plain matvecs
        assert rel(outs[0][1][8], sh.Ax(x1)) < 1e-12
        assert rel(outs[0][1][9], sh.Ax(r2)) < 1e-12
