"""Two vectors per pass over the genotype shard: gv_ax2_dev / gv_atx2_dev must reproduce the one-vector products bit
for bit (exact integer accumulation), and gv_cg_solve2 (LMMSE + Onsager solves in lock-step) must reproduce two
stand-alone gv_cg_solve runs."""
import numpy as np
import pytest

from gvamp_amd import capi, synth


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,M,mode", [(2000, 3000, 1), (1003, 517, 1), (2000, 1000, 0)])
def test_two_vector_products_are_bit_identical(N, M, mode):
    rng = np.random.default_rng(N)
    bed = synth.synth_bed(N, M, seed=5, miss_ppm=10000)
    with capi.Shard(N, M, anchor=(mode == 0)) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(mode)
        sh.compute_markers_statistics()
        xa, xb = sh.vecM(rng.standard_normal(M)), sh.vecM(rng.standard_normal(M) * 1e-6)
        za, zb, z1, z2 = sh.vecN(), sh.vecN(), sh.vecN(), sh.vecN()
        sh.counters(reset=True)
        sh.ax2_dev(xa, xb, za, zb)
        c = sh.counters()
        assert c["n_ax"] == 2 and c["n_ax_pass"] == (1 if mode == 1 else 2)
        sh.ax_dev(xa, z1)
        sh.ax_dev(xb, z2)
        assert np.array_equal(za.download(), z1.download()) and np.array_equal(zb.download(), z2.download())
        wa, wb, w1, w2 = sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        sh.atx2_dev(za, zb, wa, wb)
        sh.atx_dev(za, w1)
        sh.atx_dev(zb, w2)
        assert np.array_equal(wa.download(), w1.download()) and np.array_equal(wb.download(), w2.download())


@pytest.mark.parametrize("warm", [False, True])
def test_dual_cg_equals_two_single_solves(oracle, warm):
    N, M = 2000, 1500
    rng = np.random.default_rng(3)
    bed = synth.synth_bed(N, M, seed=11)
    v = rng.standard_normal(M)
    u = oracle.bern_vec(7, 0, M, M)
    mu0 = 0.1 * rng.standard_normal(M) if warm else None
    tau, gam2 = 2.0, 1.35
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        dv, du = sh.vecM(v), sh.vecM(u)
        dm0 = sh.vecM(mu0) if warm else None
        a1, b1, a2, b2 = sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        sa, ra = sh.cg_solve(dv, dm0, tau, gam2, 1, 25, a1)
        sb, rb = sh.cg_solve(du, None, tau, gam2, 0, 25, b1)
        sh.counters(reset=True)
        (s2a, r2a), (s2b, r2b) = sh.cg_solve2(dv, dm0, du, tau, gam2, 25, a2, b2)
        c = sh.counters()
        assert (s2a.iters, s2b.iters) == (sa.iters, sb.iters)
        assert np.array_equal(r2a, ra) and np.array_equal(r2b, rb)
        assert np.array_equal(a2.download(), a1.download()) and np.array_equal(b2.download(), b1.download())
        assert s2b.onsager == sb.onsager
        # vector products are unchanged, passes over the shard are fewer
        nvec = sa.iters + sb.iters + (1 if warm else 0)
        assert c["n_ax"] == nvec and c["n_atx"] == nvec
        assert c["n_ax_pass"] == max(sa.iters + (1 if warm else 0), sb.iters) and c["n_ax_pass"] < nvec
    # and against the oracle
    o_mu, o_rr = oracle.cg_solve(bed, N, M, v, mu0, tau, gam2, 1, 25)
    assert np.allclose(ra, o_rr, rtol=1e-9)


@pytest.mark.parametrize("warm,gam2", [(False, 1.35), (True, 1.35), (True, 250.0), (False, 1e-6)])
def test_cg_by_products_equal_explicit_products(warm, gam2):
    """gv_cg_solve2x: the rider is a plain Ax (bit-identical); A mu_a and A^T A mu_b from the CG recurrences agree with
    the explicit products to rounding (tolerances: 1e-12, and 1e-11 x the cancellation factor of the residual form)."""
    N, M = 2000, 1500
    rng = np.random.default_rng(5)
    bed = synth.synth_bed(N, M, seed=11)
    v, xr = rng.standard_normal(M), rng.standard_normal(M)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    mu0 = 0.1 * rng.standard_normal(M) if warm else None
    tau = 2.0
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        dv, du, dx = sh.vecM(v), sh.vecM(u), sh.vecM(xr)
        dm0 = sh.vecM(mu0) if warm else None
        a1, b1, a2, b2 = sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        zr, za, wb = sh.vecN(), sh.vecN(), sh.vecM()
        (s1a, _), (s1b, _) = sh.cg_solve2(dv, dm0, du, tau, gam2, 25, a1, b1)
        sh.counters(reset=True)
        (s2a, _), (s2b, _) = sh.cg_solve2x(dv, dm0, du, tau, gam2, 25, a2, b2, ride_x=dx, ride_out=zr, a_mu_a=za, ata_mu_b=wb)
        c = sh.counters()
        # the solves themselves are untouched
        assert (s2a.iters, s2b.iters) == (s1a.iters, s1b.iters)
        assert np.array_equal(a2.download(), a1.download()) and np.array_equal(b2.download(), b1.download())
        # the rider took no pass of its own unless both solves ran equally long
        napp_a, napp_b = s1a.iters + (1 if warm else 0), s1b.iters
        assert c["n_ax_pass"] == max(napp_a, napp_b) + (1 if napp_a == napp_b else 0)
        assert np.array_equal(zr.download(), sh.Ax(xr))
        mu_a, mu_b = a2.download(), b2.download()
        assert rel(za.download(), sh.Ax(mu_a)) < 1e-12
        ata = sh.ATx(sh.Ax(mu_b))
        amp = max(1.0, np.linalg.norm(u) / (tau * np.linalg.norm(ata)))     # (v - r - gam2 mu) / tau cancels to A^T A mu
        assert rel(wb.download(), ata) < 1e-11 * amp
        # what VAMP takes from it (vamp.cpp:914): <u, A^T A invQ u>
        t1, t2 = float(u @ wb.download()), float(u @ ata)
        assert abs(t1 - t2) < 1e-11 * amp * abs(t2)


def test_vamp_run_with_by_products_equals_reference_sequence():
    """--fuse-solves 2 (rider + CG by-products) against --fuse-solves 0 (the reference's sequence of products): same CG
    and EM counts, estimates equal to rounding, three passes fewer per iteration."""
    from gvamp_amd import hostapi
    N, M = 3000, 4000
    with capi.Shard(N, M) as sh:
        sh.set_kernel_mode(1)
        sh.synth_bed(99, 5000)
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, 300, 3)
        kw = dict(iterations=5, CG_max_iter=40, rho=0.5, seed=3, true_signal=beta, history=True)
        r0 = hostapi.infere_linear(sh, y, [0.9, 0.07, 0.03], [0, 1e-3, 1e-2], fuse_solves=0, **kw)
        r2 = hostapi.infere_linear(sh, y, [0.9, 0.07, 0.03], [0, 1e-3, 1e-2], fuse_solves=2, **kw)
    assert r0.niter == r2.niter
    for a, b in zip(r0.trace, r2.trace):
        assert (a["cg_iters"], a["onsager_iters"], a["revar_rounds"], a["L_after"]) == \
               (b["cg_iters"], b["onsager_iters"], b["revar_rounds"], b["L_after"])
        for k in ("gam1_denoise", "alpha1", "gam2", "alpha2", "gamw", "R2_denoise", "R2_lmmse", "gam1_next"):
            assert abs(a[k] - b[k]) <= 1e-9 * abs(a[k]), k
        assert b["n_ax_pass"] + b["n_atx_pass"] < a["n_ax_pass"] + a["n_atx_pass"]
    for xa, xb in zip(r0.x1, r2.x1):
        assert rel(xb, xa) < 1e-9
    assert rel(r2.x_est, r0.x_est) < 1e-9


@pytest.mark.parametrize("max_iter", [0, 1])
def test_joint_solvers_with_no_or_one_step(max_iter):
    """CG-max-iter 0 / 1: the joint solvers must stop where the separate ones stop (no step taken, or exactly one)."""
    N, M = 900, 1300
    rng = np.random.default_rng(8)
    bed = synth.synth_bed(N, M, seed=12)
    npad = 4 * ((N + 3) // 4)
    v, xr = rng.standard_normal(M), rng.standard_normal(M)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    mu0 = 0.1 * rng.standard_normal(M)
    vn = np.zeros(npad)
    vn[:N] = rng.standard_normal(N)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        sh.compute_people_statistics()
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        dv, du, dx, dm0, dvn = sh.vecM(v), sh.vecM(u), sh.vecM(xr), sh.vecM(mu0), sh.vecN(vn)
        a1, b1, a2, b2 = sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        zr, za, wb = sh.vecN(), sh.vecN(), sh.vecM()
        sa, _ = sh.cg_solve(dv, dm0, 2.0, 1.1, 1, max_iter, a1)
        sb, _ = sh.cg_solve(du, None, 2.0, 1.1, 0, max_iter, b1)
        (s2a, _), (s2b, _) = sh.cg_solve2x(dv, dm0, du, 2.0, 1.1, max_iter, a2, b2, ride_x=dx, ride_out=zr, a_mu_a=za,
                                             ata_mu_b=wb)
        assert (s2a.iters, s2b.iters) == (sa.iters, sb.iters) == (max_iter, max_iter)
        assert np.array_equal(a2.download(), a1.download()) and np.array_equal(b2.download(), b1.download())
        assert np.array_equal(zr.download(), sh.Ax(xr))
        assert rel(za.download(), sh.Ax(a2.download())) < 1e-12
        if max_iter:
            assert rel(wb.download(), sh.ATx(sh.Ax(b2.download()))) < 1e-10
        else:
            assert np.all(wb.download() == 0) and np.all(b2.download() == 0)
        # N-space / M-space pair
        n1, n2, m1, m2, at1, at2 = sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        sna, _ = sh.cg_solve_aat(dvn, None, 2.0, 1.1, max_iter, n1)
        sh.atx_dev(n1, at1)
        (t2a, _), (t2b, _) = sh.cg_solve_aat2(dvn, None, du, 2.0, 1.1, max_iter, n2, at2, m2)
        assert (t2a.iters, t2b.iters) == (sna.iters, sb.iters)
        assert np.array_equal(n2.download(), n1.download()) and np.array_equal(at2.download(), at1.download())
        assert np.array_equal(m2.download(), b1.download())
