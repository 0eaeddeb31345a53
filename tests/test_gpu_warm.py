"""--fuse-solves 3: the warm start of the LMMSE solve without its opening operator application.

precondCG_solver starts with r = v - Q mu_start (vamp.cpp:1142-1145: one Ax + one ATx).  mu_start is the solution of the
previous iteration's solve, whose final residual already holds the product: Q' mu = v' - r'.  gv_cg_solve2w /
gv_cg_solve_aat2w take that product from the caller; these tests hold them against the explicit warm start (same step counts,
iterates to rounding, two passes fewer) and whole VAMP runs against --fuse-solves 2 and the oracle."""
import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


@pytest.mark.parametrize("layout", [1, 2])
@pytest.mark.parametrize("N,M", [(2000, 1500), (1003, 2049)])
def test_chained_solves_with_known_products_match_the_explicit_warm_start(N, M, layout):
    """three solves in a row, each warm-started from the one before with (tau, gam2, v) changing as in a VAMP run: the
    chain that passes A^T A mu / A mu along must follow the chain that applies the operator"""
    rng = np.random.default_rng(N + layout)
    bed = synth.synth_bed(N, M, seed=21, miss_ppm=8000)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    with capi.Shard(N, M) as sh:
        sh.set_layout(True, layout)
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        du = sh.vecM(u)
        mu_e, mu_w, mb_e, mb_w = sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()      # explicit / known-product chains
        start_e, start_w = sh.vecM(), sh.vecM()
        az_e, az_w, ata_w = sh.vecN(), sh.vecN(), sh.vecM()
        for k, (tau, gam2) in enumerate([(2.0, 1.35), (1.7, 2.9), (2.4, 0.6)]):
            dv = sh.vecM(rng.standard_normal(M))
            warm = k > 0
            sh.counters(reset=True)
            (se, re_), (sbe, _) = sh.cg_solve2x(dv, start_e if warm else None, du, tau, gam2, 30, mu_e, mb_e, a_mu_a=az_e)
            ce = sh.counters()
            sh.counters(reset=True)
            (sw, rw), (sbw, _) = sh.cg_solve2x(dv, start_w if warm else None, du, tau, gam2, 30, mu_w, mb_w, a_mu_a=az_w,
                                               ata_mu_start_a=ata_w if warm else None, a_mu_start_a=az_w if warm else None,
                                               ata_mu_a=ata_w)
            cw = sh.counters()
            assert (sw.iters, sbw.iters, sw.converged) == (se.iters, sbe.iters, se.converged)
            assert np.allclose(rw, re_, rtol=1e-7)                      # residual traces (values ~1e-5 at the exit)
            assert rel(mu_w.download(), mu_e.download()) < 1e-11
            assert rel(mb_w.download(), mb_e.download()) < 1e-11
            assert rel(az_w.download(), sh.Ax(mu_w.download())) < 1e-11   # A mu keeps accumulating in place across the calls
            assert rel(ata_w.download(), sh.ATx(sh.Ax(mu_w.download()))) < 1e-9   # (v - r - gam2 mu) / tau: r is 1e-5 |v|
            # the warm start's operator application is gone: one Ax pass and one ATx pass fewer
            assert cw["n_ax_pass"] == ce["n_ax_pass"] - (1 if warm else 0)
            assert cw["n_atx_pass"] == ce["n_atx_pass"] - (1 if warm else 0)
            start_e.upload(mu_e.download())
            start_w.upload(mu_w.download())
            dv.free()


def test_known_product_arguments_are_checked():
    N, M = 400, 300
    bed = synth.synth_bed(N, M, seed=2)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        v, u, mu, mb, ata, az = sh.vecM(np.ones(M)), sh.vecM(np.ones(M) / np.sqrt(M)), sh.vecM(), sh.vecM(), sh.vecM(), sh.vecN()
        with pytest.raises(capi.GvError):        # a known product without the start it belongs to
            sh.cg_solve2x(v, None, u, 2.0, 1.0, 5, mu, mb, ata_mu_start_a=ata)
        start = sh.vecM(np.zeros(M))
        with pytest.raises(capi.GvError):        # a_mu_a wanted, A mu_start not given
            sh.cg_solve2x(v, start, u, 2.0, 1.0, 5, mu, mb, a_mu_a=az, ata_mu_start_a=ata)
        with pytest.raises(capi.GvError):        # output aliasing the solution
            sh.cg_solve2x(v, start, u, 2.0, 1.0, 5, mu, mb, ata_mu_a=mu)
        # and the context still works
        sh.cg_solve2x(v, start, u, 2.0, 1.0, 5, mu, mb, ata_mu_start_a=ata, ata_mu_a=ata)
        assert np.all(np.isfinite(mu.download()))


def test_xxt_joint_solver_with_a_known_start_product():
    N, M = 1200, 900
    rng = np.random.default_rng(5)
    bed = synth.synth_bed(N, M, seed=31, miss_ppm=5000)
    npad = 4 * ((N + 3) // 4)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        sh.compute_people_statistics()
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        du = sh.vecM(u)
        n_e, n_w, at_e, at_w, m_e, m_w = sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        s_e, s_w, aat_w = sh.vecN(), sh.vecN(), sh.vecN()
        n_c, at_c, m_c, s_c, aat_c = sh.vecN(), sh.vecM(), sh.vecM(), sh.vecN(), sh.vecN()
        for k, (tau, gam2) in enumerate([(2.0, 1.1), (1.4, 2.2), (2.6, 0.8)]):
            vn = np.zeros(npad)
            vn[:N] = rng.standard_normal(N)
            dvn = sh.vecN(vn)
            warm = k > 0
            sh.counters(reset=True)
            (ae, _), (be, _) = sh.cg_solve_aat2(dvn, s_e if warm else None, du, tau, gam2, 30, n_e, at_e, m_e)
            ce = sh.counters()
            sh.counters(reset=True)
            (aw, _), (bw, _) = sh.cg_solve_aat2(dvn, s_w if warm else None, du, tau, gam2, 30, n_w, at_w, m_w, aat_mu_a=aat_w,
                                                aat_mu_start_a=aat_w if warm else None)
            cw = sh.counters()
            # third chain: A^T mu_a accumulated inside the solve as well (no closing ATx pass), from the previous call's value
            sh.counters(reset=True)
            (ac, _), (bc, _) = sh.cg_solve_aat2(dvn, s_c if warm else None, du, tau, gam2, 30, n_c, at_c, m_c, aat_mu_a=aat_c,
                                                aat_mu_start_a=aat_c if warm else None, at_mu_start_a=at_c if warm else None,
                                                accumulate_at_mu_a=True)
            cc = sh.counters()
            assert (ac.iters, bc.iters) == (ae.iters, be.iters)
            assert rel(n_c.download(), n_e.download()) < 1e-10 and rel(m_c.download(), m_e.download()) < 1e-10
            assert rel(at_c.download(), at_e.download()) < 1e-10
            assert cc["n_atx"] == cw["n_atx"] - 1                         # the closing A^T mu_a is gone
            assert cc["n_ax_pass"] + cc["n_atx_pass"] <= cw["n_ax_pass"] + cw["n_atx_pass"]
            s_c.upload(n_c.download())
            # and accumulation with an explicit warm start (its opening application delivers A^T mu0)
            n_x, at_x, m_x = sh.vecN(), sh.vecM(), sh.vecM()
            sh.cg_solve_aat2(dvn, s_e if warm else None, du, tau, gam2, 30, n_x, at_x, m_x, accumulate_at_mu_a=True)
            assert rel(at_x.download(), at_e.download()) < 1e-10 and rel(n_x.download(), n_e.download()) < 1e-12
            for q in (n_x, at_x, m_x):
                q.free()
            assert (aw.iters, bw.iters) == (ae.iters, be.iters)
            assert rel(n_w.download(), n_e.download()) < 1e-10
            assert rel(at_w.download(), at_e.download()) < 1e-10
            assert rel(m_w.download(), m_e.download()) < 1e-10
            # (the passes are shared with the M-space solve, half an application out of phase: the two half-applications saved
            # show as fewer passes only when solve a is the longer chain)
            assert cw["n_ax_pass"] + cw["n_atx_pass"] <= ce["n_ax_pass"] + ce["n_atx_pass"]
            assert cw["n_ax"] + cw["n_atx"] == ce["n_ax"] + ce["n_atx"] - (2 if warm else 0)      # vector products: two fewer
            s_e.upload(n_e.download())
            s_w.upload(n_w.download())
            dvn.free()


@pytest.mark.parametrize("level", [3, 4])
@pytest.mark.parametrize("xxt", [0, 1])
def test_vamp_runs_at_fuse_3_and_4_follow_fuse_2_and_the_oracle(oracle, xxt, level, monkeypatch):
    monkeypatch.setenv("GV_LINEARITY_MAX", "1e300")     # (product counts of the linearity path; its cancellation guard is tested below)
    N, M = 2000, 3000
    bed = synth.synth_bed(N, M, seed=44, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 150, 4)
    probs, vars_ = [0.9, 0.07, 0.03], [0, 1e-3, 1e-2]
    kw = dict(iterations=6, CG_max_iter=40, rho=0.5, seed=4, true_signal=beta, history=True)
    if xxt:
        kw["use_XXT_denoiser"] = 1
    ref = oracle.infere(bed, N, M, y, probs, vars_, **{k: v for k, v in kw.items() if k != "history"})
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        r2 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=2, **kw)
        r3 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=level, **kw)
    assert r3.niter == r2.niter == ref.niter
    for i, (a, b, o) in enumerate(zip(r2.trace, r3.trace, ref.trace)):
        assert (b["cg_iters"], b["onsager_iters"], b["L_after"]) == (a["cg_iters"], a["onsager_iters"], a["L_after"])
        assert b["cg_iters"] == int(o["cg_iters"])
        for k in ("gam1_denoise", "alpha1", "gam2", "alpha2", "gamw", "gam1_next"):
            assert abs(a[k] - b[k]) <= 1e-9 * abs(a[k]), (i, k)
        # the warm start's Ax + ATx are gone as vector products; as PASSES the saving is 0, 1 or 2 -- the opening application of
        # solve a shared its pass pair with the first step of solve b, so a pair disappears only when a was the longer chain,
        # and z1 = A x1_hat, which rode in the slot a solve that finished early leaves free, takes a pass of its own when
        # both solves now finish together
        passes2, passes3 = a["n_ax_pass"] + a["n_atx_pass"], b["n_ax_pass"] + b["n_atx_pass"]
        # (XXT: the closing A^T u of denoiserXXT.cpp:46 is accumulated inside the solve as well -- one more product, every iteration;
        # and from the second iteration on the Onsager solve takes its first application from A^T A u of the probe: two more)
        # (XXT at level 4: A r2 = c1 A x1_hat - c2 A r1 by linearity -- the first pass of the joint solve carries z1 = A x1_hat and the
        # product A r2 is never formed: one more, every iteration)
        fewer = (2 if i > 0 else 0) + (1 if xxt else 0) + (2 if level >= 4 and i > 0 else 0) + (1 if xxt and level >= 4 else 0)
        assert passes2 - fewer <= passes3 <= passes2, (i, passes2, passes3)
        # (level 4 keeps A^T A u from the first iteration whose gam2 / tau lets it be captured without cancellation: the first or
        # the second -- so the second iteration may still apply the operator for the Onsager solve's first step)
        ok = {fewer, fewer - 2} if (level >= 4 and i == 1) else {fewer}
        assert a["n_ax"] + a["n_atx"] - (b["n_ax"] + b["n_atx"]) in ok, (i, fewer)
    saved = sum(a["n_ax_pass"] + a["n_atx_pass"] - b["n_ax_pass"] - b["n_atx_pass"] for a, b in zip(r2.trace, r3.trace))
    assert saved >= (1 if (level >= 4 or not xxt) else 0), saved
    assert rel(r3.x_est, r2.x_est) < 1e-9
    assert rel(r3.x_est, ref.x_est) < 1e-7


@pytest.mark.parametrize("layout", [1, 2])
def test_zero_started_solve_with_the_product_of_its_right_hand_side(layout):
    """solve b (the Onsager probe solve: zero start, the same v_b call after call): A^T A v_b captured from its first application,
    then handed back -- same steps, same iterates to rounding, one Ax and one ATx fewer"""
    N, M = 1500, 2300
    rng = np.random.default_rng(17)
    bed = synth.synth_bed(N, M, seed=9, miss_ppm=5000)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    with capi.Shard(N, M) as sh:
        sh.set_layout(True, layout)
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        du, atau = sh.vecM(u), sh.vecM()
        mu_a, mu_b, mu_a2, mu_b2, wb, wb2 = sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        for k, (tau, gam2) in enumerate([(2.0, 1.35), (1.1, 3.0), (3.0, 0.4)]):
            dv = sh.vecM(rng.standard_normal(M))
            sh.counters(reset=True)
            (sa, _), (sb, rb) = sh.cg_solve2x(dv, None, du, tau, gam2, 30, mu_a, mu_b, ata_mu_b=wb)
            c0 = sh.counters()
            sh.counters(reset=True)
            (sa2, _), (sb2, rb2) = sh.cg_solve2x(dv, None, du, tau, gam2, 30, mu_a2, mu_b2, ata_mu_b=wb2, ata_v_b=atau,
                                                 have_ata_v_b=k > 0)
            c1 = sh.counters()
            assert (sa2.iters, sb2.iters, sb2.converged) == (sa.iters, sb.iters, sb.converged)
            assert len(rb2) == len(rb) and np.allclose(rb2, rb, rtol=1e-8)
            assert abs(sb2.onsager - sb.onsager) <= 1e-13 * abs(sb.onsager)
            assert rel(mu_b2.download(), mu_b.download()) < 1e-12 and rel(wb2.download(), wb.download()) < 1e-10
            if k == 0:      # capture only: nothing may change, bit for bit
                assert np.array_equal(mu_b2.download(), mu_b.download()) and np.array_equal(mu_a2.download(), mu_a.download())
                assert rel(atau.download(), sh.ATx(sh.Ax(u))) < 1e-12
                assert (c1["n_ax"], c1["n_atx"]) == (c0["n_ax"], c0["n_atx"])
            else:
                assert (c1["n_ax"], c1["n_atx"]) == (c0["n_ax"] - 1, c0["n_atx"] - 1)
                assert c1["n_ax_pass"] + c1["n_atx_pass"] <= c0["n_ax_pass"] + c0["n_atx_pass"]
            dv.free()
        # the same through the XXT joint solver
        sh.compute_people_statistics()
        npad = 4 * ((N + 3) // 4)
        vn = np.zeros(npad)
        vn[:N] = rng.standard_normal(N)
        dvn, n1, n2, at1, at2, m1, m2 = sh.vecN(vn), sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM(), sh.vecM(), sh.vecM()
        (a1, _), (b1, r1) = sh.cg_solve_aat2(dvn, None, du, 2.0, 1.1, 30, n1, at1, m1)
        sh.counters(reset=True)
        (a2, _), (b2, r2) = sh.cg_solve_aat2(dvn, None, du, 2.0, 1.1, 30, n2, at2, m2, ata_v_b=atau, have_ata_v_b=True)
        assert (a2.iters, b2.iters) == (a1.iters, b1.iters) and len(r2) == len(r1) and np.allclose(r2, r1, rtol=1e-8)
        assert rel(m2.download(), m1.download()) < 1e-12 and rel(n2.download(), n1.download()) < 1e-12
        fresh = sh.vecM()
        (a3, _), (b3, _) = sh.cg_solve_aat2(dvn, None, du, 2.0, 1.1, 30, n2, at2, m2, ata_v_b=fresh)          # capture there too
        assert rel(fresh.download(), atau.download()) < 1e-12


@pytest.mark.parametrize("layout", [1, 2])
def test_xxt_joint_solver_completes_its_right_hand_side_and_carries_a_rider(layout):
    """gv_aat_warm.pre_x / ride_x: v_a = y - A r2 formed inside the call (the A r2 in the pass of solve b's first half-application)
    and z1 = A x1_hat in a free slot -- every output bit-identical to forming them outside, in fewer passes"""
    N, M = 1400, 1100
    rng = np.random.default_rng(23)
    bed = synth.synth_bed(N, M, seed=13, miss_ppm=5000)
    npad = 4 * ((N + 3) // 4)
    u = np.where(rng.random(M) < 0.5, -1.0, 1.0) / np.sqrt(M)
    yv = np.zeros(npad)
    yv[:N] = rng.standard_normal(N)
    r2, x1 = rng.standard_normal(M) * 0.05, rng.standard_normal(M) * 0.05
    with capi.Shard(N, M) as sh:
        sh.set_layout(True, layout)
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        sh.compute_people_statistics()
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        du, dr2, dx1, dy = sh.vecM(u), sh.vecM(r2), sh.vecM(x1), sh.vecN(yv)
        # outside: one two-vector Ax for z1 and A r2, then the joint solve
        z1_e, ar2_e, v_e = sh.vecN(), sh.vecN(), sh.vecN()
        n_e, at_e, m_e = sh.vecN(), sh.vecM(), sh.vecM()
        sh.counters(reset=True)
        sh.ax2_dev(dx1, dr2, z1_e, ar2_e)
        v_e.upload(yv - ar2_e.download())
        (ae, rae), (be, rbe) = sh.cg_solve_aat2(v_e, None, du, 2.0, 1.1, 30, n_e, at_e, m_e)
        ce = sh.counters()
        # inside
        z1_w, ar2_w, v_w = sh.vecN(), sh.vecN(), sh.vecN(yv)
        n_w, at_w, m_w = sh.vecN(), sh.vecM(), sh.vecM()
        sh.counters(reset=True)
        (aw, raw), (bw, rbw) = sh.cg_solve_aat2(v_w, None, du, 2.0, 1.1, 30, n_w, at_w, m_w, pre_x=dr2, pre_out=ar2_w, ride_x=dx1,
                                                ride_out=z1_w)
        cw = sh.counters()
        assert (aw.iters, bw.iters) == (ae.iters, be.iters)
        assert np.array_equal(raw, rae) and np.array_equal(rbw, rbe)
        for a, b in ((z1_w, z1_e), (ar2_w, ar2_e), (n_w, n_e), (at_w, at_e), (m_w, m_e)):
            assert np.array_equal(a.download(), b.download())
        assert np.array_equal(v_w.download(), yv - ar2_e.download())          # v_a was completed in place
        assert (cw["n_ax"], cw["n_atx"]) == (ce["n_ax"], ce["n_atx"])            # the same products ...
        assert cw["n_ax_pass"] + cw["n_atx_pass"] <= ce["n_ax_pass"] + ce["n_atx_pass"]     # ... in no more passes
        # with a solve b that does not run at all the two extra products still come out (they share one pass)
        z1_0, ar2_0, v_0 = sh.vecN(), sh.vecN(), sh.vecN(yv)
        sh.cg_solve_aat2(v_0, None, du, 2.0, 1.1, 0, n_w, at_w, m_w, pre_x=dr2, pre_out=ar2_0, ride_x=dx1, ride_out=z1_0)
        assert np.array_equal(z1_0.download(), z1_e.download()) and np.array_equal(ar2_0.download(), ar2_e.download())
        with pytest.raises(capi.GvError):
            sh.cg_solve_aat2(v_0, None, du, 2.0, 1.1, 5, n_w, at_w, m_w, pre_x=dr2)          # pre_x without pre_out


@pytest.mark.parametrize("xxt", [0, 1])
@pytest.mark.parametrize("variant", ["fp64 kernels", "host-driven CG"])
def test_level_4_on_the_other_code_paths(oracle, monkeypatch, variant, xxt):
    """the identities of levels 3 / 4 do not depend on the fixed-point family or on the device-resident loop: the same runs through
    the fp64 VALU kernels (kernel mode 0: host-driven CG, raw rows) and through GV_CG_DEVICE=0 must follow level 0 and the oracle"""
    N, M = 1200, 1700
    bed = synth.synth_bed(N, M, seed=52, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 90, 6)
    probs, vars_ = [0.9, 0.07, 0.03], [0, 1e-3, 1e-2]
    kw = dict(iterations=5, CG_max_iter=40, rho=0.5, seed=6, true_signal=beta, history=False)
    if xxt:
        kw["use_XXT_denoiser"] = 1
    ref = oracle.infere(bed, N, M, y, probs, vars_, **{k: v for k, v in kw.items() if k != "history"})
    if variant == "host-driven CG":
        monkeypatch.setenv("GV_CG_DEVICE", "0")
    with capi.Shard(N, M, anchor=(variant == "fp64 kernels")) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(0 if variant == "fp64 kernels" else 1)
        sh.compute_markers_statistics()
        r0 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=0, **kw)
        r4 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=4, **kw)
    assert r4.niter == r0.niter == ref.niter
    for a, b, o in zip(r0.trace, r4.trace, ref.trace):
        assert (b["cg_iters"], b["onsager_iters"], b["L_after"]) == (a["cg_iters"], a["onsager_iters"], a["L_after"])
        assert b["cg_iters"] == int(o["cg_iters"])
    assert sum(t["n_ax"] + t["n_atx"] for t in r4.trace) < sum(t["n_ax"] + t["n_atx"] for t in r0.trace)
    assert rel(r4.x_est, r0.x_est) < 1e-8 and rel(r4.x_est, ref.x_est) < 1e-7


@pytest.mark.parametrize("reanchor", [0, 2, 3])
def test_xxt_level_4_takes_A_r2_by_linearity_and_reanchors(oracle, reanchor, monkeypatch):
    """--use-XXT-denoiser 1 at level 4: r2 = (eta1 x1_hat - gam1 r1) / gam2 and r1 = (eta2 x2_hat - gam2 r2) / gam1 are linear in
    vectors whose products are at hand, so the joint solve's first pass carries z1 = A x1_hat and A r2 is never multiplied out
    (host/vamp.cpp: ar1 / ar2).  Follows level 0 and the oracle; one product fewer than the explicit form in every iteration but the
    re-anchored ones (--reanchor-every K: explicit A r2 and z1 there, so that rounding never chains over more than K iterations)."""
    N, M = 1500, 2200
    bed = synth.synth_bed(N, M, seed=61, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 110, 9)
    probs, vars_ = [0.9, 0.07, 0.03], [0, 1e-3, 1e-2]
    kw = dict(iterations=7, CG_max_iter=40, rho=0.5, seed=9, true_signal=beta, history=True, use_XXT_denoiser=1)
    ref = oracle.infere(bed, N, M, y, probs, vars_, **{k: v for k, v in kw.items() if k != "history"})
    # (the cancellation guard of host/vamp.cpp, linearity_max, lifted: this test is about the product counts of the linearity
    # path on every iteration that is not re-anchored; the guard has its own test below)
    monkeypatch.setenv("GV_LINEARITY_MAX", "1e300")
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        r0 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=0, **kw)
        r3 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=3, reanchor_every=reanchor, **kw)
        r4 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=4, reanchor_every=reanchor, **kw)
        r4b = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=4, reanchor_every=reanchor, **kw)      # a second run on the shard
    assert r4.niter == r0.niter == ref.niter == 7
    for i, (a, b, o) in enumerate(zip(r0.trace, r4.trace, ref.trace)):
        assert (b["cg_iters"], b["onsager_iters"], b["L_after"]) == (a["cg_iters"], a["onsager_iters"], a["L_after"]), i
        assert b["cg_iters"] == int(o["cg_iters"])
        for k in ("gam1_denoise", "alpha1", "gam2", "alpha2", "gamw", "gam1_next", "R2_denoise", "R2_lmmse"):
            assert abs(a[k] - b[k]) <= 1e-9 * abs(a[k]), (i, k, a[k], b[k])
    assert rel(r4.x_est, r0.x_est) < 1e-9 and rel(r4.x_est, ref.x_est) < 1e-7
    for i in range(7):
        assert rel(r4.x1[i], r0.x1[i]) < 1e-9 and rel(r4.x2[i], r0.x2[i]) < 1e-9
    assert np.array_equal(r4b.x_est, r4.x_est)
    # products: level 4 against level 3 on the same schedule of re-anchored iterations (iteration numbers it = i + 1 > 1 with
    # it % reanchor == 0) -- A r2 is gone wherever the iteration is not re-anchored; the probe's A^T A u (level 4) saves two more
    for i, (a, b) in enumerate(zip(r3.trace, r4.trace)):
        anchored = reanchor > 0 and i > 0 and (i + 1) % reanchor == 0
        d = a["n_ax"] + a["n_atx"] - (b["n_ax"] + b["n_atx"])
        assert d in ({0, 2} if anchored else {1, 3}), (i, anchored, d)
        assert b["n_ax_pass"] + b["n_atx_pass"] <= a["n_ax_pass"] + a["n_atx_pass"], i


def test_xxt_level_4_linearity_guard_falls_back_to_the_explicit_product(oracle, monkeypatch):
    """A r2 = c1 z1 - c2 A r1 cancels (c1 - c2 = 1, c1 + c2 = (eta1 + gam1) / (eta1 - gam1)): iterations whose |c1| + |c2| exceeds
    linearity_max (and the one after an iteration whose A r1 = (eta2 A x2_hat - gam2 A r2) / gam1 did) take A r2 explicitly, exactly as
    a re-anchored iteration does.  With the guard at 0 every iteration falls back and
    level 4 issues level 3's products plus the probe's known A^T A u; with the guard lifted it saves one more product per iteration;
    the default sits between the two and every setting follows level 0."""
    N, M = 1500, 2200
    bed = synth.synth_bed(N, M, seed=61, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, M, 0.5, 110, 9)
    probs, vars_ = [0.9, 0.07, 0.03], [0, 1e-3, 1e-2]
    kw = dict(iterations=7, CG_max_iter=40, rho=0.5, seed=9, true_signal=beta, history=False, use_XXT_denoiser=1, reanchor_every=0)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        r0 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=0, **kw)
        runs = {}
        for name, val in (("never", "0"), ("default", None), ("always", "1e300")):
            if val is None:
                monkeypatch.delenv("GV_LINEARITY_MAX", raising=False)
            else:
                monkeypatch.setenv("GV_LINEARITY_MAX", val)
            runs[name] = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=4, **kw)
    prods = {k: [t["n_ax"] + t["n_atx"] for t in r.trace] for k, r in runs.items()}
    amp = [(2 * t["eta1"] - t["gam2"]) / t["gam2"] for t in r0.trace]          # c1 + c2 of each iteration
    amp_r1 = [(t["eta2"] + t["gam2_reest"]) / t["gam1_next"] for t in r0.trace]   # of the A r1 the iteration leaves for the next
    for i in range(1, 7):           # (iteration 1 has no A r1 yet: explicit whatever the guard says)
        assert prods["never"][i] - prods["always"][i] == 1, (i, prods)
        want = prods["always"][i] if max(amp[i], amp_r1[i - 1]) <= 100.0 else prods["never"][i]
        assert prods["default"][i] == want, (i, amp[i], amp_r1[i - 1], prods)
    for r in runs.values():
        assert rel(r.x_est, r0.x_est) < 1e-9
        for a, b in zip(r.trace, r0.trace):
            assert (a["cg_iters"], a["onsager_iters"], a["L_after"]) == (b["cg_iters"], b["onsager_iters"], b["L_after"])
