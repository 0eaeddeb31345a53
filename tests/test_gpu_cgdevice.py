"""The device-resident CG loop (gv_solvers.hip: cg_run_device, default in kernel mode 1) against the host-driven loop it
replaces (GV_CG_DEVICE=0) and against the oracle: per system the same iterates, traces, iteration counts and product
counts -- the scalars only moved from the host to the device (vamp.cpp:1160-1223)."""
import os

import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


class host_loop:
    """GV_CG_DEVICE=0 for the duration of the block (read by libgvamp at every solve)"""
    def __enter__(self):
        os.environ["GV_CG_DEVICE"] = "0"

    def __exit__(self, *a):
        os.environ.pop("GV_CG_DEVICE", None)


def _shard(N, M, seed=3, fna=0.0):
    bed = synth.synth_bed(N, M, seed=seed, miss_ppm=10000)
    sh = capi.Shard(N, M)
    sh.upload_bed(bed)
    if fna > 0 or N % 4:
        rng = np.random.default_rng(N)
        present = rng.random(N) >= fna
        m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
        for n in np.nonzero(present)[0]:
            m4[n >> 2] |= 1 << (n & 3)
        sh.set_mask(m4, int(present.sum()))
    sh.set_kernel_mode(1)
    sh.compute_markers_statistics()
    return sh, bed


@pytest.mark.parametrize("N,M,warm,denoiser,max_iter", [(2000, 3000, False, 1, 40), (2000, 3000, True, 1, 40), (1003, 700, False, 0, 40),
                                                        (1500, 5000, True, 1, 3), (800, 400, False, 1, 0)])
def test_single_solve_device_loop_equals_host_loop(oracle, N, M, warm, denoiser, max_iter):
    sh, bed = _shard(N, M, fna=0.01 if N % 4 else 0.0)
    with sh:
        rng = np.random.default_rng(M)
        v = rng.standard_normal(M) * (np.sign(rng.standard_normal(M)) / np.sqrt(M) if denoiser == 0 else 1.0)
        mu0 = sh.vecM(rng.standard_normal(M) * 0.1) if warm else None
        dv = sh.vecM(v)
        tau, gam2 = 2.0, 0.7
        mu_d, mu_h = sh.vecM(), sh.vecM()
        c0 = sh.counters(reset=True)
        st_d, rr_d = sh.cg_solve(dv, mu0, tau, gam2, denoiser, max_iter, mu_d)
        c_d = sh.counters(reset=True)
        with host_loop():
            st_h, rr_h = sh.cg_solve(dv, mu0, tau, gam2, denoiser, max_iter, mu_h)
        c_h = sh.counters(reset=True)
        assert (st_d.iters, st_d.converged, st_d.n_relres, st_d.n_ax, st_d.n_atx) == \
               (st_h.iters, st_h.converged, st_h.n_relres, st_h.n_ax, st_h.n_atx)
        for k in ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass"):
            assert c_d[k] == c_h[k], k
        assert len(rr_d) == len(rr_h) and np.allclose(rr_d, rr_h, rtol=1e-10, atol=0)
        assert np.isclose(st_d.rel_res, st_h.rel_res, rtol=1e-10) and np.isclose(st_d.onsager, st_h.onsager, rtol=1e-12)
        assert rel(mu_d.download(), mu_h.download()) < 1e-13
        if not warm and N % 4 == 0 and max_iter > 0:
            o_mu, o_rr = oracle.cg_solve(bed, N, M, v, None, tau, gam2, denoiser, max_iter)
            assert len(o_rr) == len(rr_d) and np.allclose(rr_d, o_rr, rtol=1e-9) and rel(mu_d.download(), o_mu) < 1e-11


@pytest.mark.parametrize("N,M,warm,ride,max_iter", [(2000, 3000, True, True, 40), (1200, 6000, False, True, 40), (2000, 3000, True, False, 2)])
def test_dual_solve_with_by_products_device_loop_equals_host_loop(N, M, warm, ride, max_iter):
    """gv_cg_solve2x: LMMSE + Onsager solves in lock-step, the rider and the recurrence by-products (docs/history/rounds1-3.md section 5)."""
    sh, _ = _shard(N, M, seed=11)
    with sh:
        rng = np.random.default_rng(N + M)
        va = sh.vecM(rng.standard_normal(M))
        vb = sh.vecM(np.sign(rng.standard_normal(M)) / np.sqrt(M))
        mu0 = sh.vecM(rng.standard_normal(M) * 0.05) if warm else None
        rx = sh.vecM(rng.standard_normal(M)) if ride else None
        tau, gam2 = 1.3, 0.9

        def run():
            mu_a, mu_b, ro, amu, ata = sh.vecM(), sh.vecM(), sh.vecN(), sh.vecN(), sh.vecM()
            sh.counters(reset=True)
            (sa, ra), (sb, rb) = sh.cg_solve2x(va, mu0, vb, tau, gam2, max_iter, mu_a, mu_b, ride_x=rx, ride_out=ro if ride else None,
                                               a_mu_a=amu, ata_mu_b=ata)
            return dict(sa=(sa.iters, sa.converged, sa.n_relres), sb=(sb.iters, sb.converged, sb.n_relres), ra=ra, rb=rb,
                        ons=sb.onsager, mu_a=mu_a.download(), mu_b=mu_b.download(), ro=ro.download(), amu=amu.download(),
                        ata=ata.download(), cnt=sh.counters())

        d = run()
        with host_loop():
            h = run()
        assert d["sa"] == h["sa"] and d["sb"] == h["sb"]
        for k in ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass"):
            assert d["cnt"][k] == h["cnt"][k], (k, d["cnt"][k], h["cnt"][k])
        assert np.allclose(d["ra"], h["ra"], rtol=1e-10) and np.allclose(d["rb"], h["rb"], rtol=1e-10)
        assert np.isclose(d["ons"], h["ons"], rtol=1e-12)
        for k in ("mu_a", "mu_b", "amu", "ata"):
            assert rel(d[k], h[k]) < 1e-12, k
        if ride:
            assert np.array_equal(d["ro"], h["ro"])              # the rider's product is exact integer arithmetic either way
        # and the device loop is deterministic
        d2 = run()
        assert np.array_equal(d["mu_a"], d2["mu_a"]) and np.array_equal(d["ra"], d2["ra"]) and np.array_equal(d["rb"], d2["rb"])


@pytest.mark.parametrize("max_iter", [1, 2])
def test_opening_on_the_device_that_ends_a_system_before_the_loop(max_iter):
    """The Onsager solve's first step from the known product A^T A u runs inside the device-side opening (cg_open_device); with
    CG-max-iter 1 that step ENDS the system before the host has read any status.  The loop must take the go word from the state
    blocks, not from what the host believes: same iterates, step and pass counts as the host-driven loop, which knows the outcome."""
    N, M = 1600, 2100
    sh, _ = _shard(N, M, seed=23)
    with sh:
        rng = np.random.default_rng(5)
        va = sh.vecM(rng.standard_normal(M))
        vb = sh.vecM(np.sign(rng.standard_normal(M)) / np.sqrt(M))
        atau = sh.vecM(sh.ATx(sh.Ax(vb.download())))
        tau, gam2 = 1.7, 0.8

        def run():
            mu_a, mu_b, ata = sh.vecM(), sh.vecM(), sh.vecM()
            sh.counters(reset=True)
            (sa, ra), (sb, rb) = sh.cg_solve2x(va, None, vb, tau, gam2, max_iter, mu_a, mu_b, ata_mu_b=ata, ata_v_b=atau, have_ata_v_b=True)
            return dict(sa=(sa.iters, sa.converged, sa.n_relres), sb=(sb.iters, sb.converged, sb.n_relres), ra=ra, rb=rb,
                        ons=sb.onsager, mu_a=mu_a.download(), mu_b=mu_b.download(), cnt=sh.counters())

        d = run()
        with host_loop():
            h = run()
        assert d["sa"] == h["sa"] and d["sb"] == h["sb"] and d["sb"][0] == max_iter
        for k in ("n_ax", "n_atx", "n_ax_pass", "n_atx_pass"):
            assert d["cnt"][k] == h["cnt"][k], (k, d["cnt"][k], h["cnt"][k])
        assert d["cnt"]["n_ax_pass"] == max_iter            # system a's steps; system b's first one cost no pass
        assert np.allclose(d["ra"], h["ra"], rtol=1e-10) and np.allclose(d["rb"], h["rb"], rtol=1e-10)
        assert np.isclose(d["ons"], h["ons"], rtol=1e-12)
        assert rel(d["mu_a"], h["mu_a"]) < 1e-12 and rel(d["mu_b"], h["mu_b"]) < 1e-12


def test_full_vamp_run_device_loop_vs_host_loop_and_oracle(oracle):
    N, M = 2000, 6000
    bed = synth.synth_bed(N, M, seed=2024, miss_ppm=5000)
    probs, vars_ = [0.90, 0.07, 0.03], [0, 0.001, 0.01]
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        beta, y = hostapi.sim_phen(sh, 0.5, 300, 7)
        kw = dict(iterations=5, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8, gamw=2.0, true_signal=beta)
        runs = {}
        for fuse in (0, 1, 2):
            runs[("d", fuse)] = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=fuse, **kw)
            with host_loop():
                runs[("h", fuse)] = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=fuse, **kw)
    ref = oracle.infere(bed, N, M, y, probs, vars_, **kw)
    for fuse in (0, 1, 2):
        d, h = runs[("d", fuse)], runs[("h", fuse)]
        for it in range(5):
            td, th, to = d.trace[it], h.trace[it], ref.trace[it]
            # (the rider z1 = A x1_hat of fuse 2 is placed on the device, in the very step the host-driven loop would use)
            for f in ("cg_iters", "onsager_iters", "n_ax", "n_atx", "n_ax_pass", "n_atx_pass", "L_after"):
                assert td[f] == th[f], (fuse, it, f, td[f], th[f])
            assert (td["cg_iters"], td["onsager_iters"]) == (to["cg_iters"], to["onsager_iters"])
            assert np.isclose(td["gamw"], th["gamw"], rtol=1e-10) and np.isclose(td["gamw"], to["gamw"], rtol=1e-6)
        assert rel(d.x_est, h.x_est) < 1e-11 and rel(d.x_est, ref.x_est) < 1e-7
