"""Two vectors per pass over the genotype shard: gv_ax2_dev / gv_atx2_dev must reproduce the one-vector products bit
for bit (exact integer accumulation), and gv_cg_solve2 (LMMSE + Onsager solves in lock-step) must reproduce two
stand-alone gv_cg_solve runs."""
import numpy as np
import pytest

from gvamp_amd import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,M,mode", [(2000, 3000, 1), (1003, 517, 1), (2000, 1000, 0)])
def test_two_vector_products_are_bit_identical(N, M, mode):
    rng = np.random.default_rng(N)
    bed = synth.synth_bed(N, M, seed=5, miss_ppm=10000)
    with capi.Shard(N, M) as sh:
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
