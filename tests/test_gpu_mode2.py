"""Kernel mode 2 (gv_set_kernel_mode(ctx, 2)): data::Ax / data::ATx (data.cpp:848-1009, :810-835) in TWO-LEVEL fixed point -- the
vector's head digits and the digits of its exact fp64 residual in the two slots of one two-vector pass, over planes that give a
missing genotype an exact zero as the reference's table does (data.cpp:951-988).  The fast remedy for input whose dynamic range
defeats mode 1's one-exponent-per-vector contract (include/gvamp.h); until this round the only one was the fp64 VALU family
at 4-9 % of the HBM roofline.

* against the oracle on both resident layouts, ragged N, NA phenotypes, every work decomposition: 1e-12, bit-identical across layouts;
* the adversarial dynamic-range case of tests/test_gpu_matvec.py: per-entry relative accuracy < 1e-12 where mode 1 has 1e-9 .. 1e-3;
* vectors spanning 2^50 of dynamic range, entry by entry;
* lmmse_mult, the CG solvers (host-driven loops in this mode) and a whole VAMP run against the oracle.
"""
import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-12


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def _mask(N, fna, seed):
    rng = np.random.default_rng(seed)
    present = rng.random(N) >= fna
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    idx = np.nonzero(present)[0]
    np.bitwise_or.at(m4, idx >> 2, (1 << (idx & 3)).astype(np.uint8))
    return present, m4


@pytest.mark.parametrize("N,M,fna", [(2000, 1024, 0.0), (1003, 777, 0.02), (4097, 300, 0.0), (256, 5000, 0.01)])
def test_products_vs_oracle_both_layouts_bit_identical(oracle, N, M, fna):
    bed = synth.synth_bed(N, M, seed=N + M, miss_ppm=15000)
    present, m4 = _mask(N, fna, N)
    nonas = int(present.sum())
    rng = np.random.default_rng(M)
    x = rng.standard_normal(M)
    npad = 4 * ((N + 3) // 4)
    p = np.zeros(npad)
    p[:N] = rng.standard_normal(N) * present
    o_mave, o_msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    oz = oracle.ax(bed, N, M, o_mave, o_msig, x, mask4=m4)
    ow = oracle.atx(bed, N, M, o_mave, o_msig, p)
    got = {}
    for layout in (1, 2):
        with capi.Shard(N, M) as sh:
            sh.set_layout(False, layout)
            sh.upload_bed(bed)
            sh.set_mask(m4, nonas)
            sh.compute_markers_statistics()
            sh.set_kernel_mode(1)
            z1, w1 = sh.Ax(x), sh.ATx(p)
            sh.set_kernel_mode(2)
            assert sh.get_kernel_mode() == 2
            z, w = sh.Ax(x), sh.ATx(p)
            assert rel(z[:N], oz[:N]) < TOL and rel(w, ow) < TOL
            assert np.all(z[N:] == 0) and np.all(z[:N][~present] == 0)
            assert rel(z, z1) < 1e-13 and rel(w, w1) < 1e-13           # the two modes agree to mode 1's own accuracy
            assert np.array_equal(sh.Ax(x), z) and np.array_equal(sh.ATx(p), w)      # exact integer accumulation: reproducible
            assert np.all(sh.Ax(np.zeros(M)) == 0) and np.all(sh.ATx(np.zeros(npad)) == 0)
            # the device-pointer forms, and the two-vector entry points (two two-level passes in this mode)
            xa, xb, za, zb = sh.vecM(x), sh.vecM(-2.0 * x), sh.vecN(), sh.vecN()
            sh.ax2_dev(xa, xb, za, zb)
            assert np.array_equal(za.download(), z) and np.array_equal(zb.download(), -2.0 * z)     # (a power-of-two multiple: same digits)
            pa, pb, wa, wb = sh.vecN(p), sh.vecN(0.5 * p), sh.vecM(), sh.vecM()
            sh.atx2_dev(pa, pb, wa, wb)
            assert np.array_equal(wa.download(), w) and np.array_equal(wb.download(), 0.5 * w)
            # every work decomposition gives the same bits
            for cls, kws in (("ax2", [dict(ks=2), dict(ks=1, prio=1), dict(ks=2, geo=0.5, prio=1, wgs_per_cu=2), dict(balanced_cells=16, prio=1)]),
                             ("atx2", [dict(ks=2), dict(ks=3, taper=0.5), dict(balanced_cells=8, prio=1), dict(ks=1, wgs_per_cu=2)])):
                for kw in kws:
                    try:
                        sh.set_decomp(cls, **kw)
                    except capi.GvError:
                        continue                     # not admissible for this shape
                    assert np.array_equal(sh.Ax(x), z) and np.array_equal(sh.ATx(p), w), (cls, kw)
            got[layout] = (z, w)
    assert np.array_equal(got[1][0], got[2][0]) and np.array_equal(got[1][1], got[2][1])


def test_adversarial_dynamic_range_per_entry_accuracy(oracle):
    """tests/test_gpu_matvec.py::test_fixed_point_per_entry_bound_on_adversarial_dynamic_range in mode 2: one entry 2^45 above the
    rest on a marker with missing genotypes.  The individuals that do not see it (their genotype is missing there) get outputs
    accurate to THEIR OWN magnitude -- the reference's table gives those terms an exact 0 and so do the planes of this mode --,
    where mode 1 carries the quantisation of every other entry at the scale of the huge one."""
    N, M = 4000, 600
    rng = np.random.default_rng(5)
    bed = synth.synth_bed(N, M, seed=31, miss_ppm=20000).reshape(M, N // 4).copy()
    bed[7, :] = 0x00                                    # marker 7: monomorphic (every genotype a = 2, none missing)
    bed = bed.reshape(-1)
    codes = ((bed.reshape(M, N // 4)[:, :, None] >> (2 * np.arange(4))) & 3).reshape(M, N)
    j = 11
    miss_j = codes[j] == 1                              # PLINK 01 = missing
    assert miss_j.sum() >= 20
    for layout in (1, 2):
        with capi.Shard(N, M, anchor=True) as sh:
            sh.set_layout(True, layout)
            sh.upload_bed(bed)
            sh.compute_markers_statistics()
            mave, msig = sh.marker_stats()
            x = rng.standard_normal(M)
            x[j] = 2.0 ** 45
            ref = oracle.ax(bed, N, M, mave, msig, x)[:N]
            sh.set_kernel_mode(1)
            z1 = sh.Ax(x)[:N]
            sh.set_kernel_mode(2)
            z2 = sh.Ax(x)[:N]
            err1, err2 = np.abs(z1 - ref), np.abs(z2 - ref)
            rel_blind_1 = np.max(err1[miss_j] / np.abs(ref[miss_j]))
            rel_blind = np.max(err2[miss_j] / np.abs(ref[miss_j]))
            assert rel_blind_1 > 1e-9                       # mode 1: the documented deviation
            assert rel_blind < 1e-12, rel_blind             # mode 2: gone
            assert np.max(err2[~miss_j] / np.abs(ref[~miss_j])) < 1e-12
            # ATx: one individual 2^45 above the rest; a marker where that individual's genotype is missing does not see it, and the
            # monomorphic marker's exact output is 0 (sum a p - mave sum b p = 2 sum p - 2 sum p)
            p = np.zeros(N)
            p[:N] = rng.standard_normal(N)
            big = 123
            p[big] = 2.0 ** 45
            refT = oracle.atx(bed, N, M, mave, msig, p)
            w2 = sh.ATx(p)
            blind_m = codes[:, big] == 1
            assert blind_m.sum() >= 3
            assert np.max(np.abs(w2 - refT)[blind_m] / np.abs(refT[blind_m])) < 1e-11
            sees = ~blind_m
            sees[7] = False
            assert np.max(np.abs(w2 - refT)[sees] / np.abs(refT[sees])) < 1e-11
            # (both the reference's fp64 sums and these leave ~ulp(2^46) * msig / sqrt(N) where the exact answer is 0)
            assert abs(w2[7]) <= 64 * 2.0 ** (46 - 52) * msig[7] / np.sqrt(N)


def test_wide_dynamic_range_entry_by_entry(oracle):
    """Vectors whose entries span 2^50: every output entry against the oracle relative to the l1 mass of its OWN terms (what fp64
    summation itself guarantees), not to the vector's largest entry."""
    N, M = 3000, 900
    bed = synth.synth_bed(N, M, seed=77, miss_ppm=30000)
    codes = ((bed.reshape(M, (N + 3) // 4)[:, :, None] >> (2 * np.arange(4))) & 3).reshape(M, -1)[:, :N]
    a = np.where(codes == 0, 2.0, np.where(codes == 2, 1.0, 0.0))
    b = (codes != 1).astype(float)
    rng = np.random.default_rng(3)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        mave, msig = sh.marker_stats()
        sh.set_kernel_mode(2)
        x = rng.standard_normal(M) * 2.0 ** rng.integers(-50, 1, M)
        z = sh.Ax(x)[:N]
        oz = oracle.ax(bed, N, M, mave, msig, x)[:N]
        # own scale of output n: sum_i |(a_ni - mave_i) b_ni msig_i x_i| / sqrt(N)
        own = (np.abs((a - mave[:, None]) * b) * np.abs(msig * x)[:, None]).sum(axis=0) / np.sqrt(N)
        assert np.max(np.abs(z - oz) / own) < 1e-13
        p = rng.standard_normal(N) * 2.0 ** rng.integers(-50, 1, N)
        w = sh.ATx(p)
        ow = oracle.atx(bed, N, M, mave, msig, p)
        ownT = msig * (np.abs(a - mave[:, None]) * b * np.abs(p)[None, :]).sum(axis=1) / np.sqrt(N)
        assert np.max(np.abs(w - ow) / ownT) < 1e-13
        # mode 1 on the same input: fine in l2, not entry by entry
        sh.set_kernel_mode(1)
        z1 = sh.Ax(x)[:N]
        assert rel(z1, oz) < TOL


def test_lmmse_mult_and_cg_solvers_in_mode_2(oracle):
    """lmmse_mult (vamp.cpp:1074) and precondCG_solver (:1130-1229) through the host-driven loops (the device-resident loop is mode-1
    machinery): residual traces and step counts of the oracle; the dual solver too."""
    N, M = 2000, 3000
    bed = synth.synth_bed(N, M, seed=8, miss_ppm=10000)
    rng = np.random.default_rng(4)
    v = rng.standard_normal(M)
    u = np.sign(rng.standard_normal(M)) / np.sqrt(M)
    o_mave, o_msig = oracle.marker_stats(bed, N, M)
    tau, gam2 = 2.0, 1.35
    o_q = tau * oracle.atx(bed, N, M, o_mave, o_msig, oracle.ax(bed, N, M, o_mave, o_msig, v)) + gam2 * v
    o_mu, o_rr = oracle.cg_solve(bed, N, M, v, None, tau, gam2, 1, 30)
    o_mub, o_rrb = oracle.cg_solve(bed, N, M, u, None, tau, gam2, 0, 30)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        sh.set_kernel_mode(2)
        dv, du, out = sh.vecM(v), sh.vecM(u), sh.vecM()
        sh.lmmse_mult(dv, tau, gam2, out)
        assert rel(out.download(), o_q) < TOL
        mu = sh.vecM()
        st, rr = sh.cg_solve(dv, None, tau, gam2, 1, 30, mu)
        assert len(rr) == len(o_rr) and np.allclose(rr, o_rr, rtol=1e-9) and rel(mu.download(), o_mu) < 1e-11
        mu_a, mu_b = sh.vecM(), sh.vecM()
        (sa, ra), (sb, rb) = sh.cg_solve2(dv, None, du, tau, gam2, 30, mu_a, mu_b)
        assert len(ra) == len(o_rr) and len(rb) == len(o_rrb)
        assert rel(mu_a.download(), o_mu) < 1e-11 and rel(mu_b.download(), o_mub) < 1e-11


def test_vamp_run_in_mode_2_follows_the_oracle(oracle):
    """vamp::infere (vamp.cpp:190-803) with every product in two-level fixed point: the oracle's CG / EM counts and estimates."""
    N, M = 2000, 6000
    bed = synth.synth_bed(N, M, seed=2024, miss_ppm=5000)
    probs, vars_ = [0.90, 0.07, 0.03], [0, 0.001, 0.01]
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        beta, y = hostapi.sim_phen(sh, 0.5, 300, 7)
        kw = dict(iterations=4, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8, gamw=2.0, true_signal=beta)
        r1 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=0, **kw)
        sh.set_kernel_mode(2)
        sh.compute_markers_statistics()
        r2 = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=0, **kw)
        r2f = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=4, **kw)
    ref = oracle.infere(bed, N, M, y, probs, vars_, **kw)
    for a, b, f, o in zip(r1.trace, r2.trace, r2f.trace, ref.trace):
        assert (b["cg_iters"], b["onsager_iters"], b["L_after"]) == (a["cg_iters"], a["onsager_iters"], a["L_after"])
        assert (f["cg_iters"], f["onsager_iters"], f["L_after"]) == (a["cg_iters"], a["onsager_iters"], a["L_after"])
        assert b["cg_iters"] == int(o["cg_iters"])
        assert np.isclose(b["gamw"], o["gamw"], rtol=1e-6)
    assert rel(r2.x_est, ref.x_est) < 1e-7 and rel(r2.x_est, r1.x_est) < 1e-9 and rel(r2f.x_est, r2.x_est) < 1e-9


@pytest.mark.parametrize("transport", [1, 3])
def test_mode_2_takes_the_multi_rank_branches_bit_for_bit(transport):
    """gv_debug_force_multi in kernel mode 2: the exchange inside Ax (data.cpp:928/:995: partial sums un-scaled, all-reduce, 1/sqrt(N)
    afterwards), the packed scalars of the host-driven CG loops -- a sum over one rank is the identity, so every output equals the
    plain run's bit for bit."""
    N, M = 3001, 2500
    bed = synth.synth_bed(N, M, seed=5, miss_ppm=8000)
    present, m4 = _mask(N, 0.02, 3)
    rng = np.random.default_rng(1)
    x, v = rng.standard_normal(M), rng.standard_normal(M)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_mask(m4, int(present.sum()))
        sh.compute_markers_statistics()
        sh.set_kernel_mode(2)

        def run():
            z = sh.Ax(x)
            w = sh.ATx(z)
            dv, mu = sh.vecM(v), sh.vecM()
            st, rr = sh.cg_solve(dv, None, 2.0, 0.8, 1, 25, mu)
            return z, w, mu.download(), rr

        plain = run()
        sh.force_multi(transport, 20)
        try:
            forced = run()
        finally:
            sh.force_multi(0)
        for a, b in zip(plain, forced):
            assert not np.isnan(b).any() and np.array_equal(a, b)


def test_mode_2_at_the_8gpu_shard_shape_rate_and_parity(oracle):
    """N = 400k x M = 125k: sampled columns against the oracle in mode 2, and what the two-level pass costs beside mode 1's
    one-vector pass (asserted loosely: it is ONE two-vector pass, not two)."""
    N, M, S, seed, miss = 400000, 125000, 375000, 20240601, 5000
    rng = np.random.default_rng(17)
    sample = np.array([0, 1, 63, 64, 4095, 4096, 65535, 65536, 99999, 124928, 124999])
    mini = np.concatenate([synth.synth_bed(N, 1, seed=seed, miss_ppm=miss, S=S + int(j)) for j in sample])
    o_mave, o_msig = oracle.marker_stats(mini, N, len(sample))
    with capi.Shard(N, M, Mt=1000000, S=S) as sh:
        sh.synth_bed(seed, miss)
        sh.compute_markers_statistics()
        sh.set_kernel_mode(2)
        p = rng.standard_normal(N)
        w = sh.ATx(p)
        assert rel(w[sample], oracle.atx(mini, N, len(sample), o_mave, o_msig, p)) < 1e-10
        xs = rng.standard_normal(len(sample)) * 2.0 ** rng.integers(-40, 1, len(sample))
        x = np.zeros(M)
        x[sample] = xs
        assert rel(sh.Ax(x), oracle.ax(mini, N, len(sample), o_mave, o_msig, xs)) < 1e-10
        x1 = rng.standard_normal(M)
        z1 = sh.Ax(x1)
        lhs, rhs = float(z1 @ p), float(x1 @ w)
        assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))
        # cost: device-pointer products, mode 2 against mode 1
        xv, zv, pv, wv = sh.vecM(x1), sh.vecN(), sh.vecN(p), sh.vecM()
        t = {}
        for mode in (1, 2):
            sh.set_kernel_mode(mode)
            for _ in range(2):
                sh.ax_dev(xv, zv); sh.atx_dev(pv, wv)
            sh.set_timing(1)
            sh.counters(reset=True)
            for _ in range(5):
                sh.ax_dev(xv, zv); sh.atx_dev(pv, wv)
            c = sh.counters(reset=True)
            sh.set_timing(0)
            t[mode] = (c["ms_ax"] / 5, c["ms_atx"] / 5)
        print("mode 1 Ax %.3f ATx %.3f ms; mode 2 Ax %.3f ATx %.3f ms" % (t[1] + t[2]))
        assert t[2][0] < 1.6 * t[1][0] and t[2][1] < 1.6 * t[1][1]
