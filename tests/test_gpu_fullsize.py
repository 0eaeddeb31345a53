"""Parity at BASELINE.json's full sizes, through properties that do not need the oracle to walk the whole matrix.

* sampled columns: the synthetic .bed generator is an integer hash of (marker, individual), so any marker of the
  400 000 x 1 000 000 matrix can be regenerated on the host on its own.  Marker statistics, A^T p and A x (x supported
  on the sample) of the full-size resident shard are compared with the oracle run on just those markers;
* adjoint identity <Ax, p> == <x, A^T p>: Ax reads the individual-major stripes, ATx the marker-major ones, so this
  cross-checks the two independently built layouts over every byte;
* linearity, bitwise reproducibility, and the two-vector kernels against the one-vector ones;
* config 2 (N=100k x M=500k, CG-max-iter 50): a VAMP run is reproducible, identical with and without shared passes,
  and recovers the simulated effects.
"""
import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_config3_full_matrix_sampled_columns_adjoint_linearity(oracle):
    N, Mt, seed, miss = 400000, 1000000, 20240601, 5000
    rng = np.random.default_rng(5)
    sample = np.array([0, 1, 63, 64, 255, 256, 32767, 32768, 65535, 499999, 500000, 524288, 777777, 999998, 999999])
    mini = np.concatenate([synth.synth_bed(N, 1, seed=seed, miss_ppm=miss, S=int(j)) for j in sample])
    o_mave, o_msig = oracle.marker_stats(mini, N, len(sample))
    with capi.Shard(N, Mt) as sh:
        sh.set_layout(False, True)          # stripes only: 2 x 100 GB resident
        sh.set_kernel_mode(1)
        sh.synth_bed(seed, miss)
        sh.compute_markers_statistics()
        mave, msig = sh.marker_stats()
        assert np.allclose(mave[sample], o_mave, rtol=1e-13, atol=1e-15)
        # the oracle (like data.cpp:392-546) accumulates 400 000 fp64 terms serially; the product derives the
        # statistics from exact genotype counts, so the difference is the oracle's own rounding (~N eps)
        assert np.allclose(msig[sample], o_msig, rtol=1e-10, atol=0)

        # A^T p on the sampled markers
        p = rng.standard_normal(N)
        w = sh.ATx(p)
        ow = oracle.atx(mini, N, len(sample), o_mave, o_msig, p)
        assert rel(w[sample], ow) < 1e-10

        # A x with x supported on the sample = the oracle's Ax over just those columns
        xs = rng.standard_normal(len(sample))
        x = np.zeros(Mt)
        x[sample] = xs
        z = sh.Ax(x)
        oz = oracle.ax(mini, N, len(sample), o_mave, o_msig, xs)
        assert rel(z, oz) < 1e-10

        # adjoint identity and linearity with dense vectors over the whole 100 GB
        x1, x2 = rng.standard_normal(Mt), rng.standard_normal(Mt)
        z1, z2 = sh.Ax(x1), sh.Ax(x2)
        lhs, rhs = float(z1 @ p), float(x1 @ w)
        assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))
        assert rel(sh.Ax(2.5 * x1 - 0.75 * x2), 2.5 * z1 - 0.75 * z2) < 1e-12
        assert np.array_equal(sh.Ax(x1), z1) and np.array_equal(sh.ATx(p), w)      # integer accumulation

        # two-vector passes equal the one-vector ones bit for bit
        va, vb, oa, ob = sh.vecM(x1), sh.vecM(x2), sh.vecN(), sh.vecN()
        sh.ax2_dev(va, vb, oa, ob)
        assert np.array_equal(oa.download()[:N], z1[:N]) and np.array_equal(ob.download()[:N], z2[:N])
        pa, pb, wa, wb = sh.vecN(p), sh.vecN(z1), sh.vecM(), sh.vecM()
        sh.atx2_dev(pa, pb, wa, wb)
        assert np.array_equal(wa.download(), w)
        assert np.array_equal(wb.download(), sh.ATx(z1))


def test_config2_vamp_run_properties():
    N, M, CV = 100000, 500000, 5000
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, True)
        sh.set_kernel_mode(1)
        sh.synth_bed(424242, 5000)
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, CV, 7)
        kw = dict(iterations=4, CG_max_iter=50, rho=0.5, seed=7, true_signal=beta, history=False)
        r1 = hostapi.infere_linear(sh, y, None, None, fuse_solves=1, **kw)
        r0 = hostapi.infere_linear(sh, y, None, None, fuse_solves=0, **kw)
        r2 = hostapi.infere_linear(sh, y, None, None, fuse_solves=1, **kw)
    assert r1.niter == r0.niter == 4
    assert np.array_equal(r1.x_est, r2.x_est)                     # reproducible run to run
    assert np.array_equal(r1.x_est, r0.x_est)                     # shared passes change no bit
    for a, b in zip(r1.trace, r0.trace):
        assert a["cg_iters"] == b["cg_iters"] and a["onsager_iters"] == b["onsager_iters"]
        assert a["n_ax"] == b["n_ax"] and a["n_atx"] == b["n_atx"]
        assert a["n_ax_pass"] < b["n_ax_pass"] and a["n_atx_pass"] < b["n_atx_pass"]
        assert 0 < a["cg_iters"] <= 50
    # the estimate explains the simulated effects: correlation with the truth and a sane noise precision
    # (h2 = 0.5 on a standardised phenotype: gamw -> 1 / (1 - h2) = 2)
    c = np.corrcoef(r1.x_est, beta)[0, 1]
    assert c > 0.5
    assert 1.5 < r1.trace[-1]["gamw"] < 2.6


def test_ragged_multi_chunk_shard_with_na_phenotypes(oracle):
    """N % 4 = 3, M not a multiple of anything (7 ingest chunks of 32 768 markers, the last one partial), 1 % NA phenotypes,
    global marker offset S > 0: sampled columns against the oracle, the adjoint identity across the two stripe layouts,
    zeros at NA / pad slots."""
    N, M, S, Mt, seed = 100003, 200001, 12345, 400000, 77
    rng = np.random.default_rng(9)
    present = rng.random(N) >= 0.01
    mb = (N + 3) // 4
    m4 = np.zeros(mb, dtype=np.uint8)
    idx = np.nonzero(present)[0]
    np.bitwise_or.at(m4, idx >> 2, (1 << (idx & 3)).astype(np.uint8))
    nonas = int(present.sum())
    sample = np.array([0, 1, 63, 64, 32767, 32768, 65535, 65536, 131071, 196607, 196608, 199999, 200000])
    mini = np.concatenate([synth.synth_bed(N, 1, seed=seed, miss_ppm=8000, S=S + int(j)) for j in sample])
    o_mave, o_msig = oracle.marker_stats(mini, N, len(sample), mask4=m4, nonas=nonas)
    with capi.Shard(N, M, Mt=Mt, S=S) as sh:
        sh.set_layout(False, True)
        sh.set_kernel_mode(1)
        sh.synth_bed(seed, 8000)
        sh.set_mask(m4, nonas)
        sh.compute_markers_statistics()
        mave, msig = sh.marker_stats()
        assert np.allclose(mave[sample], o_mave, rtol=1e-13, atol=1e-15)
        assert np.allclose(msig[sample], o_msig, rtol=1e-10, atol=0)
        p = np.zeros(4 * mb)
        p[:N] = rng.standard_normal(N) * present
        w = sh.ATx(p)
        assert rel(w[sample], oracle.atx(mini, N, len(sample), o_mave, o_msig, p)) < 1e-10
        xs = rng.standard_normal(len(sample))
        x = np.zeros(M)
        x[sample] = xs
        z = sh.Ax(x)
        oz = oracle.ax(mini, N, len(sample), o_mave, o_msig, xs, mask4=m4)
        # the oracle scales by its own 1/sqrt(N); A depends on Mt only through nothing else
        assert rel(z, oz) < 1e-10
        assert np.all(z[N:] == 0) and np.all(z[:N][~present] == 0)
        x1 = rng.standard_normal(M)
        z1 = sh.Ax(x1)
        lhs, rhs = float(z1 @ p), float(x1 @ w)
        assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))


def test_kernel_families_agree_on_a_vamp_run_at_scale():
    """A whole VAMP run (N=100k x M=200k, 5 GB shard) on the fp64 VALU family and on the i8 MFMA fixed-point family:
    same CG / EM counts, estimates equal to the ~1e-9 that iteration 1's cancellation leaves (docs/history/rounds1-3.md section 2)."""
    N, M = 100000, 200000
    with capi.Shard(N, M, anchor=True) as sh:
        sh.synth_bed(77, 5000)
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, 2000, 3)
        kw = dict(iterations=3, CG_max_iter=50, rho=0.5, seed=3, true_signal=beta, history=False)
        r0 = hostapi.infere_linear(sh, y, None, None, fuse_solves=0, **kw)           # kernel mode 0, reference sequence
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        r1 = hostapi.infere_linear(sh, y, None, None, fuse_solves=2, **kw)           # kernel mode 1, shared passes
    assert r0.niter == r1.niter == 3
    for a, b in zip(r0.trace, r1.trace):
        assert (a["cg_iters"], a["onsager_iters"], a["L_after"]) == (b["cg_iters"], b["onsager_iters"], b["L_after"])
        assert abs(a["gamw"] - b["gamw"]) < 1e-7 * abs(a["gamw"])
    assert rel(r1.x_est, r0.x_est) < 1e-7


def test_config4_probit_run_properties():
    """BASELINE config 4 (probit, N=100k x M=500k): reproducible, the same with and without shared passes / by-products to
    rounding, and the estimate points along the simulated effects."""
    N, M = 100000, 500000
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, True)
        sh.set_kernel_mode(1)
        sh.synth_bed(515151, 5000)
        sh.compute_markers_statistics()
        beta, ylin = hostapi.sim_phen(sh, 0.5, 5000, 11)
        y = (ylin > 0).astype(float)
        kw = dict(iterations=4, CG_max_iter=50, rho=0.5, seed=11, gam1=1e-8, gamw=1.0, model="bin_class", history=False)
        r2 = hostapi.infere_linear(sh, y, None, None, fuse_solves=2, **kw)
        r2b = hostapi.infere_linear(sh, y, None, None, fuse_solves=2, **kw)
        r0 = hostapi.infere_linear(sh, y, None, None, fuse_solves=0, **kw)
    assert r2.niter == r0.niter == 4
    assert np.array_equal(r2.x_est, r2b.x_est)
    assert rel(r2.x_est, r0.x_est) < 1e-9
    for a, b in zip(r2.trace, r0.trace):
        assert (a["cg_iters"], a["onsager_iters"]) == (b["cg_iters"], b["onsager_iters"])
        assert a["n_ax_pass"] + a["n_atx_pass"] < b["n_ax_pass"] + b["n_atx_pass"]
    assert np.corrcoef(r2.x_est, beta)[0, 1] > 0.5


def test_config5_xxt_run_properties():
    """BASELINE config 5 (--use-XXT-denoiser 1, N=50k; M=200k): the joint N-space / Onsager solver against the reference
    sequence, and the Woodbury agreement with the M-space LMMSE path (two CG tolerances, 1e-4 / 1e-5)."""
    N, M = 50000, 200000
    with capi.Shard(N, M, anchor=True) as sh:          # people statistics: fp64 on the raw rows vs fixed point on the stripes
        sh.synth_bed(616161, 5000)
        sh.compute_markers_statistics()
        p0 = sh.compute_people_statistics()
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        p1 = sh.compute_people_statistics()
    for a, b in zip(p1, p0):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-12)     # the means are sums that cancel to ~1e-3
    assert np.array_equal(p1[2], p0[2]) and p0[2].min() > 0.98 * M
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, True)                     # stripes only
        sh.synth_bed(616161, 5000)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, 2000, 13)
        kw = dict(iterations=3, CG_max_iter=50, rho=0.5, seed=13, true_signal=beta, history=False)
        x2 = hostapi.infere_linear(sh, y, None, None, use_XXT_denoiser=1, fuse_solves=2, **kw)
        x0 = hostapi.infere_linear(sh, y, None, None, use_XXT_denoiser=1, fuse_solves=0, **kw)
        std = hostapi.infere_linear(sh, y, None, None, fuse_solves=2, **kw)
    assert rel(x2.x_est, x0.x_est) < 1e-9
    for a, b in zip(x2.trace, x0.trace):
        assert (a["cg_iters"], a["onsager_iters"]) == (b["cg_iters"], b["onsager_iters"])
        assert a["n_ax_pass"] + a["n_atx_pass"] < 0.7 * (b["n_ax_pass"] + b["n_atx_pass"])
    assert rel(x2.x_est, std.x_est) < 2e-2


def test_tile_layout_holds_twice_the_shard(oracle):
    """N=400k x M=2M = 200 GB of 2-bit genotypes on ONE GPU: two stripe sets would need 400 GB of the 288; gv_set_layout(.., 3)
    (auto) falls back to the one tile layout.  Sampled columns vs the oracle, the adjoint identity over all 200 GB (Ax and
    ATx read the same bytes through different lane patterns), linearity, reproducibility, two-vector = one-vector."""
    N, Mt, seed, miss = 400000, 2000000, 777, 5000
    rng = np.random.default_rng(9)
    sample = np.array([0, 63, 64, 4095, 65536, 999999, 1000000, 1048576, 1999935, 1999999])
    mini = np.concatenate([synth.synth_bed(N, 1, seed=seed, miss_ppm=miss, S=int(j)) for j in sample])
    o_mave, o_msig = oracle.marker_stats(mini, N, len(sample))
    with capi.Shard(N, Mt) as sh:
        sh.set_layout(False, 3)
        sh.set_kernel_mode(1)
        sh.synth_bed(seed, miss)
        assert sh.get_layout() == 2                       # the two stripe sets did not fit
        sh.compute_markers_statistics()
        mave, msig = sh.marker_stats()
        assert np.allclose(mave[sample], o_mave, rtol=1e-13, atol=1e-15) and np.allclose(msig[sample], o_msig, rtol=1e-10)
        p = rng.standard_normal(N)
        w = sh.ATx(p)
        assert rel(w[sample], oracle.atx(mini, N, len(sample), o_mave, o_msig, p)) < 1e-10
        xs = rng.standard_normal(len(sample))
        x = np.zeros(Mt)
        x[sample] = xs
        assert rel(sh.Ax(x), oracle.ax(mini, N, len(sample), o_mave, o_msig, xs)) < 1e-10
        x1, x2 = rng.standard_normal(Mt), rng.standard_normal(Mt)
        z1, z2 = sh.Ax(x1), sh.Ax(x2)
        lhs, rhs = float(z1 @ p), float(x1 @ w)
        assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))
        assert rel(sh.Ax(2.5 * x1 - 0.75 * x2), 2.5 * z1 - 0.75 * z2) < 1e-12
        assert np.array_equal(sh.Ax(x1), z1) and np.array_equal(sh.ATx(p), w)
        va, vb, oa, ob = sh.vecM(x1), sh.vecM(x2), sh.vecN(), sh.vecN()
        sh.ax2_dev(va, vb, oa, ob)
        assert np.array_equal(oa.download()[:N], z1[:N]) and np.array_equal(ob.download()[:N], z2[:N])
    for passes, want in ((0, 2), (5000, 1)):              # a small shard: one layout unless the run is announced as long
        with capi.Shard(20000, 30000) as sh:
            sh.set_layout(False, 3)
            sh.set_expected_passes(passes)
            sh.set_kernel_mode(1)
            sh.synth_bed(1, 5000)
            assert sh.get_layout() == want
