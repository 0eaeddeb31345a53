"""Parity at BASELINE.json's full sizes, through properties that do not need the oracle to walk the whole matrix.

* sampled columns: the synthetic .bed generator is an integer hash of (marker, individual), so any marker of the
  400 000 x 1 000 000 matrix can be regenerated on the host on its own.  Marker statistics, A^T p and A x (x supported
  on the sample) of the full-size resident shard are compared with the oracle run on just those markers;
* adjoint identity <Ax, p> == <x, A^T p>: on the tile layout (what the library builds when nothing is configured) Ax and
  ATx read the same bytes through different lane patterns; on two stripe sets Ax reads the individual-major set and ATx
  the marker-major one -- either way the identity cross-checks both products over every byte;
* linearity, bitwise reproducibility, and the two-vector kernels against the one-vector ones;
* configs 2 / 4 / 5 and the 8-GPU shard shape (N=400k x M=125k): whole VAMP runs ON THE ENGINE THAT SHIPS -- layout auto
  (-> tile), --fuse-solves 4, device-resident CG -- against the reference's own sequence of products (level 0) and the
  bit-identical level 1, on both resident layouts; the shard shape also under the forced multi-rank branches.
"""
import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


_X_EST = {}          # level-4 estimates per (row, layout): the two resident layouts must agree bit for bit
LAYOUTS = [pytest.param(None, id="default-engine"), pytest.param(1, id="two-stripe-sets")]


def _configure(sh, layout):
    """layout None: NOTHING is configured -- kernel mode, resident layout (auto -> tile) and decomposition are the library's
    defaults, i.e. the engine a binding or bench.py gets.  layout 1: two stripe sets (2 x the bytes), pinned."""
    if layout is not None:
        sh.set_layout(False, layout)
        sh.set_kernel_mode(1)


def _same_counts(a, b):
    return (a["cg_iters"], a["onsager_iters"], a["revar_rounds"], a["L_after"]) == \
           (b["cg_iters"], b["onsager_iters"], b["revar_rounds"], b["L_after"])


def test_config3_full_matrix_sampled_columns_adjoint_linearity(oracle):
    N, Mt, seed, miss = 400000, 1000000, 20240601, 5000
    rng = np.random.default_rng(5)
    sample = np.array([0, 1, 63, 64, 255, 256, 32767, 32768, 65535, 499999, 500000, 524288, 777777, 999998, 999999])
    mini = np.concatenate([synth.synth_bed(N, 1, seed=seed, miss_ppm=miss, S=int(j)) for j in sample])
    o_mave, o_msig = oracle.marker_stats(mini, N, len(sample))
    with capi.Shard(N, Mt) as sh:           # nothing configured: the shipped engine, 100 GB resident on the tile layout
        sh.synth_bed(seed, miss)
        assert sh.get_kernel_mode() == 1 and sh.get_layout() == 2
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


@pytest.mark.parametrize("layout", LAYOUTS)
def test_config2_vamp_run_properties(layout):
    """BASELINE config 2 (N=100k x M=500k, CG-max-iter 50).  Level 4 is what the drivers and bench.py run; level 1 is
    bit-identical to the reference's own sequence (level 0)."""
    N, M, CV = 100000, 500000, 5000
    with capi.Shard(N, M) as sh:
        _configure(sh, layout)
        sh.synth_bed(424242, 5000)
        assert sh.get_kernel_mode() == 1 and sh.get_layout() == (2 if layout is None else layout)
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, CV, 7)
        kw = dict(iterations=4, CG_max_iter=50, rho=0.5, seed=7, true_signal=beta, history=False)
        r4 = hostapi.infere_linear(sh, y, None, None, fuse_solves=4, **kw)
        r0 = hostapi.infere_linear(sh, y, None, None, fuse_solves=0, **kw)
        r4b = hostapi.infere_linear(sh, y, None, None, fuse_solves=4, **kw)
        r1 = hostapi.infere_linear(sh, y, None, None, fuse_solves=1, **kw)
    assert r4.niter == r1.niter == r0.niter == 4
    assert np.array_equal(r4.x_est, r4b.x_est)                    # reproducible run to run
    assert np.array_equal(r1.x_est, r0.x_est)                     # level 1: shared passes change no bit
    assert rel(r4.x_est, r0.x_est) < 1e-9                         # level 4: by-products and identities, equal to rounding
    for a4, a, b in zip(r4.trace, r1.trace, r0.trace):
        assert _same_counts(a4, b) and _same_counts(a, b)
        assert a["n_ax"] == b["n_ax"] and a["n_atx"] == b["n_atx"]
        assert a4["n_ax_pass"] + a4["n_atx_pass"] <= a["n_ax_pass"] + a["n_atx_pass"] < b["n_ax_pass"] + b["n_atx_pass"]
        assert abs(a4["gamw"] - b["gamw"]) < 1e-9 * abs(b["gamw"])
        assert 0 < b["cg_iters"] <= 50
    # the estimate explains the simulated effects: correlation with the truth and a sane noise precision
    # (h2 = 0.5 on a standardised phenotype: gamw -> 1 / (1 - h2) = 2)
    assert np.corrcoef(r4.x_est, beta)[0, 1] > 0.5
    assert 1.5 < r4.trace[-1]["gamw"] < 2.6
    _X_EST.setdefault("config2", {})[layout] = r4.x_est


@pytest.mark.parametrize("layout", LAYOUTS)
def test_ragged_multi_chunk_shard_with_na_phenotypes(oracle, layout):
    """N % 4 = 3, M not a multiple of anything (7 ingest chunks of 32 768 markers, the last one partial), 1 % NA phenotypes,
    global marker offset S > 0: sampled columns against the oracle, the adjoint identity across Ax and ATx (one tile layout
    or two stripe sets), zeros at NA / pad slots."""
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
        _configure(sh, layout)
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


@pytest.mark.parametrize("layout", LAYOUTS)
def test_config4_probit_run_properties(layout):
    """BASELINE config 4 (probit, N=100k x M=500k): reproducible, the shipped level 4 and level 2 equal to the reference's
    sequence to rounding with identical step counts, and the estimate points along the simulated effects."""
    N, M = 100000, 500000
    with capi.Shard(N, M) as sh:
        _configure(sh, layout)
        sh.synth_bed(515151, 5000)
        assert sh.get_layout() == (2 if layout is None else layout)
        sh.compute_markers_statistics()
        beta, ylin = hostapi.sim_phen(sh, 0.5, 5000, 11)
        y = (ylin > 0).astype(float)
        kw = dict(iterations=4, CG_max_iter=50, rho=0.5, seed=11, gam1=1e-8, gamw=1.0, model="bin_class", history=False)
        r4 = hostapi.infere_linear(sh, y, None, None, fuse_solves=4, **kw)
        r4b = hostapi.infere_linear(sh, y, None, None, fuse_solves=4, **kw)
        r2 = hostapi.infere_linear(sh, y, None, None, fuse_solves=2, **kw)
        r0 = hostapi.infere_linear(sh, y, None, None, fuse_solves=0, **kw)
    assert r4.niter == r2.niter == r0.niter == 4
    assert np.array_equal(r4.x_est, r4b.x_est)
    assert rel(r4.x_est, r0.x_est) < 1e-9 and rel(r2.x_est, r0.x_est) < 1e-9
    for a4, a, b in zip(r4.trace, r2.trace, r0.trace):
        assert _same_counts(a4, b) and _same_counts(a, b)
        assert a4["n_ax_pass"] + a4["n_atx_pass"] <= a["n_ax_pass"] + a["n_atx_pass"] < b["n_ax_pass"] + b["n_atx_pass"]
    assert np.corrcoef(r4.x_est, beta)[0, 1] > 0.5
    _X_EST.setdefault("config4", {})[layout] = r4.x_est


def test_config5_people_statistics_kernel_families():
    """compute_people_statistics (data.cpp:558-716) at config 5's size: fp64 on the raw rows vs fixed point (MODE 4 pass)."""
    N, M = 50000, 200000
    with capi.Shard(N, M, anchor=True) as sh:
        sh.synth_bed(616161, 5000)
        sh.compute_markers_statistics()
        p0 = sh.compute_people_statistics()
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        p1 = sh.compute_people_statistics()
    for a, b in zip(p1, p0):
        assert np.allclose(a, b, rtol=1e-10, atol=1e-12)     # the means are sums that cancel to ~1e-3
    assert np.array_equal(p1[2], p0[2]) and p0[2].min() > 0.98 * M


@pytest.mark.parametrize("layout", LAYOUTS)
def test_config5_xxt_run_properties(layout):
    """BASELINE config 5 (--use-XXT-denoiser 1, N=50k; M=200k): the joint N-space / Onsager solver at the shipped level 4
    (A r2 by linearity, z1 riding the solve's first pass) and at level 2 against the reference sequence, and the Woodbury
    agreement with the M-space LMMSE path (two CG tolerances, 1e-4 / 1e-5)."""
    N, M = 50000, 200000
    with capi.Shard(N, M) as sh:
        _configure(sh, layout)
        sh.synth_bed(616161, 5000)
        assert sh.get_layout() == (2 if layout is None else layout)
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, 2000, 13)
        kw = dict(iterations=3, CG_max_iter=50, rho=0.5, seed=13, true_signal=beta, history=False)
        x4 = hostapi.infere_linear(sh, y, None, None, use_XXT_denoiser=1, fuse_solves=4, **kw)
        x2 = hostapi.infere_linear(sh, y, None, None, use_XXT_denoiser=1, fuse_solves=2, **kw)
        x0 = hostapi.infere_linear(sh, y, None, None, use_XXT_denoiser=1, fuse_solves=0, **kw)
        std = hostapi.infere_linear(sh, y, None, None, fuse_solves=4, **kw)
    assert rel(x4.x_est, x0.x_est) < 1e-9 and rel(x2.x_est, x0.x_est) < 1e-9
    for a4, a, b in zip(x4.trace, x2.trace, x0.trace):
        assert _same_counts(a4, b) and _same_counts(a, b)
        assert a4["n_ax_pass"] + a4["n_atx_pass"] <= a["n_ax_pass"] + a["n_atx_pass"] < 0.7 * (b["n_ax_pass"] + b["n_atx_pass"])
    assert rel(x4.x_est, std.x_est) < 2e-2
    _X_EST.setdefault("config5", {})[layout] = x4.x_est


SHARD8 = dict(N=400000, M=125000, Mt=1000000, S=375000, rank=3, seed=20240601, miss=5000)      # rank 3 of 8 by divide_work


@pytest.mark.parametrize("layout", LAYOUTS)
def test_8gpu_shard_shape_sampled_columns_and_vamp(oracle, layout):
    """One shard of BASELINE config 3 as an 8-GPU job holds it (utilities.cpp:259-291: Mt=1M over 8 ranks = 125 000 markers,
    rank 3 starts at S=375 000).  Sampled columns against the oracle -- the generator hashes the GLOBAL marker index, so the
    columns are those of the full matrix --, the adjoint identity, then whole VAMP runs at the shipped level 4 against the
    reference sequence."""
    c = SHARD8
    N, M, S = c["N"], c["M"], c["S"]
    rng = np.random.default_rng(17)
    sample = np.array([0, 1, 63, 64, 255, 256, 4095, 4096, 32767, 32768, 65535, 65536, 99999, 124928, 124998, 124999])
    mini = np.concatenate([synth.synth_bed(N, 1, seed=c["seed"], miss_ppm=c["miss"], S=S + int(j)) for j in sample])
    o_mave, o_msig = oracle.marker_stats(mini, N, len(sample))
    with capi.Shard(N, M, Mt=c["Mt"], S=S) as sh:
        _configure(sh, layout)
        sh.synth_bed(c["seed"], c["miss"])
        assert sh.get_kernel_mode() == 1 and sh.get_layout() == (2 if layout is None else layout)
        sh.compute_markers_statistics()
        mave, msig = sh.marker_stats()
        assert np.allclose(mave[sample], o_mave, rtol=1e-13, atol=1e-15)
        assert np.allclose(msig[sample], o_msig, rtol=1e-10, atol=0)
        p = rng.standard_normal(N)
        w = sh.ATx(p)
        assert rel(w[sample], oracle.atx(mini, N, len(sample), o_mave, o_msig, p)) < 1e-10
        xs = rng.standard_normal(len(sample))
        x = np.zeros(M)
        x[sample] = xs
        assert rel(sh.Ax(x), oracle.ax(mini, N, len(sample), o_mave, o_msig, xs)) < 1e-10
        x1 = rng.standard_normal(M)
        z1 = sh.Ax(x1)
        lhs, rhs = float(z1 @ p), float(x1 @ w)
        assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs))
        beta, y = hostapi.sim_phen(sh, 0.5, 1250, 1, rank=c["rank"])
        kw = dict(iterations=4, CG_max_iter=50, rho=0.5, seed=1, true_signal=beta, history=False, rank=c["rank"])
        r4 = hostapi.infere_linear(sh, y, None, None, fuse_solves=4, **kw)
        r0 = hostapi.infere_linear(sh, y, None, None, fuse_solves=0, **kw)
        r1 = hostapi.infere_linear(sh, y, None, None, fuse_solves=1, **kw)
    assert r4.niter == r0.niter == 4
    assert np.array_equal(r1.x_est, r0.x_est)
    assert rel(r4.x_est, r0.x_est) < 1e-9
    for a4, a, b in zip(r4.trace, r1.trace, r0.trace):
        assert _same_counts(a4, b) and _same_counts(a, b)
        assert a4["n_ax_pass"] + a4["n_atx_pass"] <= a["n_ax_pass"] + a["n_atx_pass"] < b["n_ax_pass"] + b["n_atx_pass"]
    assert np.corrcoef(r4.x_est, beta)[0, 1] > 0.4          # (1 250 causal markers of 125 000 at h2 = 0.5, four iterations)
    _X_EST.setdefault("shard8", {})[layout] = r4.x_est


@pytest.mark.parametrize("transport,delay_us,overlap", [(1, 0, 0), (1, 40, 4), (3, 0, 0)])
def test_8gpu_shard_shape_forced_multi_rank_branches(transport, delay_us, overlap):
    """The same shard, default engine, with the MULTI-RANK branches forced (gv_debug_force_multi: the in-stream exchange
    inside Ax = data.cpp:928/:995, packed scalar all-reduces = utilities.cpp:203, k_finalize + all-reduce in place of the
    one-rank shortcuts, the device opening's in-stream all-reduces; transport 3 = ncclAllReduce on a 1-rank communicator
    followed by the loop-back).  A sum over one rank is the identity: vamp::infere at level 4 must reproduce the plain run
    BIT FOR BIT at the size an 8-GPU job runs -- 12.5 GB per pass, 3.2 MB per exchange -- not only at toy sizes."""
    c = SHARD8
    N, M, S = c["N"], c["M"], c["S"]
    with capi.Shard(N, M, Mt=c["Mt"], S=S) as sh:
        sh.synth_bed(c["seed"], c["miss"])
        sh.compute_markers_statistics()
        beta, y = hostapi.sim_phen(sh, 0.5, 1250, 1, rank=c["rank"])
        kw = dict(iterations=3, CG_max_iter=50, rho=0.5, seed=1, true_signal=beta, history=False, rank=c["rank"], fuse_solves=4)
        plain = hostapi.infere_linear(sh, y, None, None, **kw)
        rng = np.random.default_rng(3)
        xv, zv = sh.vecM(rng.standard_normal(M)), sh.vecN()
        sh.ax_dev(xv, zv)
        z_plain = zv.download()
        sh.force_multi(transport, delay_us)
        sh.set_overlap(overlap)
        try:
            forced = hostapi.infere_linear(sh, y, None, None, **kw)
            sh.ax_dev(xv, zv)
            z_forced = zv.download()
        finally:
            sh.force_multi(0)
            sh.set_overlap(0)
        again = hostapi.infere_linear(sh, y, None, None, **kw)
    assert not np.isnan(forced.x_est).any(), "a consumer ran ahead of its exchange"
    assert np.array_equal(forced.x_est, plain.x_est) and np.array_equal(again.x_est, plain.x_est)
    assert np.array_equal(z_forced, z_plain)
    for a, b in zip(forced.trace, plain.trace):
        assert _same_counts(a, b) and a["gamw"] == b["gamw"] and a["gam1_next"] == b["gam1_next"]
        assert (a["n_ax_pass"], a["n_atx_pass"]) == (b["n_ax_pass"], b["n_atx_pass"])


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


@pytest.mark.parametrize("row", ["config2", "config4", "config5", "shard8"])
def test_layouts_agree_bit_for_bit_on_whole_runs(row):
    got = _X_EST.get(row, {})
    if len(got) < 2:
        pytest.skip("needs both layout legs of %s in this session" % row)
    assert np.array_equal(got[None], got[1])
