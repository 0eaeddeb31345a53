"""ONE resident layout (gv_set_layout(raw, 2): the "tile" layout, M*N/4 bytes) against the two stripe sets it replaces
(2 x M*N/4 bytes) and against the oracle.  Both run the same exact integer arithmetic, so every product must be
BIT-IDENTICAL between the two layouts -- at every shape, work decomposition and vector count."""
import os

import numpy as np
import pytest

from gvamp_amd import capi, hostapi, synth

pytestmark = pytest.mark.gpu
TOL = 1e-12


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def make_mask(N, rng, frac_na):
    present = rng.random(N) >= frac_na
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    return m4, int(present.sum()), present


def shard(bed, N, M, stripes, m4=None, nonas=None, raw=False):
    sh = capi.Shard(N, M)
    sh.set_layout(raw, stripes)
    sh.set_kernel_mode(1)
    sh.upload_bed(bed)
    if m4 is not None:
        sh.set_mask(m4, nonas)
    sh.compute_markers_statistics()
    return sh


CASES = [
    # N, M, miss_ppm, frac_na
    (2000, 300, 10000, 0.0),
    (1003, 129, 20000, 0.01),     # N % 4 != 0, NA phenotypes, M % 64 != 0
    (5, 3, 0, 0.0),               # tiny / ragged
    (4100, 64, 5000, 0.002),      # pitch padding
    (70000, 40, 5000, 0.0),       # many individual blocks, one marker group
    (300, 9000, 5000, 0.0),       # many marker groups, two individual blocks
    (1111, 2049, 30000, 0.05),
]


@pytest.mark.parametrize("N,M,miss,fna", CASES)
def test_tile_layout_products_bit_identical_to_two_layouts_and_vs_oracle(oracle, N, M, miss, fna):
    rng = np.random.default_rng(N * 7 + M)
    bed = synth.synth_bed(N, M, seed=99, miss_ppm=miss)
    m4, nonas, present = make_mask(N, rng, fna) if (fna > 0 or N % 4) else (None, N, np.ones(N, bool))
    n4 = 4 * ((N + 3) // 4)
    x, x2 = rng.standard_normal(M), rng.standard_normal(M) * 1e-5
    p, p2 = np.zeros(n4), np.zeros(n4)
    p[:N] = rng.standard_normal(N) * present
    p2[:N] = rng.standard_normal(N) * present * 1e3
    out = {}
    for stripes in (1, 2):
        with shard(bed, N, M, stripes, m4, nonas) as sh:
            mave, msig = sh.marker_stats()
            z, w = sh.Ax(x), sh.ATx(p)
            xa, xb, za, zb = sh.vecM(x), sh.vecM(x2), sh.vecN(), sh.vecN()
            sh.ax2_dev(xa, xb, za, zb)
            pa, pb, wa, wb = sh.vecN(p), sh.vecN(p2), sh.vecM(), sh.vecM()
            sh.atx2_dev(pa, pb, wa, wb)
            lm = sh.vecM()
            sh.lmmse_mult(xa, 1.3, 0.4, lm)
            ps = sh.compute_people_statistics()
            pv = sh.pvals_calc(za, pa, xb)
            out[stripes] = dict(mave=mave, msig=msig, z=z, w=w, za=za.download(), zb=zb.download(), wa=wa.download(),
                                wb=wb.download(), lm=lm.download(), ps=ps, pv=pv)
    a, b = out[1], out[2]
    for k in ("mave", "msig", "z", "w", "za", "zb", "wa", "wb", "lm", "pv"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k
    for q in range(3):
        assert np.array_equal(a["ps"][q], b["ps"][q], equal_nan=True), q
    assert np.array_equal(b["z"], b["za"]) and np.array_equal(b["w"], b["wa"])
    o_mave, o_msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    assert np.allclose(b["mave"], o_mave, rtol=1e-13, atol=1e-15) and np.allclose(b["msig"], o_msig, rtol=1e-12)
    assert rel(b["z"], oracle.ax(bed, N, M, o_mave, o_msig, x, mask4=m4)) < TOL
    assert rel(b["w"], oracle.atx(bed, N, M, o_mave, o_msig, p)) < TOL
    assert np.all(b["z"][N:] == 0) and np.all(b["z"][:N][~present] == 0)


@pytest.mark.parametrize("env", [{"GV_KS_M": "1", "GV_KS_N": "1"}, {"GV_KS_M": "3", "GV_KS_N": "5", "GV_TAPER": "0.9", "GV_PRIO": "1"},
                                 {"GV_SK_M": "768", "GV_SK_N": "768"}, {"GV_SK_M": "97", "GV_SK_N": "1536"},
                                 {"GV_HY_M": "60:33", "GV_HY_N": "5:11"}, {"GV_HY_M": "1:768", "GV_HY_N": "8:2", "GV_PRIO": "0"}])
def test_tile_layout_every_decomposition_gives_the_same_bits(env):
    """uniform / tapered / balanced decompositions, with and without wave priority, on a shape with several quads of row groups
    and enough K-steps on both sides"""
    N, M = 9000, 20000
    rng = np.random.default_rng(5)
    bed = synth.synth_bed(N, M, seed=7, miss_ppm=8000)
    x = rng.standard_normal(M)
    with shard(bed, N, M, 1) as sh:
        z0 = sh.Ax(x)
        w0 = sh.ATx(z0)
        xa, xb, za, zb = sh.vecM(x), sh.vecM(x[::-1].copy()), sh.vecN(), sh.vecN()
        sh.ax2_dev(xa, xb, za, zb)
        wa, wb = sh.vecM(), sh.vecM()
        sh.atx2_dev(za, zb, wa, wb)
        ref = (z0, w0, zb.download(), wb.download())
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    os.environ["GV_TUNE_CACHE"] = "0"
    try:
        with shard(bed, N, M, 2) as sh:
            z1 = sh.Ax(x)
            w1 = sh.ATx(z1)
            xa, xb, za, zb = sh.vecM(x), sh.vecM(x[::-1].copy()), sh.vecN(), sh.vecN()
            sh.ax2_dev(xa, xb, za, zb)
            wa, wb = sh.vecM(), sh.vecM()
            sh.atx2_dev(za, zb, wa, wb)
            got = (z1, w1, zb.download(), wb.download())
            assert all(v["tuned"] for v in sh.decomp().values())
            if "GV_HY_M" in env:      # the override took: some quads whole, the rest in balanced ranges
                assert all("whole_quads" in v for v in sh.decomp().values()), sh.decomp()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        os.environ.pop("GV_TUNE_CACHE", None)
    for r, g in zip(ref, got):
        assert np.array_equal(r, g)


def test_tile_layout_full_vamp_runs_equal_two_layout_runs(oracle):
    N, M = 2000, 6000
    bed = synth.synth_bed(N, M, seed=2024, miss_ppm=5000)
    probs, vars_ = [0.90, 0.07, 0.03], [0, 0.001, 0.01]
    res = {}
    for stripes in (1, 2):
        with capi.Shard(N, M) as sh:
            sh.set_layout(False, stripes)
            sh.set_kernel_mode(1)
            sh.upload_bed(bed)
            beta, y = hostapi.sim_phen(sh, 0.5, 300, 7)
            kw = dict(iterations=4, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8, gamw=2.0, true_signal=beta)
            res[stripes] = [hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=f, **kw) for f in (0, 2)]
            res[(stripes, "xxt")] = hostapi.infere_linear(sh, y, probs, vars_, fuse_solves=2, use_XXT_denoiser=1, **kw)
            yb = (y > 0).astype(float)
            res[(stripes, "probit")] = hostapi.infere_linear(sh, yb, probs, vars_, fuse_solves=2, model="bin_class",
                                                             iterations=3, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8, gamw=1.0)
    for f in (0, 1):
        assert np.array_equal(res[1][f].x_est, res[2][f].x_est)
        assert [t["cg_iters"] for t in res[1][f].trace] == [t["cg_iters"] for t in res[2][f].trace]
    assert np.array_equal(res[(1, "xxt")].x_est, res[(2, "xxt")].x_est)
    assert np.array_equal(res[(1, "probit")].x_est, res[(2, "probit")].x_est)
    ref = oracle.infere(bed, N, M, y, probs, vars_, iterations=4, CG_max_iter=25, rho=0.5, seed=7, gam1=1e-8, gamw=2.0,
                        true_signal=beta)
    assert rel(res[2][0].x_est, ref.x_est) < 1e-7


def test_tile_layout_halves_the_resident_bytes_and_mode0_still_needs_raw_rows():
    N, M = 4096, 8192
    bed = synth.synth_bed(N, M, seed=1)
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, 2)
        sh.set_kernel_mode(1)
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        assert sh.Ax(np.ones(M)).shape == (N,)
        sh.set_kernel_mode(0)
        with pytest.raises(capi.GvError, match="raw row layout"):
            sh.Ax(np.ones(M))
        with pytest.raises(capi.GvError):
            sh.set_layout(False, 4)


def test_reingest_under_the_automatic_choice_keeps_the_resident_layout():
    """gv_set_layout(.., 3) decides at the FIRST ingest; a later ingest on the same context (bench.py's LD leg, any second upload)
    keeps the layout that is resident -- a context that had to take the tile layout because two stripe sets did not fit must not
    tear it down and try the two stripe sets again.  Both directions, results unchanged."""
    N, M = 3000, 4096
    bed, bed2 = synth.synth_bed(N, M, seed=1), synth.synth_bed(N, M, seed=2)
    x = np.random.default_rng(0).standard_normal(M)
    for first in (2, 1):
        with capi.Shard(N, M) as sh:
            sh.set_layout(False, first)
            sh.upload_bed(bed)
            sh.compute_markers_statistics()
            z1 = sh.Ax(x)
            assert sh.get_layout() == first
            sh.set_layout(False, 3)                 # automatic from here on: a long run with plenty of free HBM, it would pick two sets
            sh.set_expected_passes(5000 if first == 2 else 0)   # (... and a run of unknown length the tile layout)
            sh.upload_bed(bed2)
            assert sh.get_layout() == first
            sh.compute_markers_statistics()
            z2 = sh.Ax(x)
            sh.upload_bed(bed)
            sh.compute_markers_statistics()
            assert np.array_equal(sh.Ax(x), z1) and not np.array_equal(z2, z1)
    with capi.Shard(N, M) as sh:                    # nothing configured: the automatic choice, one tile layout
        sh.upload_bed(bed)
        assert sh.get_layout() == 2 and sh.get_kernel_mode() == 1


@pytest.mark.parametrize("layout", [1, 2])
def test_pinned_decompositions_with_xcd_skew_and_occupancy_cap_give_the_same_bits(layout):
    """gv_set_decomp with the round-6 fields: xcd_skew (segment boundaries per quad parity: the workgroups with an odd block index
    get longer K-segments; also on an EVEN number of quads, where the quads are rotated by one per segment row) and wgs_per_cu = 2
    (dynamic LDS caps a CU at two workgroups).  Every product equals the library's own pick bit for bit."""
    for N, M in ((9000, 20000), (2048, 30000)):      # quads per side: ATx 79 / 118 (odd / even), Ax tile 9 / 2, stripes 36 / 8
        rng = np.random.default_rng(N)
        bed = synth.synth_bed(N, M, seed=7, miss_ppm=8000)
        x = rng.standard_normal(M)
        with shard(bed, N, M, layout) as sh:
            z0 = sh.Ax(x)
            w0 = sh.ATx(z0)
            xa, xb, za, zb, wa, wb = sh.vecM(x), sh.vecM(x[::-1].copy()), sh.vecN(), sh.vecN(), sh.vecM(), sh.vecM()
            sh.ax2_dev(xa, xb, za, zb)
            sh.atx2_dev(za, zb, wa, wb)
            ref = (z0, w0, zb.download(), wb.download())
            for kw in (dict(ks=2, xcd_skew=0.03), dict(ks=3, geo=0.5, prio=1, xcd_skew=0.2), dict(ks=5, taper=0.5, xcd_skew=-0.1),
                       dict(ks=4, geo=0.6, prio=1, wgs_per_cu=2, xcd_skew=0.025), dict(ks=1, wgs_per_cu=2), dict(balanced_cells=16, prio=1, wgs_per_cu=2)):
                took = 0
                for cls in ("atx", "atx2", "ax", "ax2"):
                    try:
                        sh.set_decomp(cls, **kw)
                    except capi.GvError:
                        continue              # more pieces than this small shard's partial-sum buffer holds: the class keeps its pick
                    took += 1
                    v = sh.decomp()[cls]
                    assert abs(v.get("xcd_skew", 0.0) - (kw.get("xcd_skew", 0.0) if "balanced_cells" not in kw else 0.0)) < 1e-6, (cls, v)
                    assert v.get("wgs_per_cu", 3) == (2 if kw.get("wgs_per_cu") == 2 else 3), (cls, v)
                assert took >= 2, kw
                z1 = sh.Ax(x)
                w1 = sh.ATx(z1)
                sh.ax2_dev(xa, xb, za, zb)
                sh.atx2_dev(za, zb, wa, wb)
                for r, g in zip(ref, (z1, w1, zb.download(), wb.download())):
                    assert np.array_equal(r, g), (N, M, kw)
            with pytest.raises(capi.GvError):
                sh.set_decomp("ax", ks=2, xcd_skew=0.5)
