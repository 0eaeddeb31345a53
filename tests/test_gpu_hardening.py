"""Direct unit tests of entry points round 1 only covered end to end, and defined behaviour at the edges of the contract:
gv_lmmse_mult and gv_prior_estep against the oracle, unfiltered phenotypes (DBL_MAX at NA individuals, data.cpp:147) through
gv_atx / the p-value entry points, non-finite vector entries, work-vector allocation, the --kernel-mode 0 warning."""
import os
import subprocess
import sys

import numpy as np
import pytest

from gvamp_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("N,M,fna", [(2000, 700, 0.0), (1003, 257, 0.02)])
def test_lmmse_mult_vs_oracle(oracle, mode, N, M, fna):
    """vamp::lmmse_mult (vamp.cpp:1074-1118): tau A^T (A v) + gam2 v, the ATx epilogue fused in kernel mode 1."""
    rng = np.random.default_rng(N + M + mode)
    bed = synth.synth_bed(N, M, seed=12, miss_ppm=8000)
    m4, nonas, _ = make_mask(N, rng, fna) if (fna > 0 or N % 4) else (None, N, None)
    mave, msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    v = rng.standard_normal(M)
    tau, gam2 = 1.7, 0.42
    ref = tau * oracle.atx(bed, N, M, mave, msig, oracle.ax(bed, N, M, mave, msig, v, mask4=m4)) + gam2 * v
    with capi.Shard(N, M, anchor=(mode == 0)) as sh:
        sh.upload_bed(bed)
        if m4 is not None:
            sh.set_mask(m4, nonas)
        sh.set_kernel_mode(mode)
        sh.compute_markers_statistics()
        dv, out = sh.vecM(v), sh.vecM()
        sh.lmmse_mult(dv, tau, gam2, out)
        assert rel(out.download(), ref) < TOL
        assert np.array_equal(dv.download(), v)                       # the operand is borrowed, never written
        sh.lmmse_mult(sh.vecM(np.zeros(M)), tau, gam2, out)           # the reference's all-zero shortcut (:1079)
        assert np.all(out.download() == 0)


@pytest.mark.parametrize("L", [3, 23])
def test_prior_estep_vs_oracle_update_prior(oracle, L):
    """One E-step of vamp::updatePrior (vamp.cpp:953-1013) on the device, M-step restated from SURVEY Appendix A on its sums,
    against one EM round of the oracle's updatePrior; and the sums themselves against a dense numpy evaluation."""
    M = 30011
    rng = np.random.default_rng(L)
    if L == 3:
        probs, vars_ = np.array([0.9, 0.07, 0.03]), np.array([0.0, 2.0, 20.0])
    else:
        probs = np.concatenate([[0.7], 0.3 * np.full(L - 1, 1.0 / (L - 1))])
        vars_ = np.concatenate([[0.0], 1e-2 * 10.0 ** (np.arange(L - 1) * 4.0 / (L - 2))])   # ratios > 1.5: no merge
    comp = rng.choice(L, size=M, p=probs)
    gam1 = 3.5
    r1 = rng.standard_normal(M) * np.sqrt(vars_[comp] + 1.0 / gam1)
    lam = 1.0 - probs[0]
    omegas = probs.copy()
    omegas[1:] /= lam
    with capi.Shard(64, M) as sh:
        sums = sh.prior_estep(sh.vecM(r1), gam1, lam, omegas, vars_)
    # dense numpy evaluation of Appendix A
    nu, vmax = 1.0 / gam1, vars_.max()
    v = vars_[1:]
    num = lam * omegas[1:] * np.exp(-0.5 * r1[:, None] ** 2 * (vmax - v) / ((v + nu) * (vmax + nu))) / np.sqrt(v + nu) / np.sqrt(2 * np.pi)
    S = num.sum(1)
    beta = num / S[:, None]
    pin = 1.0 / (1.0 + (1 - lam) / np.sqrt(2 * np.pi * nu) * np.exp(-0.5 * r1 ** 2 * vmax / (nu * (nu + vmax))) / S)
    gam = gam1 * r1[:, None] / (1.0 / v + gam1)
    vhat = 1.0 / (1.0 / v + gam1)
    want = np.empty(1 + 2 * (L - 1))
    want[0] = pin.sum()
    want[1::2] = (beta * pin[:, None]).sum(0)
    want[2::2] = (beta * (gam ** 2 + vhat) * pin[:, None]).sum(0)
    assert np.allclose(sums, want, rtol=1e-11, atol=0)
    # M-step (vamp.cpp:1015-1022) on the device sums vs the oracle's one-round updatePrior
    Pi, R, Gs = sums[0], sums[1::2], sums[2::2]
    new_vars = np.concatenate([[0.0], Gs / R])
    new_lam = Pi / M
    new_probs = np.concatenate([[1 - new_lam], new_lam * R / Pi])
    o_probs, o_vars = oracle.update_prior(r1, M, gam1, probs, vars_, EM_max_iter=1, EM_err_thr=1e-30, learn_vars=1)
    assert len(o_probs) == L
    assert np.allclose(new_probs, o_probs, rtol=1e-10) and np.allclose(new_vars, o_vars, rtol=1e-10)


@pytest.mark.parametrize("mode", [0, 1])
def test_atx_of_an_unfiltered_phenotype_drops_the_na_individuals(oracle, mode):
    """data::get_phen() keeps DBL_MAX at NA individuals (data.cpp:147); data::dot_product applies no mask (data.cpp:728-801).
    gv_atx masks the staged copy, so ATx(get_phen()) is the ATx of the filtered phenotype instead of overflow garbage."""
    N, M = 1003, 300
    rng = np.random.default_rng(3)
    bed = synth.synth_bed(N, M, seed=4, miss_ppm=10000)
    m4, nonas, present = make_mask(N, rng, 0.03)
    p = np.zeros(4 * ((N + 3) // 4))
    p[:N] = np.where(present, rng.standard_normal(N), np.finfo(np.float64).max)
    p[N:] = 7.0                                                    # pad slots are not trusted either
    mave, msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    want = oracle.atx(bed, N, M, mave, msig, p, mask4=m4)
    assert np.all(np.isfinite(want))
    with capi.Shard(N, M, anchor=(mode == 0)) as sh:
        sh.upload_bed(bed)
        sh.set_mask(m4, nonas)
        sh.set_kernel_mode(mode)
        sh.compute_markers_statistics()
        assert rel(sh.ATx(p), want) < TOL
        # p-values with an unfiltered y (DBL_MAX at NA individuals): masked like the reference's na_lut (data.cpp:1155-1175)
        x1 = rng.standard_normal(M) * (rng.random(M) < 0.05)
        z1 = sh.vecN()
        sh.ax_dev(sh.vecM(x1), z1)
        y = np.zeros(4 * ((N + 3) // 4))
        y[:N] = np.where(present, rng.standard_normal(N), np.finfo(np.float64).max)
        pv = sh.pvals_calc(z1, sh.vecN(y), sh.vecM(x1))
        yf = np.where(np.arange(y.size) < N, y, 0.0) * np.concatenate([present, np.zeros(y.size - N, bool)])
        opv = oracle.pvals(bed, N, M, z1.download(), yf, x1, mask4=m4, nonas=nonas)
        assert np.all(np.isfinite(pv)) and np.allclose(pv, opv, rtol=1e-8, atol=1e-300)


@pytest.mark.parametrize("bad", [np.nan, np.inf, -np.inf])
def test_non_finite_entries_give_nan_not_garbage(bad):
    """Fixed point cannot encode NaN / inf.  Defined behaviour of kernel mode 1, both directions: a non-finite entry anywhere
    in the operand makes EVERY entry of the product NaN (what the reference's fp64 sums over the whole vector give), masked /
    pad slots of an N-space result stay 0, and the context stays usable.  Kernel mode 0 propagates through fp64 arithmetic."""
    N, M = 1500, 400
    rng = np.random.default_rng(1)
    bed = synth.synth_bed(N, M, seed=9, miss_ppm=5000)
    m4, nonas, present = make_mask(N, rng, 0.01)
    with capi.Shard(N, M, anchor=True) as sh:       # the raw rows too: the fp64 family is looked at last
        sh.upload_bed(bed)
        sh.set_mask(m4, nonas)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        x = rng.standard_normal(M)
        good = sh.Ax(x)
        xb = x.copy()
        xb[137] = bad
        z = sh.Ax(xb)
        assert np.all(np.isnan(z[:N][present])) and np.all(z[:N][~present] == 0) and np.all(z[N:] == 0)
        p = np.zeros(4 * ((N + 3) // 4))
        p[:N] = rng.standard_normal(N) * present
        pb = p.copy()
        pb[np.nonzero(present)[0][5]] = bad
        assert np.all(np.isnan(sh.ATx(pb)))
        xa, xbv, za, zb = sh.vecM(x), sh.vecM(xb), sh.vecN(), sh.vecN()
        sh.ax2_dev(xa, xbv, za, zb)                                   # two-vector pass: the clean vector is untouched
        assert np.array_equal(za.download(), good) and np.all(np.isnan(zb.download()[:N][present]))
        assert np.array_equal(sh.Ax(x), good)                          # nothing sticks
        sh.set_kernel_mode(0)
        sh.compute_markers_statistics()
        z0 = sh.Ax(xb)
        # fp64 table kernels: the bad entry reaches every individual whose genotype at that marker is not missing (a missing
        # genotype is skipped, not multiplied by 0 as in the reference) -- ~99 % of them here
        assert np.mean(~np.isfinite(z0[:N][present])) > 0.95


def test_huge_finite_entries_still_work(oracle):
    """|v| up to DBL_MAX / 4 is finite input: the scale follows it (round 1 switched the kernel off above 1.7e308 / 2)."""
    N, M = 800, 200
    rng = np.random.default_rng(2)
    bed = synth.synth_bed(N, M, seed=6, miss_ppm=0)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.set_kernel_mode(1)
        sh.compute_markers_statistics()
        mave, msig = sh.marker_stats()
        p = np.zeros(N)
        p[:] = rng.standard_normal(N) * 1e300
        w = sh.ATx(p)
        assert np.all(np.isfinite(w)) and rel(w * 1e-300, oracle.atx(bed, N, M, mave, msig, p) * 1e-300) < TOL   # (norms of 1e300 overflow)


def test_kernel_mode_0_warns_in_the_drivers(tmp_path):
    N, M = 400, 300
    bedp = str(tmp_path / "t.bed")
    synth.write_bed(bedp, synth.synth_bed(N, M, seed=2))
    base = [os.path.join(ROOT, "gvamp_amd", "gvamp_sim"), "--bed-file", bedp, "--N", str(N), "--Mt", str(M), "--out-dir",
            str(tmp_path) + "/", "--out-name", "t", "--iterations", "1", "--probs", "0.9,0.1", "--vars", "0,0.01", "--CV", "10",
            "--h2", "0.5", "--model", "linear", "--store-pvals", "0"]
    r0 = subprocess.run(base + ["--kernel-mode", "0"], capture_output=True, text=True, timeout=300)
    r1 = subprocess.run(base + ["--kernel-mode", "1"], capture_output=True, text=True, timeout=300)
    assert r0.returncode == 0 and r1.returncode == 0
    assert "--kernel-mode 0" in r0.stderr and "slower" in r0.stderr and "WARNING" not in r1.stderr


def test_decomposition_picks_are_cached_across_processes(tmp_path):
    """The first run on a (device, N, M, layout) measures the work decompositions and appends them to the cache file; the next
    process reads them back instead of measuring (gv_tune_info).  Results never depend on the picks."""
    import json
    import sys
    code = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np
from gvamp_amd import capi
out = {}
for stripes in (1, 2):
    with capi.Shard(20000, 30000) as sh:
        sh.set_layout(False, stripes)
        sh.set_kernel_mode(1)
        sh.synth_bed(5, 5000)
        sh.compute_markers_statistics()
        assert sh.tune_info()[1] == "pending"
        z = sh.Ax(np.ones(30000))
        sec, src = sh.tune_info()
        out[str(stripes)] = {"src": src, "sec": sec, "picks": sh.decomp(), "norm": float(np.linalg.norm(z))}
print(json.dumps(out))
""" % ROOT
    env = dict(os.environ, GV_TUNE_CACHE_DIR=str(tmp_path / "cache"))
    runs = []
    for _ in range(2):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        runs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    for k in ("1", "2"):
        assert runs[0][k]["src"] == "measured" and runs[0][k]["sec"] > 0
        assert runs[1][k]["src"] == "cache" and runs[1][k]["sec"] == 0
        assert runs[0][k]["picks"] == runs[1][k]["picks"] and runs[0][k]["norm"] == runs[1][k]["norm"]
    assert runs[0]["1"]["norm"] == runs[0]["2"]["norm"]
    lines = open(tmp_path / "cache" / "decomp.txt").read().splitlines()
    assert len(lines) == 2 and "|L0|" in lines[0] and "|L1|" in lines[1]
    env["GV_TUNE_CACHE"] = "0"                                     # switched off: measured again, nothing written
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert json.loads(r.stdout.strip().splitlines()[-1])["1"]["src"] == "measured"
    assert len(open(tmp_path / "cache" / "decomp.txt").read().splitlines()) == 2


def test_shipped_picks_cover_the_baseline_shapes(tmp_path):
    """gv_tune_builtin.h: on the shapes BASELINE.json names (config 1 here: N=2000 x M=10000) a cold run -- no cache -- takes its
    decompositions from the table shipped with the library instead of measuring (gv_tune_info source "builtin", 0 s);
    GV_TUNE_BUILTIN=0 measures; the cache wins over the table; the bits are the same either way."""
    import json
    import re
    import sys
    from gvamp_amd import build
    txt = open(os.path.join(ROOT, "gvamp_amd", "csrc", "gv_tune_builtin.h")).read()
    if re.search(r'GV_BUILTIN_FOR_HASH = "([0-9a-f]+|none)"', txt).group(1) != build.kernel_src_hash()[:16]:
        pytest.skip("gv_tune_builtin.h was measured for other kernel sources (scripts/tune_table.py regenerates it on an MI355X)")
    code = r"""
import json, sys
sys.path.insert(0, %r)
import numpy as np
from gvamp_amd import capi
out = {}
for stripes in (1, 2):
    with capi.Shard(2000, 10000) as sh:
        sh.set_layout(False, stripes)
        sh.synth_bed(5, 5000)
        sh.compute_markers_statistics()
        z = sh.Ax(np.ones(10000))
        sec, src = sh.tune_info()
        out[str(stripes)] = {"src": src, "sec": sec, "norm": float(np.linalg.norm(z))}
print(json.dumps(out))
""" % ROOT
    def run(**extra):
        env = dict(os.environ, GV_TUNE_CACHE_DIR=str(tmp_path / "cache"), **extra)
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    cold = run(GV_TUNE_CACHE="0")
    assert all(cold[k]["src"] == "builtin" and cold[k]["sec"] == 0 for k in ("1", "2")), cold
    meas = run(GV_TUNE_CACHE="0", GV_TUNE_BUILTIN="0")
    assert all(meas[k]["src"] == "measured" for k in ("1", "2"))
    assert all(meas[k]["norm"] == cold[k]["norm"] for k in ("1", "2"))
    run(GV_TUNE_BUILTIN="0")                                        # measures and fills the cache ...
    assert run()["1"]["src"] == "cache"                             # ... which is consulted before the table


@pytest.mark.parametrize("stripes", [1, 2])
def test_bed_file_slab_equals_upload_from_memory_and_a_short_file_fails_loudly(tmp_path, stripes):
    """read_genotype_data (data.cpp:201-234): a rank's slab at byte offset 3 + S*mbytes of the .bed, streamed through the
    double-buffered pinned staging (several chunks, several reader threads) -- the same resident bytes as an upload from
    memory; a file that ends inside the slab, or does not exist, is an error, never zero-filled genotypes."""
    N, Mt = 4101, 20000                       # mbytes = 1026: 3 staging chunks of 8192 markers = 8.4 MB each, 4 reader threads
    S, M = 700, 19000
    bed = synth.synth_bed(N, Mt, seed=5, miss_ppm=7000)
    mb = (N + 3) // 4
    path = str(tmp_path / "f.bed")
    synth.write_bed(path, bed)
    x = np.random.default_rng(0).standard_normal(M)
    res = []
    for from_file in (False, True):
        with capi.Shard(N, M, Mt=Mt, S=S) as sh:
            sh.set_layout(stripes == 1, stripes)
            sh.set_kernel_mode(1)
            if from_file:
                sh.upload_bed_file(path)                      # default offset: 3 + S * mbytes
            else:
                sh.upload_bed(bed[S * mb:(S + M) * mb])
            sh.compute_markers_statistics()
            res.append((sh.marker_stats(), sh.Ax(x), sh.download_bed() if stripes == 1 else None))
    assert np.array_equal(res[0][0][0], res[1][0][0]) and np.array_equal(res[0][0][1], res[1][0][1])
    assert np.array_equal(res[0][1], res[1][1])
    if stripes == 1:
        assert np.array_equal(res[1][2], bed[S * mb:(S + M) * mb])
    short = str(tmp_path / "short.bed")
    with open(short, "wb") as f:
        f.write(open(path, "rb").read()[:3 + (S + M) * mb - 1000])
    with capi.Shard(N, M, Mt=Mt, S=S) as sh:
        sh.set_layout(False, stripes)
        with pytest.raises(capi.GvError, match="short file"):
            sh.upload_bed_file(short)
        with pytest.raises(capi.GvError, match="could not open"):
            sh.upload_bed_file(str(tmp_path / "nope.bed"))
        with pytest.raises(capi.GvError):                     # nothing usable is resident after the failed ingest
            sh.compute_markers_statistics()


@pytest.mark.parametrize("N,M", [(3001, 70001), (1200, 300000)])
def test_gathered_dots_equal_their_own_launches_bit_for_bit(N, M):
    """gv_vec_dots_ex (the iteration's tail in one launch and one read-back, host/vamp.cpp) against the launches it replaces:
    gv_vec_axpby(t, 1, xa, -1, xb) + gv_vec_dot / gv_vec_dots per scalar -- mixed spaces and lengths (12 to 1024 reduction blocks),
    plain and difference operands, squares and mixed products: the same bits."""
    rng = np.random.default_rng(N + M)
    with capi.Shard(N, M) as sh:
        sh.synth_bed(1)
        sh.compute_markers_statistics()
        a, b, c = (sh.vecM(rng.standard_normal(M)) for _ in range(3))
        npad = 4 * ((N + 3) // 4)                      # N-space handles hold 4 * mbytes entries, the tail at zero

        def nvec():
            h = np.zeros(npad)
            h[:N] = rng.standard_normal(N) * 1e3
            return sh.vecN(h)
        u, v = nvec(), nvec()
        tM, tM2, tN = sh.vecM(), sh.vecM(), sh.vecN()
        want = []
        sh.axpby(tN, 1.0, u, -1.0, v)
        want.append(sh.dot(tN, tN, 0))                     # <u - v, u - v>
        want.append(sh.dot(u, u, 0))                       # <u, u>
        want.append(sh.dot(a, b, 1))                       # <a, b>
        sh.axpby(tM, 1.0, a, -1.0, b)
        want.append(sh.dot(tM, tM, 1))                     # <a - b, a - b>
        sh.axpby(tM2, 1.0, c, -1.0, a)
        want.append(sh.dot(tM, tM2, 1))                    # <a - b, c - a>
        want.append(sh.dot(tM, c, 1))                      # <a - b, c>
        got = sh.dots_ex([(u, v, u, v, 0), (u, None, u, None, 0), (a, None, b, None, 1), (a, b, a, b, 1), (a, b, c, a, 1),
                          (a, b, c, None, 1)])
        assert [float(x).hex() for x in got] == [float(x).hex() for x in want]
        # and against numpy, to a rounding-level tolerance
        A, B, Cc, U, V = (x.download() for x in (a, b, c, u, v))
        ref = [np.dot(U - V, U - V), np.dot(U, U), np.dot(A, B), np.dot(A - B, A - B), np.dot(A - B, Cc - A), np.dot(A - B, Cc)]
        assert np.allclose(got, ref, rtol=1e-12, atol=1e-9)
        with pytest.raises(capi.GvError, match="one space"):
            sh.dots_ex([(a, u, a, u, 0)])


def test_stripe_sets_survive_a_reingest():
    """The two stripe sets are views into one allocation (gv_capi.hip: where the driver places a large allocation moves the kernel
    that streams it).  A second ingest into the same context releases and re-allocates: same products for the same matrix."""
    N, M = 5003, 20011
    rng = np.random.default_rng(5)
    x = rng.standard_normal(M)
    p = np.zeros(4 * ((N + 3) // 4))
    p[:N] = rng.standard_normal(N)
    with capi.Shard(N, M) as sh:
        sh.set_layout(False, 1)
        sh.synth_bed(3, 5000)
        sh.compute_markers_statistics()
        a1, t1 = sh.Ax(x), sh.ATx(p)
        sh.synth_bed(4, 5000)                      # re-ingest: the resident layouts are rebuilt in place
        sh.compute_markers_statistics()
        a2 = sh.Ax(x)
        sh.synth_bed(3, 5000)
        sh.compute_markers_statistics()
        a3, t3 = sh.Ax(x), sh.ATx(p)
        assert np.array_equal(a1, a3) and np.array_equal(t1, t3) and not np.array_equal(a1, a2)


def _estep_case(M, L, seed):
    rng = np.random.default_rng(seed)
    r1 = rng.standard_normal(M) * 0.3
    vars_ = np.concatenate([[0.0], np.sort(10.0 ** rng.uniform(-4, -1, L - 1))])
    probs = np.concatenate([[0.9], rng.dirichlet(np.ones(L - 1)) * 0.1])
    lam = 1 - probs[0]
    omegas = probs.copy()
    omegas[1:] /= lam
    return r1, vars_, lam, omegas


def _estep_formulas(r1, gam1, lam, omegas, vars_):
    """the E-step sums of vamp.cpp:953-1013 evaluated in numpy"""
    L = vars_.size
    nu, vmax = 1.0 / gam1, vars_.max()
    v = vars_[1:][None, :]
    num = lam * omegas[1:][None, :] * np.exp(-0.5 * r1[:, None] ** 2 * (vmax - v) / ((v + nu) * (vmax + nu))) / np.sqrt(v + nu) / np.sqrt(2 * np.pi)
    S = num.sum(axis=1)
    beta = num / S[:, None]
    pin = 1.0 / (1.0 + (1 - lam) / np.sqrt(2 * np.pi * nu) * np.exp(-0.5 * r1 ** 2 * vmax / (nu * (nu + vmax))) / S)
    mean, var = gam1 * r1[:, None] / (1.0 / v + gam1), 1.0 / (1.0 / v + gam1)
    want = np.empty(1 + 2 * (L - 1))
    want[0] = pin.sum()
    want[1::2] = (beta * pin[:, None]).sum(axis=0)
    want[2::2] = (beta * (mean ** 2 + var) * pin[:, None]).sum(axis=0)
    return want


@pytest.mark.parametrize("L", [2, 3, 5, 6, 9, 12, 17, 18, 23, 26, 32])
def test_prior_estep_every_instantiation_vs_the_formulas(L):
    """gv_prior_estep keeps its per-thread accumulators in registers, instantiated for up to 5, 9, 17, 25 and 32 components: each
    against the E-step sums of vamp.cpp:953-1013 evaluated in numpy, and reproducible bit for bit."""
    M = 70001
    r1, vars_, lam, omegas = _estep_case(M, L, L)
    gam1 = 2.5
    out = []
    with capi.Shard(2000, M) as sh:
        sh.synth_bed(1)
        sh.compute_markers_statistics()
        dr = sh.vecM(r1)
        for _ in range(2):
            out.append(np.array(sh.prior_estep(dr, gam1, lam, omegas, vars_)))
    want = _estep_formulas(r1, gam1, lam, omegas, vars_)
    assert out[0].shape == (1 + 2 * (L - 1),) and np.all(np.isfinite(out[0]))
    assert np.allclose(out[0], want, rtol=1e-11, atol=1e-13 * want[0])
    assert [float(v).hex() for v in out[0]] == [float(v).hex() for v in out[1]]


@pytest.mark.parametrize("M,L", [(1, 3), (255, 5), (257, 17), (262145, 23), (600001, 17), (600001, 32), (1000003, 23)])
def test_prior_estep_and_denoiser_block_shapes_vs_the_formulas(oracle, M, L):
    """The E-step runs in 256-thread blocks, one element per thread up to 262 144 markers and a strided loop beyond (1 024 blocks at
    most), its numerators staged in LDS and its block sums taken through LDS rows; the denoiser keeps per-block component tables.
    One marker, one block short of / past a block edge, the first size that loops, config 2's and the headline's sizes: the sums
    against the formulas, x1 / g1d against the oracle's vamp::g1 / g1d, and bit-reproducible."""
    r1, vars_, lam, omegas = _estep_case(M, L, 1000 + L)
    probs = omegas.copy()
    probs[1:] *= lam
    gam1 = 2.5
    with capi.Shard(256, M) as sh:
        dr, x1, dd = sh.vecM(r1), sh.vecM(), sh.vecM()
        e = [np.array(sh.prior_estep(dr, gam1, lam, omegas, vars_)) for _ in range(2)]
        s = [np.array(sh.denoise(dr, gam1, probs, vars_, x1, dd)) for _ in range(2)]
        gx, gd = x1.download(), dd.download()
    want = _estep_formulas(r1, gam1, lam, omegas, vars_)
    assert np.allclose(e[0], want, rtol=1e-11, atol=1e-13 * want[0])
    ox, od = oracle.g1_g1d(r1, gam1, probs, vars_)
    assert np.allclose(gx, ox, rtol=1e-13, atol=1e-15) and np.allclose(gd, od, rtol=1e-11, atol=1e-13)
    assert np.isclose(s[0][0], od.sum(), rtol=1e-12) and np.isclose(s[0][1], ((ox - r1) ** 2).sum(), rtol=1e-12)
    assert [float(v).hex() for v in e[0]] == [float(v).hex() for v in e[1]] and [float(v).hex() for v in s[0]] == [float(v).hex() for v in s[1]]


def test_auto_layout_weighs_the_length_of_the_run_and_ingest_reports_its_parts(tmp_path):
    """gv_set_expected_passes: with the automatic layout a short run (< 1000 ATx passes) or one that says nothing about its length
    takes the one tile layout -- half the bytes to allocate and fill -- although two stripe sets would fit; an announced long run
    takes the two sets; an explicit layout is never overruled; results are the same bits either way.  gv_ingest_info2: the allocation of the resident layout runs beside the
    preparation of the source (a file: the pinned staging buffers and the first two chunks), and what that hid is reported."""
    N, M = 4101, 20000
    bed = synth.synth_bed(N, M, seed=9, miss_ppm=5000)
    path = str(tmp_path / "g.bed")
    synth.write_bed(path, bed)
    x = np.random.default_rng(1).standard_normal(M)
    got = {}
    for name, lay, passes in (("unknown", 3, 0), ("short", 3, 60), ("long", 3, 5000), ("explicit", 1, 60)):
        with capi.Shard(N, M) as sh:
            sh.set_layout(False, lay)
            sh.set_expected_passes(passes)
            sh.upload_bed_file(path, offset=3)
            st = sh.ingest_stats()
            sh.compute_markers_statistics()
            got[name] = (sh.get_layout(), sh.Ax(x), st)
    assert [got[k][0] for k in ("unknown", "short", "long", "explicit")] == [2, 2, 1, 1]
    for k in ("short", "long", "explicit"):
        assert np.array_equal(got[k][1], got["unknown"][1])
    mb = (N + 3) // 4
    for k, (lay, _, st) in got.items():
        assert st["layout"] == lay and st["alloc_s"] >= st["overlap_s"] >= 0.0 and st["fill_s"] > 0.0
        assert st["resident_GB"] * 1e9 >= (1 if lay == 2 else 2) * M * mb        # (padded to whole 4 KiB blocks)
    assert got["short"][2]["resident_GB"] < 0.6 * got["long"][2]["resident_GB"]
    assert got["unknown"][2]["resident_GB"] == got["short"][2]["resident_GB"]
    assert got["short"][2]["expected_passes"] == 60
    with capi.Shard(N, M) as sh:
        with pytest.raises(capi.GvError, match="passes >= 0"):
            sh.set_expected_passes(-1)


def test_bind_host_numa_keeps_a_usable_affinity():
    """gv_bind_host_numa (one process per GPU on a multi-socket node): in a child process, so that the test runner keeps its own
    CPUs.  On a single-node host it changes nothing and says so (-1); where it binds, the node's CPUs are a subset of what the
    process had, and the library still works from there."""
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from gvamp_amd import capi, synth\n"
            "before = os.sched_getaffinity(0)\n"
            "node = capi.bind_host_numa(0)\n"
            "after = os.sched_getaffinity(0)\n"
            "assert after and after <= before, (before, after)\n"
            "assert node >= -1 and (node >= 0 or after == before)\n"
            "os.environ['GVAMP_NUMA_BIND'] = '0'\n"
            "assert capi.bind_host_numa(0) == -1\n"
            "with capi.Shard(300, 64) as sh:\n"
            "    sh.upload_bed(synth.synth_bed(300, 64, seed=1)); sh.compute_markers_statistics()\n"
            "    assert np.isfinite(sh.Ax(np.ones(64))).all()\n"
            "print('node', node, len(before), len(after))\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-600:]
