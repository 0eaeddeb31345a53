"""Pin of oracle/ against fixtures generated from the REAL reference by oracle/ref_recipe/make_fixtures.py.

The reference's translation units include Boost headers (data.cpp:17, utilities.cpp:8, vamp.cpp:19, options.cpp:9); this
image has none, and stand-ins are not allowed, so today the recipe reports "reference unbuildable: parity unpinned" and
tests/golden/ref/ does not exist: the comparisons below SKIP with that reason.  The day real Boost headers are present,
`python oracle/ref_recipe/make_fixtures.py` writes tests/golden/ref/ and every comparison here becomes live -- SURVEY 8c's
list: mave / msig, Ax, ATx on three tiny beds, g1 / g1d grids, one updatePrior step, one CG solve with its residual trace,
people statistics, LOO p-values, full sim runs at np = 1 / 2 / 8, main_real (NA phenotypes, XXT denoiser), probit.
(CPU only: the oracle against reference outputs; the HIP path is compared with the oracle in the -m gpu tests.)"""
import lzma
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "ref")
RECIPE = os.path.join(ROOT, "oracle", "ref_recipe")
HAVE = os.path.exists(os.path.join(REF, "PROVENANCE.json"))
unpinned = pytest.mark.skipif(not HAVE, reason="parity unpinned: tests/golden/ref/ absent (reference unbuildable here, no Boost)")


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def _boost_present():
    for d in (os.environ.get("BOOST_ROOT", ""), "/usr/include", "/usr/local/include", "/opt/conda/include"):
        if d and os.path.exists(os.path.join(d, "boost", "math", "distributions", "students_t.hpp")):
            return True
    return False


def test_recipe_is_committed_and_refuses_to_fake_a_build():
    """The generator of every reference-derived fixture is in the repo; without real Boost it builds nothing and says so."""
    for f in ("build_ref.sh", "harness.cpp", "make_fixtures.py"):
        assert os.path.exists(os.path.join(RECIPE, f)), f
    src = open(os.path.join(RECIPE, "harness.cpp")).read()
    for sym in ("compute_people_statistics", "precondCG_solver", "updatePrior", "pvals_calc", "->Ax(", "->ATx("):
        assert sym in src, sym
    if not os.path.exists("/root/reference/vamp.cpp"):
        pytest.skip("no reference tree on this box")
    if _boost_present():
        pytest.skip("Boost present: run oracle/ref_recipe/make_fixtures.py to pin the oracle")
    before = set(os.listdir(os.path.join(ROOT, "oracle", "_ref"))) if os.path.isdir(os.path.join(ROOT, "oracle", "_ref")) else set()
    r = subprocess.run([os.path.join(RECIPE, "build_ref.sh")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 3 and "reference unbuildable: parity unpinned" in r.stdout
    after = set(os.listdir(os.path.join(ROOT, "oracle", "_ref"))) if os.path.isdir(os.path.join(ROOT, "oracle", "_ref")) else set()
    assert after == before                                            # nothing was built
    assert not HAVE or os.path.exists(os.path.join(REF, "PROVENANCE.json"))


BEDS = [("t1", 400, 300, False), ("t2", 400, 300, False), ("t3", 403, 257, True)]


def _load_bed(name):
    return np.frombuffer(lzma.open(os.path.join(REF, name + ".bed.xz")).read(), dtype=np.uint8)[3:]


def _phen_mask(name, N):
    """read_phen (data.cpp:128-192): third token, NA -> mask bit cleared"""
    toks = [ln.split() for ln in lzma.open(os.path.join(REF, name + ".phen.xz"), "rt").read().splitlines()]
    is_na = np.array([t[2] == "NA" for t in toks[:N]])
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(~is_na)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    return m4, int((~is_na).sum())


@unpinned
@pytest.mark.parametrize("name,N,M,has_phen", BEDS)
def test_stats_ax_atx_vs_reference(oracle, name, N, M, has_phen):
    bed = _load_bed(name)
    m4, nonas = _phen_mask(name, N) if has_phen else (None, N)
    f = lambda k: np.fromfile(os.path.join(REF, "%s_%s.bin" % (name, k)))      # noqa: E731
    mave, msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    assert np.allclose(mave, f("mave"), rtol=1e-13, atol=1e-15) and np.allclose(msig, f("msig"), rtol=1e-12)
    assert rel(oracle.ax(bed, N, M, mave, msig, f("x"), mask4=m4), f("Ax")) < 1e-13
    assert rel(oracle.atx(bed, N, M, mave, msig, f("p")), f("ATx")) < 1e-13
    pm, ps, pn = oracle.people_stats(bed, N, M, mask4=m4, nonas=nonas)
    assert rel(pm[:N], f("people_mave")[:N]) < 1e-12 and rel(ps[:N], f("people_msig")[:N]) < 1e-12
    assert np.array_equal(pn[:N], f("people_numb")[:N])


@unpinned
@pytest.mark.parametrize("name,N,M,has_phen", BEDS)
def test_prior_cg_pvals_vs_reference(oracle, name, N, M, has_phen):
    bed = _load_bed(name)
    m4, nonas = _phen_mask(name, N) if has_phen else (None, N)
    f = lambda k: np.fromfile(os.path.join(REF, "%s_%s.bin" % (name, k)))      # noqa: E731
    pin = np.loadtxt(os.path.join(REF, name + "_prior_in.txt"), skiprows=1)
    gam1 = float(open(os.path.join(REF, name + "_prior_in.txt")).readline())
    pout = np.loadtxt(os.path.join(REF, name + "_prior_out.txt"))
    probs, vars_ = oracle.update_prior(f("prior_r1"), M, gam1, pin[:, 0], pin[:, 1])
    assert len(probs) == len(pout) and np.allclose(probs, pout[:, 0], rtol=1e-11) and np.allclose(vars_, pout[:, 1], rtol=1e-11)
    if m4 is None:
        mu, rr = oracle.cg_solve(bed, N, M, f("cg_v"), None, 2.0, 1.35, 1, 60)
        assert rel(mu, f("cg_mu")) < 1e-11
        trace = [float(x) for x in re.findall(r"\|\|r_it\|\| / \|\|RHS\|\| = ([0-9.e+-]+),", open(os.path.join(REF, name + "_cg_trace.txt")).read())]
        if trace:
            assert len(trace) == len(rr) and np.allclose(rr, trace, rtol=1e-8)
    g = np.loadtxt(os.path.join(REF, name + "_g1_grid.txt"))
    for gam in np.unique(g[:, 0]):
        sel = g[:, 0] == gam
        o1, o1d = oracle.g1_g1d(g[sel, 1], gam, [0.9, 0.07, 0.03], [0, 2.0, 20.0])
        assert np.array_equal(o1, g[sel, 2]) and np.array_equal(o1d, g[sel, 3])


@unpinned
@pytest.mark.parametrize("np_", [1, 2, 8])
def test_sim_runs_vs_reference(oracle, np_):
    raw = _load_bed("toy")
    N, Mt = 2000, 10000
    beta = np.fromfile(os.path.join(REF, "sim_beta_true.bin"))
    b2, y = oracle.sim_phen(raw, N, Mt, 0.5, 500, 7, nthreads=4)
    assert np.array_equal(b2, beta)
    ref = oracle.infere(raw, N, Mt, y, [0.90, 0.07, 0.03], [0, 0.001, 0.01], nshards=np_, iterations=3, CG_max_iter=20,
                        rho=0.5, seed=7, gam1=1e-8, gamw=2.0, true_signal=beta, nthreads=4)
    pre = os.path.join(REF, "sim_np%d_" % np_)
    tol = 1e-11       # the reference itself moves 2e-14 between its MANVECT and scalar builds
    assert rel(ref.x2[0], np.fromfile(pre + "it_1_x2_hat.bin")) < tol
    for k in (2, 3):
        assert rel(ref.x1[k - 1], np.fromfile(pre + "it_%d.bin" % k)) < tol
        assert rel(ref.x2[k - 1], np.fromfile(pre + "it_%d_x2_hat.bin" % k)) < tol
        assert rel(ref.r1[k - 1], np.fromfile(pre + "r1_it_%d.bin" % k)) < tol
    assert np.allclose([t["gam1_denoise"] for t in ref.trace], np.loadtxt(pre + "gam1s.csv"), rtol=2e-5)   # csv: 6 digits
    assert np.allclose([t["gam2_reest"] for t in ref.trace], np.loadtxt(pre + "gam2s.csv"), rtol=2e-5)
    log = open(pre + "run.log").read()
    ref_rr = [float(x) for x in re.findall(r"\[CG\] it = \d+: \|\|r_it\|\| / \|\|RHS\|\| = ([0-9.e+-]+),", log)]
    if np_ == 1 and ref_rr:
        mine = np.concatenate(ref.relres)
        assert len(mine) == len(ref_rr) and np.allclose(mine, ref_rr, rtol=1e-8)
