"""Oracle vs outputs of the real reference (tests/golden/survey_probe/README.md -- informational pin)."""
import lzma
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(__file__), "golden", "survey_probe")
N, MT = 2000, 10000
PROBS, VARS = [0.90, 0.07, 0.03], [0, 0.001, 0.01]
TOL = 1e-11   # measured <= 9e-14; the reference itself moves 2e-14 between its MANVECT and scalar builds


@pytest.fixture(scope="module")
def toy_bed():
    raw = np.frombuffer(lzma.open(os.path.join(G, "toy.bed.xz")).read(), dtype=np.uint8)
    assert raw[:3].tolist() == [0x6C, 0x1B, 0x01] and raw.size == 3 + MT * (N // 4)
    return raw[3:].copy()


def rel(a, b):
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def test_g1_g1d_grid(oracle):
    ref = np.loadtxt(os.path.join(G, "g1_grid.txt"))
    for g in np.unique(ref[:, 0]):
        rows = ref[ref[:, 0] == g]
        g1, g1d = oracle.g1_g1d(rows[:, 1], g, PROBS, [0, 2.0, 20.0])
        assert np.allclose(g1, rows[:, 2], rtol=1e-13, atol=0)
        # g1d at gam1=1e-8 cancels catastrophically (vamp.cpp:866); bit-identical only with FMA contraction
        assert np.allclose(g1d, rows[:, 3], rtol=1e-7 if g < 1e-6 else 1e-13, atol=0)


def test_sim_beta_true(oracle, toy_bed):
    beta, _y = oracle.sim_phen(toy_bed, N, MT, 0.5, 500, 7, nthreads=4)
    ref = np.fromfile(os.path.join(G, "sim_beta_true.bin"))
    assert np.count_nonzero(ref) == np.count_nonzero(beta)
    assert np.abs(beta - ref).max() < 1e-15


@pytest.mark.parametrize("nshards", [1, 2, 8])
def test_sim_run_matches_reference(oracle, toy_bed, nshards):
    beta, y = oracle.sim_phen(toy_bed, N, MT, 0.5, 500, 7, nthreads=4)
    r = oracle.infere(toy_bed, N, MT, y, PROBS, VARS, nshards=nshards, iterations=3, CG_max_iter=20, rho=0.5,
                      seed=7, gam1=1e-8, gamw=2.0, true_signal=beta)
    pre = os.path.join(G, "sim_np%d_" % nshards)
    assert rel(r.x2[0], np.fromfile(pre + "it_1_x2_hat.bin")) < TOL
    assert rel(r.x1[2], np.fromfile(pre + "it_3.bin")) < TOL
    assert rel(r.x2[2], np.fromfile(pre + "it_3_x2_hat.bin")) < TOL
    assert rel(r.r1[2], np.fromfile(pre + "r1_it_3.bin")) < TOL
    g1s = np.loadtxt(pre + "gam1s.csv")
    g2s = np.loadtxt(pre + "gam2s.csv")
    assert np.allclose([t["gam1_denoise"] for t in r.trace], g1s, rtol=2e-5)   # csv holds 6 digits
    assert np.allclose([t["gam2_reest"] for t in r.trace], g2s, rtol=2e-5)


def test_sim_np1_cg_trace_matches_log(oracle, toy_bed):
    """CG residual traces and per-iteration scalars printed by the reference (vamp.cpp:1219-1220, :640, :696)."""
    import re
    log = open(os.path.join(G, "sim_np1_run.log")).read()
    ref_rr = [float(x) for x in re.findall(r"\[CG\] it = \d+: \|\|r_it\|\| / \|\|RHS\|\| = ([0-9.e+-]+),", log)]
    ref_a2 = [float(x) for x in re.findall(r"^alpha2 = ([0-9.e+-]+)", log, re.M)]
    ref_gw = [float(x) for x in re.findall(r"^gamw = ([0-9.e+-]+)", log, re.M)][1::2]   # printed twice per iteration
    beta, y = oracle.sim_phen(toy_bed, N, MT, 0.5, 500, 7, nthreads=4)
    r = oracle.infere(toy_bed, N, MT, y, PROBS, VARS, iterations=3, CG_max_iter=20, rho=0.5, seed=7,
                      gam1=1e-8, gamw=2.0, true_signal=beta)
    mine = np.concatenate(r.relres)
    assert len(mine) == len(ref_rr)
    assert np.allclose(mine, ref_rr, rtol=1e-8)            # log prints 10 significant digits
    assert np.allclose([t["alpha2"] for t in r.trace], ref_a2, rtol=1e-8)
    assert np.allclose([t["gamw"] for t in r.trace], ref_gw, rtol=1e-8)


def test_real_na_phenotype_run(oracle, toy_bed):
    """main_real --run-mode infere with NA phenotypes: read_phen scaling + mask4 (data.cpp:128-192), scalar build."""
    raw, na = [], []
    for line in open(os.path.join(G, "toy.phen")):
        t = line.split()
        na.append(t[2] == "NA")
        raw.append(0.0 if t[2] == "NA" else float(t[2]))
    assert sum(na) == 4
    r = oracle.infere(toy_bed, N, MT, np.array(raw), PROBS, VARS, iterations=3, CG_max_iter=20, rho=0.5, seed=7,
                      gam1=1e-6, gamw=2.0, is_na=np.array(na, dtype=np.uint8))
    pre = os.path.join(G, "real_")
    assert rel(r.x2[0], np.fromfile(pre + "it_1_x2_hat.bin")) < TOL
    assert rel(r.x1[2], np.fromfile(pre + "it_3.bin")) < TOL
    assert rel(r.x2[2], np.fromfile(pre + "it_3_x2_hat.bin")) < TOL
    assert rel(r.r1[2], np.fromfile(pre + "r1_it_3.bin")) < TOL
