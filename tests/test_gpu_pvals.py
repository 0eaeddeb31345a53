"""Association tests after the VAMP loop (vamp.cpp:761-776): data::pvals_calc / pvals_calc_LOCO (data.cpp:1108-1353)
of the product against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

from gvamp_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", [0, 1])
def test_pvals_loo_and_loco_vs_oracle(oracle, mode):
    """NA phenotypes, missing genotypes, an empty chromosome; both kernel families."""
    N, M = 1203, 900
    rng = np.random.default_rng(21)
    bed = synth.synth_bed(N, M, seed=55, miss_ppm=15000)
    present = rng.random(N) >= 0.02
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(present.sum())
    x1 = rng.standard_normal(M) * (rng.random(M) < 0.05) * 3.0
    chrom = np.sort(rng.integers(1, 24, M)).astype(np.int32)
    chrom[chrom == 7] = 8
    mave, msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    z1 = oracle.ax(bed, N, M, mave, msig, x1, mask4=m4)
    y = np.zeros(4 * ((N + 3) // 4))
    y[:N] = (z1[:N] + rng.standard_normal(N)) * present
    o_loo = oracle.pvals(bed, N, M, z1, y, x1, mask4=m4, nonas=nonas, nthreads=4)
    o_loco = oracle.pvals(bed, N, M, z1, y, x1, chrom=chrom, mask4=m4, nonas=nonas, nthreads=4)
    with capi.Shard(N, M, anchor=(mode == 0)) as sh:
        sh.upload_bed(bed)
        sh.set_mask(m4, nonas)
        sh.set_kernel_mode(mode)
        sh.compute_markers_statistics()
        dz, dy, dx = sh.vecN(z1), sh.vecN(y), sh.vecM(x1)
        loo = sh.pvals_calc(dz, dy, dx)
        loco = sh.pvals_calc(dz, dy, dx, chrom=chrom)
    assert np.all((loo >= 0) & (loo <= 1)) and loo.min() < 1e-3      # true effects are detected
    assert np.allclose(loo, o_loo, rtol=1e-8, atol=0)
    assert np.allclose(loco, o_loco, rtol=1e-8, atol=0)


def test_gvamp_sim_store_pvals_files(tmp_path, oracle):
    """--store-pvals 1 with a .bim file: _pvals.bin and _pvals_LOCO.bin of the driver vs the oracle evaluated on the
    driver's own final iterate."""
    N, M = 800, 1200
    bed = synth.synth_bed(N, M, seed=90, miss_ppm=5000)
    bedp = str(tmp_path / "s.bed")
    synth.write_bed(bedp, bed)
    chrom = np.repeat(np.arange(1, 13), 100)
    with open(tmp_path / "s.bim", "w") as f:
        for i, ch in enumerate(chrom):
            f.write("%s\trs%d\t0\t%d\tA\tG\n" % ("X" if ch == 12 else str(ch), i, i + 1))
    chrom = np.where(chrom == 12, 23, chrom).astype(np.int32)        # "X" -> 23 (data.cpp:366-367)
    out = str(tmp_path / "o") + "/"
    cmd = [os.path.join(ROOT, "gvamp_amd", "gvamp_sim"), "--bed-file", bedp, "--bim-file", str(tmp_path / "s.bim"), "--N",
           str(N), "--Mt", str(M), "--out-dir", out, "--out-name", "s", "--iterations", "3", "--probs", "0.9,0.1", "--vars",
           "0,0.01", "--CV", "60", "--h2", "0.5", "--rho", "0.5", "--CG-max-iter", "15", "--seed", "3", "--store-pvals", "1"]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    x1 = np.fromfile(out + "s_it_3.bin") * np.sqrt(N)
    mave, msig = oracle.marker_stats(bed, N, M)
    z1 = oracle.ax(bed, N, M, mave, msig, x1)
    _beta, y_full = oracle.sim_phen(bed, N, M, 0.5, 60, 3)
    assert np.allclose(np.loadtxt(out + "s_y.txt"), y_full, rtol=1e-5)
    ypad = np.zeros(z1.size)
    ypad[:N] = y_full
    assert np.allclose(np.fromfile(out + "s_pvals.bin"), oracle.pvals(bed, N, M, z1, ypad, x1), rtol=1e-7)
    assert np.allclose(np.fromfile(out + "s_pvals_LOCO.bin"), oracle.pvals(bed, N, M, z1, ypad, x1, chrom=chrom), rtol=1e-7)
    # the per-chromosome predictors the reference dumps next to them (data.cpp:1276-1281): A x1_hat restricted to a
    # chromosome, 4*mbytes values with the default 6 significant digits; zeros for chromosomes without markers
    for ch in (1, 7, 23, 15):
        pred = np.loadtxt(out + "s_LOCO_chr_%d.csv" % ch)
        want = oracle.ax(bed, N, M, mave, msig, np.where(chrom == ch, x1, 0.0))
        assert pred.shape == want.shape and np.allclose(pred, want, rtol=2e-5, atol=1e-9), ch
    assert np.all(np.loadtxt(out + "s_LOCO_chr_15.csv") == 0)


@pytest.mark.parametrize("N,M", [(26, 70), (60000, 300)])
def test_student_t_tail_branches_vs_oracle(oracle, N, M):
    """The two evaluations of the Student-t tail in gv_pval_dev.h against the oracle's continued fraction: a sample small enough
    for the Lentz fraction (nu / 2 < 15) and one deep in the large-sample expansion (nu / 2 = 3e4: three terms), with effects strong
    enough for p-values down to 1e-100 and below -- the relative error has to hold in the far tail, not only near 1."""
    rng = np.random.default_rng(N)
    bed = synth.synth_bed(N, M, seed=77, miss_ppm=8000)
    x1 = np.zeros(M)
    x1[:: max(1, M // 12)] = np.linspace(0.5, 30.0 if N > 1000 else 2.0, len(x1[:: max(1, M // 12)]))
    mave, msig = oracle.marker_stats(bed, N, M)
    z1 = oracle.ax(bed, N, M, mave, msig, x1)
    y = np.zeros(4 * ((N + 3) // 4))
    y[:N] = z1[:N] + rng.standard_normal(N)
    ref = oracle.pvals(bed, N, M, z1, y, x1, nthreads=4)
    with capi.Shard(N, M) as sh:
        sh.upload_bed(bed)
        sh.compute_markers_statistics()
        got = sh.pvals_calc(sh.vecN(z1), sh.vecN(y), sh.vecM(x1))
    assert np.all(np.isfinite(got)) and np.allclose(got, ref, rtol=1e-8, atol=0), np.max(np.abs(got / np.where(ref > 0, ref, 1) - 1))
    if N > 1000:
        assert 0 < ref.min() < 1e-100      # (the strongest effect stays clear of the underflow threshold)
