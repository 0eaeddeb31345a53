"""Randomised p-value runs (development; run on a GPU box):   python scripts/fuzz_pvals.py [cases] [seed]
data::pvals_calc / pvals_calc_LOCO (data.cpp:1108-1353) on random shards: shapes around tile / block boundaries, missing
genotypes, NA phenotypes, monomorphic markers, random chromosome labels (sorted or not, chromosomes without markers), effect
sizes from none to strong; the fixed-point family on both resident layouts and the fp64 family, against the oracle
(rtol 1e-7; p-values below 1e-290 compared as underflows), NaN pattern included."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi, synth
from oracle import gvoracle as oracle

oracle.lib()
EDGE_N = [5, 63, 64, 65, 255, 256, 257, 1023, 1025]
EDGE_M = [1, 2, 63, 64, 65, 127, 129, 255, 257, 1025]


def pick(rng, edges, lo, hi):
    return int(rng.choice(edges)) if rng.random() < 0.4 else int(rng.integers(lo, hi))


def same(a, b, rtol):
    a, b = np.asarray(a), np.asarray(b)
    fa, fb = np.isfinite(a), np.isfinite(b)
    if not np.array_equal(fa, fb):
        return False
    tiny = (np.abs(b) < 1e-290) & fb
    ok = fb & ~tiny
    return bool(np.allclose(a[ok], b[ok], rtol=rtol, atol=0) and np.all(np.abs(a[tiny]) < 1e-280))


def run_case(seed0, k):
    rng = np.random.default_rng(seed0 * 100003 + k)
    N, M = pick(rng, EDGE_N, 5, 2500), pick(rng, EDGE_M, 1, 2500)
    miss = int(rng.choice([0, 5000, 100000]))
    fna = float(rng.choice([0.0, 0.02, 0.3]))
    bed = synth.synth_bed(N, M, seed=int(rng.integers(1 << 30)), miss_ppm=miss).copy()
    mb = (N + 3) // 4
    if M >= 3 and rng.random() < 0.4:
        bed.reshape(M, mb)[int(rng.integers(M))] = 0x00          # a monomorphic marker
    present = rng.random(N) >= fna
    if present.sum() < 3:
        present[:3] = True
    m4 = np.zeros(mb, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    nonas = int(present.sum())
    x1 = rng.standard_normal(M) * (rng.random(M) < rng.choice([0.0, 0.05, 0.5])) * float(10.0 ** rng.uniform(-2, 1))
    chrom = rng.integers(1, int(rng.integers(2, 24)), M).astype(np.int32)
    if rng.random() < 0.7:
        chrom = np.sort(chrom)
    mave, msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas)
    info = dict(N=N, M=M, miss=miss, fna=fna)
    if not (np.all(np.isfinite(mave)) and np.all(np.isfinite(msig))):
        return dict(info, skipped="non-finite marker statistics")
    z1 = oracle.ax(bed, N, M, mave, msig, x1, mask4=m4)
    y = np.zeros(4 * mb)
    y[:N] = (z1[:N] + rng.standard_normal(N) * float(10.0 ** rng.uniform(-1, 1))) * present
    o_loo = oracle.pvals(bed, N, M, z1, y, x1, mask4=m4, nonas=nonas)
    o_loco = oracle.pvals(bed, N, M, z1, y, x1, chrom=chrom, mask4=m4, nonas=nonas)
    got = {}
    for mode, layout in ((1, 1), (1, 2), (0, 0)):
        with capi.Shard(N, M) as sh:
            if mode == 1:
                sh.set_layout(False, layout)
            else:
                sh.set_layout(True, False)       # the fp64 family reads the raw rows (not resident by default)
            sh.set_kernel_mode(mode)
            sh.upload_bed(bed)
            sh.set_mask(m4, nonas)
            sh.compute_markers_statistics()
            dz, dy, dx = sh.vecN(z1), sh.vecN(y), sh.vecM(x1)
            got[(mode, layout)] = (sh.pvals_calc(dz, dy, dx), sh.pvals_calc(dz, dy, dx, chrom=chrom))
    for key, (loo, loco) in got.items():
        assert same(loo, o_loo, 1e-7), ("LOO", key, info, np.nanmax(np.abs(loo / o_loo - 1)))
        assert same(loco, o_loco, 1e-7), ("LOCO", key, info, np.nanmax(np.abs(loco / o_loco - 1)))
    assert np.array_equal(got[(1, 1)][0], got[(1, 2)][0], equal_nan=True) and \
        np.array_equal(got[(1, 1)][1], got[(1, 2)][1], equal_nan=True), ("layouts differ", info)
    return info


def main(ncases, seed):
    t0 = time.time()
    bad = []
    for k in range(ncases):
        try:
            info = run_case(seed, k)
        except AssertionError as e:
            bad.append((k, str(e)))
            print("CASE %d FAILED: %s" % (k, e), flush=True)
            continue
        if k % 10 == 0:
            print("case %d ok %s  (%.0f s)" % (k, info, time.time() - t0), flush=True)
    print("%d cases, %d failed, %.0f s" % (ncases, len(bad), time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 60, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)
