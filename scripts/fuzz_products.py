"""Randomised shapes through every product of the fixed-point family (development; run on a GPU box):
   python scripts/fuzz_products.py [cases] [seed]
Per case: random N, M around the tile / block boundaries (1, 4, 63..65, 255..257, 1023..1025, ...), missing rate, NA
phenotypes, monomorphic / all-missing markers, vector magnitudes over 60 decades, a random work decomposition (uniform /
tapered / balanced, wave priority on or off).  Checks: both resident layouts bit-identical in every product; Ax / ATx /
statistics against the oracle (rel. l2 < 1e-12); two-vector passes = their one-vector passes bit for bit; pad and NA rows of
Ax exactly zero; one CG solve (both layouts bit-identical, trace vs oracle 1e-9); a pinned decomposition with xcd_skew / wgs_per_cu
(same bits); kernel mode 2 (two-level fixed point: both layouts bit-identical, oracle 1e-12, zeros at pad / NA rows)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi, synth
from oracle import gvoracle as oracle

oracle.lib()
seed0 = 1
BIG = False
NT = min(32, os.cpu_count() or 1)     # oracle threads
EDGE_N = [1, 2, 3, 4, 5, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1023, 1024, 1025, 4095, 4096, 4097]
EDGE_M = [1, 2, 3, 4, 5, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1023, 1024, 1025, 2047, 2049]
ENVK = ("GV_KS_M", "GV_KS_N", "GV_SK_M", "GV_SK_N", "GV_HY_M", "GV_HY_N", "GV_TAPER", "GV_GEO", "GV_PRIO", "GV_TUNE_CACHE")


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def pick(rng, edges, hi):
    r = rng.random()
    if r < 0.5:
        return int(rng.choice(edges))
    if r < 0.9:
        return int(rng.integers(1, hi))
    return int(rng.integers(hi, 6 * hi))


def run_case(k):
    rng = np.random.default_rng(seed0 * 100003 + k)
    if BIG:     # mid-size shards: several quads of row groups, hundreds of K-blocks, the tuner's own picks among the random ones
        N, M = int(rng.integers(3000, 90000)), int(rng.integers(3000, 90000))
        if N * M > 1_200_000_000:
            M = max(1, 1_200_000_000 // N)
    else:
        N, M = pick(rng, EDGE_N, 3000), pick(rng, EDGE_M, 3000)
        if N * M > 30_000_000:
            M = max(1, 30_000_000 // N)
    miss = int(rng.choice([0, 0, 2000, 50000, 300000]))
    fna = float(rng.choice([0.0, 0.0, 0.01, 0.3]))
    bseed = int(rng.integers(1 << 30))
    if BIG:     # generated on the device (the numpy twin of the recipe takes minutes at this size), fetched for the oracle
        with capi.Shard(N, M) as g:
            g.set_layout(True, False)
            g.synth_bed(bseed, miss)
            bed = g.download_bed().copy()
    else:
        bed = synth.synth_bed(N, M, seed=bseed, miss_ppm=miss).copy()
    mb = (N + 3) // 4
    b2 = bed.reshape(M, mb)
    if M >= 3 and rng.random() < 0.5:               # a monomorphic marker (all genotype 2 -> code 00) and an all-missing one
        b2[int(rng.integers(M))] = 0x00
        b2[int(rng.integers(M))] = 0x55
    present = rng.random(N) >= fna
    if N >= 2 and not present.any():
        present[0] = True
    m4, nonas = None, N
    if fna > 0 or N % 4:
        m4 = np.zeros(mb, dtype=np.uint8)
        for n in np.nonzero(present)[0]:
            m4[n >> 2] |= 1 << (n & 3)
        nonas = int(present.sum())
    n4 = 4 * mb
    mag = 10.0 ** rng.integers(-30, 30, size=4)
    x, x2 = rng.standard_normal(M) * mag[0], rng.standard_normal(M) * mag[1]
    if M > 2 and rng.random() < 0.3:
        x[rng.integers(M)] = 0.0
        x2[:] = 0.0                                     # an all-zero operand
    p, p2 = np.zeros(n4), np.zeros(n4)
    p[:N] = rng.standard_normal(N) * present * mag[2]
    p2[:N] = rng.standard_normal(N) * present * mag[3]
    env = {}
    mode = rng.integers(5)
    if mode == 4:       # hybrid: some quads whole (clamped to what the shape has), the rest in balanced ranges
        env = {"GV_HY_M": "%d:%d" % (rng.integers(1, 6), rng.choice([1, 7, 97, 768])),
               "GV_HY_N": "%d:%d" % (rng.integers(1, 6), rng.choice([1, 7, 97, 768])), "GV_PRIO": str(int(rng.integers(2)))}
    elif mode == 1:
        env = {"GV_KS_M": str(int(rng.integers(1, 6))), "GV_KS_N": str(int(rng.integers(1, 6))),
               "GV_TAPER": str(float(rng.choice([0.0, 0.5, 0.9]))), "GV_GEO": str(float(rng.choice([0.0, 0.0, 0.5, 0.8]))),
               "GV_PRIO": str(int(rng.integers(2)))}
    elif mode == 2:
        env = {"GV_SK_M": str(int(rng.choice([1, 7, 97, 768, 1536]))), "GV_SK_N": str(int(rng.choice([1, 7, 97, 768, 1536])))}
    for kk in ENVK:
        os.environ.pop(kk, None)
    os.environ.update(env)
    os.environ["GV_TUNE_CACHE"] = "0"
    out = {}
    for layout in (1, 2):
        with capi.Shard(N, M) as sh:
            sh.set_layout(False, layout)
            sh.set_kernel_mode(1)
            sh.upload_bed(bed)
            if m4 is not None:
                sh.set_mask(m4, nonas)
            sh.compute_markers_statistics()
            mave, msig = sh.marker_stats()
            z, w = sh.Ax(x), sh.ATx(p)
            xa, xb, za, zb = sh.vecM(x), sh.vecM(x2), sh.vecN(), sh.vecN()
            sh.ax2_dev(xa, xb, za, zb)
            pa, pb, wa, wb = sh.vecN(p), sh.vecN(p2), sh.vecM(), sh.vecM()
            sh.atx2_dev(pa, pb, wa, wb)
            z2, w2 = sh.Ax(x2), sh.ATx(p2)
            lm = sh.vecM()
            sh.lmmse_mult(xa, 1.3, 0.4, lm)
            v = sh.vecM(rng.standard_normal(M) if layout == 1 else out[1]["v"])
            vh = v.download()
            mu = sh.vecM()
            cgst, rr = sh.cg_solve(v, None, 1.7, 0.9, 1, 12, mu)
            # round 6: a pinned decomposition with the new fields (longer segments for one block-index parity, two workgroups per CU):
            # the same bits
            for cls in ("ax", "atx", "ax2", "atx2"):
                try:
                    sh.set_decomp(cls, ks=int(rng.integers(2, 5)), geo=float(rng.choice([0.0, 0.5])), prio=int(rng.integers(2)),
                                  wgs_per_cu=int(rng.choice([0, 2])), xcd_skew=float(rng.choice([-0.1, 0.02, 0.2])))
                except capi.GvError:
                    pass                          # fewer K-blocks than segments: the class keeps its decomposition
            assert np.array_equal(sh.Ax(x), z) and np.array_equal(sh.ATx(p), w), "pinned decomposition (xcd_skew / wgs_per_cu) changed a bit"
            # round 6: kernel mode 2 (two-level fixed point) on the same shard
            sh.set_kernel_mode(2)
            zw, ww = sh.Ax(x), sh.ATx(p)
            out[layout] = dict(mave=mave, msig=msig, z=z, w=w, z2=z2, w2=w2, za=za.download(), zb=zb.download(),
                               wa=wa.download(), wb=wb.download(), lm=lm.download(), v=vh, mu=mu.download(), rr=rr, zw=zw, ww=ww)
    a, b = out[1], out[2]
    for key in ("mave", "msig", "z", "w", "z2", "w2", "za", "zb", "wa", "wb", "lm", "mu", "rr", "zw", "ww"):
        assert np.array_equal(a[key], b[key], equal_nan=True), ("layouts differ", key)
    assert np.array_equal(a["z"], a["za"]) and np.array_equal(a["z2"], a["zb"]), "two-vector Ax != one-vector Ax"
    assert np.array_equal(a["w"], a["wa"]) and np.array_equal(a["w2"], a["wb"]), "two-vector ATx != one-vector ATx"
    o_mave, o_msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=nonas, nthreads=NT)
    ok = np.isfinite(o_msig)
    assert np.allclose(a["mave"], o_mave, rtol=1e-13, atol=1e-15, equal_nan=True), "mave"
    assert np.array_equal(np.isfinite(a["msig"]), ok), "msig finiteness"
    assert np.allclose(a["msig"][ok], o_msig[ok], rtol=1e-12), "msig"
    info = (N, M, miss, fna, env)
    if ok.all() and np.isfinite(o_mave).all() and nonas >= 2:
        oz, ow = oracle.ax(bed, N, M, o_mave, o_msig, x, mask4=m4, nthreads=NT), oracle.atx(bed, N, M, o_mave, o_msig, p, nthreads=NT)
        # the yardstick of a sum is the size of its terms, not of what is left after they cancel (a marker column is centred:
        # with M = 1..3 markers the entries of Ax are differences of nearly equal numbers in fp64 and in fixed point alike)
        sz = max(np.linalg.norm(oz), np.abs(o_msig * x).max() * np.sqrt(M) * 3.0 / np.sqrt(N) * np.sqrt(N))
        sw = max(np.linalg.norm(ow), np.abs(o_msig).max() * np.abs(p).max() * 3.0 * np.sqrt(M))
        ez, ew = np.linalg.norm(a["z"] - oz), np.linalg.norm(a["w"] - ow)
        assert ez < 1e-12 * sz, ("Ax vs oracle", info, ez / max(np.linalg.norm(oz), 1e-300), ez / sz)
        assert ew < 1e-12 * sw, ("ATx vs oracle", info, ew / max(np.linalg.norm(ow), 1e-300), ew / sw)
        assert np.all(a["z"][N:] == 0) and np.all(a["z"][:N][~present] == 0), "pad / NA rows of Ax"
        ezw, eww = np.linalg.norm(a["zw"] - oz), np.linalg.norm(a["ww"] - ow)
        assert ezw < 1e-12 * sz, ("kernel mode 2: Ax vs oracle", info, ezw / sz)
        assert eww < 1e-12 * sw, ("kernel mode 2: ATx vs oracle", info, eww / sw)
        assert np.all(a["zw"][N:] == 0) and np.all(a["zw"][:N][~present] == 0), "kernel mode 2: pad / NA rows of Ax"
    return info


def main(ncases, seed, big=False):
    global seed0, BIG
    seed0, BIG = seed, big
    saved = {k: os.environ.get(k) for k in ENVK}
    t0 = time.time()
    bad = []
    try:
        for k in range(ncases):
            try:
                info = run_case(k)
            except AssertionError as e:
                bad.append((k, str(e)))
                print("CASE %d FAILED: %s" % (k, e), flush=True)
                continue
            if k % 10 == 0 or BIG:
                print("case %d ok %s  (%.0f s)" % (k, info, time.time() - t0), flush=True)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    print("%d cases, %d failed, %.0f s" % (ncases, len(bad), time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
                       len(sys.argv) > 3 and sys.argv[3] == "big") else 0)
