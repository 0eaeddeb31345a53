"""Random mid-size shards (up to ~10 GB): the i8 MFMA family against the fp64 VALU family of the same library, the adjoint
identity across the two stripe layouts, and the two-vector passes -- shapes the CPU oracle is too slow for.

  python scripts/fuzz_midsize.py [n_cases] [seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from gvamp_amd import capi


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for i in range(n):
        N = int(rng.integers(1000, 250000))
        M = int(rng.integers(1000, 40_000_000_000 // max(N, 1) // 4 // 3 + 1000))     # <= ~3.3 GB per layout
        M = min(M, 400000)
        miss = int(rng.choice([0, 5000, 100000]))
        fna = float(rng.choice([0.0, 0.01]))
        S = int(rng.choice([0, 12345]))
        try:
            present = rng.random(N) >= fna
            mb = (N + 3) // 4
            m4 = np.zeros(mb, dtype=np.uint8)
            idx = np.nonzero(present)[0]
            np.bitwise_or.at(m4, idx >> 2, (1 << (idx & 3)).astype(np.uint8))
            with capi.Shard(N, M, Mt=S + M, S=S, anchor=True) as sh:
                sh.synth_bed(int(rng.integers(1, 10**6)), miss)
                if fna > 0 or N % 4:
                    sh.set_mask(m4, int(present.sum()))
                sh.compute_markers_statistics()
                x, x2 = rng.standard_normal(M), rng.standard_normal(M)
                p = np.zeros(4 * mb)
                p[:N] = rng.standard_normal(N) * present
                z0, w0 = sh.Ax(x), sh.ATx(p)
                sh.set_kernel_mode(1)
                sh.compute_markers_statistics()
                z1, w1 = sh.Ax(x), sh.ATx(p)
                assert rel(z1, z0) < 1e-12 and rel(w1, w0) < 1e-12, ("families", rel(z1, z0), rel(w1, w0))
                lhs, rhs = float(z1 @ p), float(x @ w1)
                assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs), 1e-300), ("adjoint", lhs, rhs)
                va, vb, oa, ob = sh.vecM(x), sh.vecM(x2), sh.vecN(), sh.vecN()
                sh.ax2_dev(va, vb, oa, ob)
                assert np.array_equal(oa.download(), z1) and np.array_equal(ob.download(), sh.Ax(x2)), "ax2"
                pa, pb, wa, wb = sh.vecN(p), sh.vecN(z1), sh.vecM(), sh.vecM()
                sh.atx2_dev(pa, pb, wa, wb)
                assert np.array_equal(wa.download(), w1) and np.array_equal(wb.download(), sh.ATx(z1)), "atx2"
            print("ok  ", dict(N=N, M=M, miss=miss, fna=fna, S=S), flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("FAIL", dict(N=N, M=M, miss=miss, fna=fna, S=S), repr(e), flush=True)
    print("fuzz (mid-size): %d cases, %d failures" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
