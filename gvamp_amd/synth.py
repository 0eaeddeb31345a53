"""Host twin of gv_synth_bed (csrc/gv_kernels.hip:k_synth_bed): the seeded synthetic .bed recipe of SURVEY 8d
in integer-only arithmetic, so the device generator and this numpy one produce identical bytes.  Input
generation only -- no part of the hot path."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def synth_bed(N, M, seed, miss_ppm=5000, S=0, ld_block=0, ld_ppm=0):
    """Returns M * ceil(N/4) bytes, marker-major PLINK 2-bit (no magic bytes), for global markers S..S+M.
    ld_block > 0: gv_synth_bed_ld (block-correlated columns)."""
    mbytes = (N + 3) // 4
    miss_thr = np.uint64((miss_ppm << 32) // 1000000)
    ld_thr = np.uint64(min((ld_ppm << 32) // 1000000, 0xFFFFFFFF))
    with np.errstate(over="ignore"):
        g = np.arange(S, S + M, dtype=np.uint64)
        hm = _splitmix64(np.uint64(seed) ^ (g * np.uint64(0xD1342543DE82EF95)))
        maf = np.uint64(3277) + hm % np.uint64(29491)
        qv = np.uint64(65536) - maf
        p0 = (qv * qv) & np.uint64(0xFFFFFFFF)
        p1 = (np.uint64(2) * maf * qv) & np.uint64(0xFFFFFFFF)
        base = _splitmix64(hm + np.uint64(0x632BE59BD9B4E019))
        if ld_block:
            lbase = _splitmix64(np.uint64(seed) ^ ((g // np.uint64(ld_block)) * np.uint64(0xA24BAED4963EE407)) ^
                                np.uint64(0x5851F42D4C957F2D))
        n = np.arange(N, dtype=np.uint64)
        out = np.zeros((M, mbytes * 4), dtype=np.uint8)
        step = max(1, (1 << 22) // max(N, 1))
        for m0 in range(0, M, step):
            m1 = min(M, m0 + step)
            r = _splitmix64(base[m0:m1, None] + n[None, :])
            u = r >> np.uint64(32)
            um = r & np.uint64(0xFFFFFFFF)
            if ld_block:
                rs = _splitmix64(r ^ np.uint64(0x9FB21C651E98DF25))
                lat = _splitmix64(lbase[m0:m1, None] + n[None, :]) >> np.uint64(32)
                u = np.where((rs >> np.uint64(32)) < ld_thr, lat, u)
            P0 = p0[m0:m1, None]
            P1 = p1[m0:m1, None]
            code = np.zeros(r.shape, dtype=np.uint8)             # geno 2 -> 00
            code[((u - P0) & np.uint64(0xFFFFFFFF)) < P1] = 2      # geno 1 -> 10
            code[u < P0] = 3                                       # geno 0 -> 11
            code[um < miss_thr] = 1                                # missing -> 01
            out[m0:m1, :N] = code
    c = out.reshape(M, mbytes, 4)
    packed = c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)
    return packed.astype(np.uint8).reshape(-1)


def write_bed(path, bed_bytes):
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01]))
        f.write(np.ascontiguousarray(bed_bytes, dtype=np.uint8).tobytes())
