// gv_mfma.hip -- fixed-point i8 MFMA family for data::Ax / data::ATx on gfx950 (kernel mode 1).
//
// Why: one .bed byte = 4 genotypes = 4 fp64 FMAs; at HBM rate that alone is ~80 % of the fp64 VALU peak before
// any decode, and the measured fp64 VALU kernels reach 4-9 % of HBM peak.  The matvec against a 2-bit
// matrix is exact in integers once the vector is put in fixed point, and CDNA4 has an i8 matrix pipe:
//
//   v (fp64)  ->  q = rint(v * 2^(54-e)),  2^(e-1) <= max|v| < 2^e      (absolute error <= 2^-55 max|v| per entry)
//             ->  7 balanced base-256 digits d_l in [-128,127]          (q = sum_l d_l 256^l, exact)
//   per 16x64 genotype tile:  acc[row][l] += sum_k g[row][k] * d_l[k]   v_mfma_i32_16x16x64_i8, exact int32
//   result = (sum_l acc_l 256^l) * 2^(e-54)                             exact int64 limbs -> one fp64 rounding
//
// All sums are integer: results do not depend on tile order, K-split count or wave scheduling (bitwise
// reproducible), and the error is that of the input quantisation only -- below fp64 summation error.
//
// HBM layout ("stripes", built once at upload from the PLINK rows): 2-bit codes re-encoded to
//   r' = 2 (a=2), 1 (a=1), 0 (a=0), 3 (missing)         [PLINK 00,10,11,01 ; data.cpp decode via dotp_lut]
// so that sum_k r' v = sum a v + 3 sum miss v needs no table for the first plane.  A supertile = 64 rows x 256
// K-entries = 4 KiB contiguous = 4 tiles of 16 rows; inside a tile lane l = (r = l&15, g = l>>4) owns 16 bytes =
// K-entries [64g, 64g+64) of row r, dword d = entries 64g+16d+q at bits 2q.  One wave-wide 16-byte load is
// therefore one fully coalesced KiB and feeds four MFMAs (d = 0..3) with NO cross-lane movement:
//   A operand of MFMA d, VGPR s, byte t  =  (dword_d >> 2s) & 3 at byte t   <->  K-entry 64g + 16d + 4t + s
// and the vector digits are stored in the same (d, g, s, t) order (k_quant).  Two stripe sets are resident:
//   stripes_m : rows = markers,     K = individuals  (ATx)
//   stripes_n : rows = individuals, K = markers      (Ax; the 2-bit transpose)
#include <cstdlib>

#include "gv_internal.h"
#include "gv_mfma.h"
#include "gv_pval_dev.h"

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t recode(uint32_t w) {   // PLINK code -> r'
    return ((~w) & 0xAAAAAAAAu) | (((w >> 1) ^ w) & 0x55555555u);
}
// keep entries with index < nvalid (0..16) of a 16-entry word
__device__ __forceinline__ uint32_t keep_lo(uint32_t w, int nvalid) {
    return nvalid >= 16 ? w : (nvalid <= 0 ? 0u : (w & ((1u << (2 * nvalid)) - 1u)));
}

// ---- stripes_m from a raw chunk: markers [m0, m0 + mc) of the shard, m0 % 64 == 0 ----------------------------------
// block = supertile (rg_local, kb); thread t = (i = t>>6, l = t&63 = (r, g)) moves one 16-byte piece.
__global__ __launch_bounds__(256) void k_stripes_m(const uint8_t* __restrict__ raw, int64_t pitch, int64_t mc,
                                                   int64_t N, uint4* __restrict__ stripes, int64_t rg0, int64_t nkb) {
    const int t = threadIdx.x, i = t >> 6, l = t & 63, r = l & 15, g = l >> 4;
    const int64_t kb = blockIdx.x, rgl = blockIdx.y;
    const int64_t m = rgl * 64 + 16 * i + r;          // local marker in the chunk
    const int64_t byte0 = kb * 64 + 16 * g;           // first byte of the piece in the row
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < mc && byte0 < pitch) {
        v = *reinterpret_cast<const uint4*>(raw + m * pitch + byte0);
        const int64_t n0 = byte0 * 4;                 // first individual of the piece
        v.x = keep_lo(recode(v.x), (int)min((int64_t)16, max((int64_t)0, N - n0)));
        v.y = keep_lo(recode(v.y), (int)min((int64_t)16, max((int64_t)0, N - n0 - 16)));
        v.z = keep_lo(recode(v.z), (int)min((int64_t)16, max((int64_t)0, N - n0 - 32)));
        v.w = keep_lo(recode(v.w), (int)min((int64_t)16, max((int64_t)0, N - n0 - 48)));
    }
    stripes[((rg0 + rgl) * nkb + kb) * 256 + t] = v;
}

// ---- stripes_n (2-bit transpose) from a raw chunk: markers [m0, m0 + mc), m0 % 256 == 0 ---------------------------
// block = (kb_local: 256 markers, ng: 64 individuals).  Stage the 256 x 16-byte block in LDS, then every thread
// gathers the 64 markers of one (individual, g) into its 16-byte piece.
__global__ __launch_bounds__(256) void k_stripes_n(const uint8_t* __restrict__ raw, int64_t pitch, int64_t mc,
                                                   int64_t N, uint4* __restrict__ stripes, int64_t kb0, int64_t nkb) {
    __shared__ uint32_t sh[256][5];   // +1 pad: the gather reads column i of 64 consecutive rows
    const int t = threadIdx.x;
    const int64_t kbl = blockIdx.x, ng = blockIdx.y;
    const int64_t m = kbl * 256 + t, byte0 = ng * 16;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (m < mc && byte0 < pitch) {
        v = *reinterpret_cast<const uint4*>(raw + m * pitch + byte0);
        const int64_t n0 = ng * 64;
        v.x = keep_lo(recode(v.x), (int)min((int64_t)16, max((int64_t)0, N - n0)));
        v.y = keep_lo(recode(v.y), (int)min((int64_t)16, max((int64_t)0, N - n0 - 16)));
        v.z = keep_lo(recode(v.z), (int)min((int64_t)16, max((int64_t)0, N - n0 - 32)));
        v.w = keep_lo(recode(v.w), (int)min((int64_t)16, max((int64_t)0, N - n0 - 48)));
    }
    sh[t][0] = v.x; sh[t][1] = v.y; sh[t][2] = v.z; sh[t][3] = v.w;
    __syncthreads();
    const int i = t >> 6, l = t & 63, r = l & 15, g = l >> 4;
    uint32_t o[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        uint32_t w = 0;
#pragma unroll
        for (int q = 0; q < 16; q++) w |= ((sh[64 * g + 16 * d + q][i] >> (2 * r)) & 3u) << (2 * q);
        o[d] = w;
    }
    stripes[(ng * nkb + kb0 + kbl) * 256 + t] = make_uint4(o[0], o[1], o[2], o[3]);
}

// ---- compute_markers_statistics (data.cpp:451-484) from stripes_m: one wave per row group of 64 markers -------------
// A wave reads the whole 4 KiB supertile of its row group per K-block -- four 1 KiB tiles of 16 markers, the access pattern of the
// matvec kernels -- and keeps the three genotype counts of its four tiles.  (One wave per TILE, the first form of this kernel, had
// the four waves of a workgroup pick the 1 KiB pieces of a supertile at different times: 4.5-5.0 TB/s where the matvec kernels
// stream at 6.6.)
template <int UNR>
__global__ __launch_bounds__(256) void k_stats_stripes(const uint4* __restrict__ stripes, const uint32_t* __restrict__ mask2,
                                                       int64_t M, int64_t nkb, int64_t P4, double nonas,
                                                       double alpha_scale, double* __restrict__ mave,
                                                       double* __restrict__ msig, uint32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);     // row group of 64 markers
    if (rg * 64 >= M) return;
    // per tile: nh = codes with the high bit set (a = 2 or missing), nl = with the low bit set (a = 1 or missing), nb = both
    // (missing), all among the individuals with a phenotype; np = those individuals in this lane's K slices.  Then n2 = nh - nb,
    // n1 = nl - nb, n0 = np - nh - nl + nb: nine vector instructions per dword instead of fourteen for the three masked counts
    uint32_t nh[4] = {0, 0, 0, 0}, nl[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0}, np = 0;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    struct Tiles { v4u t[4]; };
    auto load = [&](Tiles& a, const v4u* p) {
#pragma unroll
        for (int i = 0; i < 4; i++) a.t[i] = __builtin_nontemporal_load(p + i * 64);
    };
    auto count = [&](const Tiles& a, int64_t kb) {
        const int64_t w0 = kb * 16 + g * 4;   // mask2 word of entries 256kb + 64g
        uint32_t pm[4];                       // both bits of a pair are set for an individual with a phenotype
#pragma unroll
        for (int d = 0; d < 4; d++) {
            pm[d] = (w0 + d < P4) ? mask2[w0 + d] : 0u;
            np += __popc(pm[d] & 0x55555555u);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t ws[4] = {a.t[i].x, a.t[i].y, a.t[i].z, a.t[i].w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t wm = ws[d] & pm[d];
                nh[i] += __popc(wm & 0xAAAAAAAAu);
                nl[i] += __popc(wm & 0x55555555u);
                nb[i] += __popc(wm & (wm >> 1) & 0x55555555u);
            }
        }
    };
    // UNR K-blocks per trip: UNR x 4 KiB of independent non-temporal loads in flight per wave
    const v4u* src = reinterpret_cast<const v4u*>(stripes) + rg * nkb * 256 + lane;
    int64_t kb = 0;
    for (; kb + UNR <= nkb; kb += UNR) {
        Tiles a[UNR];
#pragma unroll
        for (int j = 0; j < UNR; j++) load(a[j], src + (kb + j) * 256);
#pragma unroll
        for (int j = 0; j < UNR; j++) count(a[j], kb + j);
    }
    for (; kb < nkb; kb++) {
        Tiles a;
        load(a, src + kb * 256);
        count(a, kb);
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // sum over the 4 lanes (g) that share row r of tile i
        uint32_t ch = nh[i], cl = nl[i], cb = nb[i], cp = np;
        ch += __shfl_xor(ch, 16, 64); ch += __shfl_xor(ch, 32, 64);
        cl += __shfl_xor(cl, 16, 64); cl += __shfl_xor(cl, 32, 64);
        cb += __shfl_xor(cb, 16, 64); cb += __shfl_xor(cb, 32, 64);
        cp += __shfl_xor(cp, 16, 64); cp += __shfl_xor(cp, 32, 64);
        const uint32_t c2 = ch - cb, c1 = cl - cb, c0 = cp - ch - cl + cb;
        const int64_t m = rg * 64 + 16 * i + r;
        if (g == 0 && m < M) {
            double suma = 2.0 * c2 + 1.0 * c1, sumb = (double)c0 + (double)c1 + (double)c2;
            double mu = (sumb != 0) ? suma / sumb : 0.0;
            double sumsqr = c2 * ((2.0 - mu) * (2.0 - mu)) + c1 * ((1.0 - mu) * (1.0 - mu)) + c0 * (mu * mu);
            double sg;
            if (sumsqr != 0) {
                if (alpha_scale == 1.0) sg = 1.0 / sqrt(sumsqr / (nonas - 1.0));
                else sg = 1.0 / pow(sqrt(sumsqr / (nonas - 1.0)), alpha_scale);
            } else
                sg = 1.0;
            mave[m] = mu;
            msig[m] = sg;
            if (counts) {
                counts[3 * m] = c2;
                counts[3 * m + 1] = c1;
                counts[3 * m + 2] = c0;
            }
        }
    }
}

// ---- vector preparation -------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// scal[0] = amax, scal[1] = sum (ordered), scal[2] = 2^(54-e) (quantisation multiplier, 0 if amax is 0 / not finite),
// scal[3] = 2^(e-54).  The prep kernels leave block partials; every block of the quantisation launch behind them adds them up
// for itself in a fixed order (prep_scalars) and block 0 stores the four scalars for the epilogue of the pass.  (Until round 3
// the LAST block of the prep launch did it behind a ticket counter: 256 serialised atomics per vector, 12 of the 17-21 us a
// two-vector k_prep_ax took at M = 200k -- profiles/r3_cfg5_gaps.txt.)
__device__ __forceinline__ void prep_scalars(const double* __restrict__ partial, int nblocks, double* shm, double* shs,
                                              double (&out)[4]) {
    double mx = 0.0, s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) {
        mx = fmax(mx, partial[2 * b]);
        s += partial[2 * b + 1];
    }
    shm[threadIdx.x] = mx;
    shs[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            shm[threadIdx.x] = fmax(shm[threadIdx.x], shm[threadIdx.x + off]);
            shs[threadIdx.x] += shs[threadIdx.x + off];
        }
        __syncthreads();
    }
    const double amax = shm[0];
    out[0] = amax;
    out[1] = shs[0];
    if (amax > 0.0 && amax <= 1.7976931348623157e308) {
        int e = ilogb(amax) + 1;
        out[2] = ldexp(1.0, 54 - e);
        out[3] = ldexp(1.0, e - 54);
    } else if (amax == 0.0) {
        out[2] = 0.0;
        out[3] = 0.0;
    } else {
        // a NaN or an infinity among the entries (the prep kernels raise amax to +inf for either): fixed point has no
        // encoding for it.  Defined behaviour = what fp64 sums over the whole vector give in the reference: every
        // output entry of this product is NaN (digits are quantised with multiplier 0, the epilogue multiplies by NaN).
        out[2] = 0.0;
        out[3] = __longlong_as_double(0x7ff8000000000000LL);
    }
}
// block partials of a prep launch: [2b] = max, [2b + 1] = sum (fixed order within the block)
__device__ __forceinline__ void prep_block_partials(double* partial_v, double mx, double s, double* shm, double* shs) {
    mx = wave_max(mx);
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) { shm[threadIdx.x >> 6] = mx; shs[threadIdx.x >> 6] = s; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial_v[2 * blockIdx.x] = fmax(fmax(shm[0], shm[1]), fmax(shm[2], shm[3]));
        partial_v[2 * blockIdx.x + 1] = shs[0] + shs[1] + shs[2] + shs[3];
    }
}

constexpr int PREP_STRIDE = 2 * RED_BLOCKS;   // doubles of block partials per vector

// Ax operands: c = msig*x, e = (mave-3)*c  (out = sum r' c + sum miss e - K0, K0 = sum mave*c).  blockIdx.y = vector.
// block partials: [0] = max(|c|,|e|), [1] = sum mave*c
// CG hook (device-resident CG): x is the search direction p of system st; when that system took a step and is still
// running, p <- z + beta p (vamp.cpp:1209-1210) happens here, on the way into the operands, instead of in a launch of its own.
struct PrepAx { const double* x[2]; double* cv[2]; double* ev[2]; const double* st[2]; double* pw[2]; const double* z[2];
                const int* ride; const double* alt_x; const int* go; };
__global__ __launch_bounds__(256) void k_prep_ax(PrepAx a, const double* __restrict__ mave, const double* __restrict__ msig,
                                                 int64_t M, double* __restrict__ partial) {
    __shared__ double shm[4], shs[4];
    if (a.go && *a.go == 0) return;      // a pass enqueued after every system had finished: dropped (as its streaming kernel is)
    const int v = blockIdx.y;
    const double* x = a.x[v];
    double* __restrict__ cv = a.cv[v];
    double* __restrict__ ev = a.ev[v];
    const double* st = a.st[v];
    // the rider takes the slot of the one system that has finished (uniform over the launch: nothing changes these flags
    // between the k_cgx_decide of the previous step and the one of this step)
    if (a.ride && *a.ride == 1 && st && st[gvm::ST_ACTIVE] == 0.0 && a.st[1 - v] && a.st[1 - v][gvm::ST_ACTIVE] != 0.0) x = a.alt_x;
    const bool upd = st && a.pw[v] && st[gvm::ST_STEPPED] != 0.0 && st[gvm::ST_ACTIVE] != 0.0;   // (a slot may name a state for the rider's sake only)
    const double beta = upd ? st[gvm::ST_BETA] : 0.0;
    const double* zz = a.z[v];
    double* pw = a.pw[v];
    double mx = 0.0, s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        double xi = x[i];
        if (upd) {
            xi = fma(beta, xi, zz[i]);      // = gvk::p_update of the host-driven loop, bit for bit
            pw[i] = xi;
        }
        double c = msig[i] * xi, mu = mave[i];
        double e = (mu - 3.0) * c;
        cv[i] = c;
        ev[i] = e;
        mx = (isfinite(c) && isfinite(e)) ? fmax(mx, fmax(fabs(c), fabs(e))) : __longlong_as_double(0x7ff0000000000000LL);
        s += mu * c;
    }
    prep_block_partials(partial + v * PREP_STRIDE, mx, s, shm, shs);
}
// Ax operands of the two-level fixed point (kernel mode 2): c = msig * x for the a' plane, e = -(mave * c) for the present plane
// (out = sum a' c + sum b e: a missing genotype contributes an exact 0 to both).  block partials: [0] = max(|c|, |e|), [1] = 0.
__global__ __launch_bounds__(256) void k_prep_ax_wide(const double* __restrict__ x, const double* __restrict__ mave,
                                                      const double* __restrict__ msig, int64_t M, double* __restrict__ cv,
                                                      double* __restrict__ ev, double* __restrict__ partial) {
    __shared__ double shm[4], shs[4];
    double mx = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const double c = msig[i] * x[i];
        const double e = -(mave[i] * c);
        cv[i] = c;
        ev[i] = e;
        mx = (isfinite(c) && isfinite(e)) ? fmax(mx, fmax(fabs(c), fabs(e))) : __longlong_as_double(0x7ff0000000000000LL);
    }
    prep_block_partials(partial, mx, 0.0, shm, shs);
}
// Operands of the three per-individual sums behind data::compute_people_statistics (data.cpp:590-624), in the Ax form
// out = sum_m plane_m c_m + sum_m miss_m e_m - K0 with sigma = msig, mu = mave:
//   kind 0: count of non-missing markers          c = 0,             e = -1,            K0 = -M                  (r' plane)
//   kind 1: sigma^2 a^2 + mu^2 sigma^2 if present   c = sigma^2,       e = -mu^2 sigma^2, K0 = -sum mu^2 sigma^2   (a^2 plane)
//   kind 2: -2 mu sigma^2 a if present              c = -2 mu sigma^2, e = 6 mu sigma^2,  K0 = 0                   (r' plane)
// (r' = a for a present genotype and 3 for a missing one, which contributes 0 to every sum: the miss term takes back
// what the code 3 and K0 put in.)  kind 1 + kind 2 = sum_m b ((a - mu) sigma)^2.
__global__ __launch_bounds__(256) void k_prep_people(int kind, const double* __restrict__ mave, const double* __restrict__ msig,
                                                     int64_t M, double* __restrict__ cv, double* __restrict__ ev,
                                                     double* __restrict__ partial) {
    __shared__ double shm[4], shs[4];
    double mx = 0.0, s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < M; i += (int64_t)gridDim.x * 256) {
        const double mu = mave[i], s2 = msig[i] * msig[i];
        double c, e, k;
        if (kind == 0) {
            c = 0.0; e = -1.0; k = -1.0;
        } else if (kind == 1) {
            // present: sigma^2 a^2 + mu^2 sigma^2 ; missing: plane 0, so the miss term removes mu^2 sigma^2
            c = s2; e = -mu * mu * s2; k = -mu * mu * s2;
        } else {
            // present: -2 mu sigma^2 a ; missing (r' = 3): -6 mu sigma^2 + e = 0
            c = -2.0 * mu * s2; e = 6.0 * mu * s2; k = 0.0;
        }
        cv[i] = c;
        ev[i] = e;
        mx = fmax(mx, fmax(fabs(c), fabs(e)));
        s += k;
    }
    prep_block_partials(partial, mx, s, shm, shs);
}
// ATx operand p: block partials [0] = max|p|, [1] = sum p.  blockIdx.y = vector.
// CG hook (gv_cg_solve_aat2w, N-space system): slot v's operand is the search direction p of system st[v]; when that system
// took a step and is still running, p <- z + beta p (denoiserXXT.cpp:109-110) happens here, on the way into the digits, as
// k_prep_ax does it for the M-space systems (= k_p_update_st, bit for bit).
struct PrepAtx { const double* p[2]; const int* go; const double* st[2]; double* pw[2]; const double* z[2]; };
__global__ __launch_bounds__(256) void k_prep_atx(PrepAtx a, int64_t n, double* __restrict__ partial) {
    __shared__ double shm[4], shs[4];
    if (a.go && *a.go == 0) return;
    const int v = blockIdx.y;
    const double* p = a.p[v];
    const double* st = a.st[v];
    const bool upd = st && a.pw[v] && st[gvm::ST_STEPPED] != 0.0 && st[gvm::ST_ACTIVE] != 0.0;
    const double beta = upd ? st[gvm::ST_BETA] : 0.0;
    const double* zz = a.z[v];
    double* pw = a.pw[v];
    double mx = 0.0, s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        double val = p[i];
        if (upd) {
            val = fma(beta, val, zz[i]);
            pw[i] = val;
        }
        mx = isfinite(val) ? fmax(mx, fabs(val)) : __longlong_as_double(0x7ff0000000000000LL);   // fmax would drop a NaN
        s += val;
    }
    prep_block_partials(partial + v * PREP_STRIDE, mx, s, shm, shs);
}

// The two operands of the p-value pass in one sweep: p = mask (y - z1) + add (add may be NULL; LOCO: the chromosome's predictor,
// data.cpp:1284) and p^2, with the block partials of both (slot 0: p, slot 1: p^2) -- what k_axpby, k_mask_copy, k_mul and
// k_prep_atx did in four launches.  The regression sums are matvec-shaped: they need zeros at NA / pad slots, whatever the caller
// left there (an unfiltered y holds DBL_MAX, data.cpp:147).
__global__ __launch_bounds__(256) void k_prep_pv(const double* __restrict__ y, const double* __restrict__ z1,
                                                 const double* __restrict__ add, const uint32_t* __restrict__ mask2, int64_t n,
                                                 double* __restrict__ pa, double* __restrict__ pb, double* __restrict__ partial) {
    __shared__ double shm[4], shs[4];
    double mx = 0.0, s = 0.0, mx2 = 0.0, s2 = 0.0;
    const double inf = __longlong_as_double(0x7ff0000000000000LL);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t present = (mask2[i >> 4] >> (2 * (i & 15))) & 1u;
        double v = present ? y[i] - z1[i] : 0.0;
        if (add) v = add[i] + v;
        const double v2 = v * v;
        pa[i] = v;
        pb[i] = v2;
        mx = isfinite(v) ? fmax(mx, fabs(v)) : inf;
        mx2 = isfinite(v2) ? fmax(mx2, v2) : inf;
        s += v;
        s2 += v2;
    }
    prep_block_partials(partial, mx, s, shm, shs);
    __syncthreads();
    prep_block_partials(partial + PREP_STRIDE, mx2, s2, shm, shs);
}

// fixed-point digits of v in MFMA B-operand order.  Thread = (kb, g, d, s): 4 entries k = 256kb+64g+16d+4t+s (t=0..3)
// become byte t of one dword per digit c, stored at byte  ((kb*4 + d)*4 + g)*ncol*16 + (col0 + c)*16 + 4s.
// Digit column 7 of each vector is zero.
// ncol = 8 (ATx: 2 KiB per K-block) or 16 (Ax: [c | e], 4 KiB per K-block); col0 = first column of this vector.
// part[j] / nblocks: the block partials the prep launch left for the vector of slot j; scal[j]: where block 0 of a slot with
// wr[j] != 0 stores the vector's four scalars (two slots may quantise two operands of ONE vector -- c and e -- with one scale)
struct QuantArgs { const double* v[4]; double* scal[4]; const double* part[4]; uint32_t* out[4]; int col0[4]; int wr[4]; int nblocks; };
// the scale of this block's vector from the prep launch's block partials (every block for itself, same order: same bits)
__device__ __forceinline__ double quant_scale(const QuantArgs& a) {
    __shared__ double shm[256], shs[256];
    double sc[4];
    prep_scalars(a.part[blockIdx.y], a.nblocks, shm, shs, sc);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.wr[blockIdx.y]) {
        double* o = a.scal[blockIdx.y];
        o[0] = sc[0]; o[1] = sc[1]; o[2] = sc[2]; o[3] = sc[3];
    }
    return sc[2];
}
// WIDE (kernel mode 2, two-level fixed point): the vector's digits go to columns col0 .. col0 + 7 as always (the HEAD: q = rint(v *
// 2^(54-e))), and the digits of the exact residual r = v - q 2^(e-54) -- |r| <= 2^(e-55), representable: v's own ulp is finer than
// the grid wherever r != 0 -- quantised with the fixed second exponent e - 54 (q2 = rint(r 2^(108-e)), |q2| <= 2^53) go to columns
// col0 + 8 .. col0 + 15.  v = q 2^(e-54) + q2 2^(e-108) to within 2^(e-109): ~108 bits below the vector's largest entry instead of
// 54, in the two slots of ONE two-vector pass.  (A vector whose largest entry is below 2^-915 has no second level: its residual
// multiplier would overflow.)
__device__ __forceinline__ void digits_of(long long q, int t, uint32_t (&dig)[8]) {
#pragma unroll
    for (int c = 0; c < 7; c++) {
        long long dg = (long long)(signed char)(q & 0xFF);   // balanced digit in [-128, 127]
        q = (q - dg) >> 8;
        dig[c] |= (uint32_t)(dg & 0xFF) << (8 * t);
    }
}
__device__ __forceinline__ double wide_mult2(double mult) { return (mult > 0.0 && mult < 0x1p969) ? mult * 0x1p54 : 0.0; }
template <bool WIDE>
__global__ __launch_bounds__(256) void k_quant(QuantArgs a, int64_t n, int64_t nkb, int ncol) {
    const double* __restrict__ v = a.v[blockIdx.y];
    uint32_t* __restrict__ out = a.out[blockIdx.y];
    const int col0 = a.col0[blockIdx.y];
    const double mult = quant_scale(a);
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nkb * 64) return;
    const int s = tid & 3, d = (tid >> 2) & 3, g = (tid >> 4) & 3;
    const int64_t kb = tid >> 6;
    const double mult2 = WIDE ? wide_mult2(mult) : 0.0, sinv = (WIDE && mult > 0.0) ? 1.0 / mult : 0.0;   // (powers of two: exact)
    uint32_t dig[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dig2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        int64_t k = kb * 256 + 64 * g + 16 * d + 4 * t + s;
        double val = (k < n) ? v[k] : 0.0;
        const double qd = rint(val * mult);
        digits_of((long long)qd, t, dig);
        if (WIDE && mult2 > 0.0) digits_of((long long)rint(fma(-qd, sinv, val) * mult2), t, dig2);
    }
    // dword units: K-block = 64*ncol, dword-slot d = 16*ncol, lane-group g = 4*ncol, column = 4
    uint32_t* o = out + kb * (64 * ncol) + d * (16 * ncol) + g * (4 * ncol) + col0 * 4 + s;
#pragma unroll
    for (int c = 0; c < 8; c++) o[c * 4] = dig[c];
    if (WIDE) {
#pragma unroll
        for (int c = 0; c < 8; c++) o[(8 + c) * 4] = dig2[c];
    }
}

// ---- the matvec kernel ----------------------------------------------------------------------------------------------
// One kernel for all products.  A workgroup = 4 waves = 4 consecutive row groups (64 rows each) over the SAME K range.
// Per K-block (256 K-entries) each wave reads its own supertile (4 tiles x 1 KiB, streamed once, non-temporal) and the
// workgroup reads ONE digit block; per (tile, dword) a wave issues two MFMAs: plane r' -> accX, plane miss -> accY.
//   MODE 0 (ATx):  digit block = digits(p), 8 columns (2 KiB); lanes c >= 8 alias column c-8 (their results are never
//                  stored).  result rows: X = accX[:, 0:8], Y = accY[:, 0:8].
//   MODE 1 (Ax):   digit block = [digits(c) | digits(e)], 16 columns (4 KiB).  r'.c comes out in accX[:, 0:8] and
//                  miss.e in accY[:, 8:16]; their sum is stored as one plane per piece.
//   MODE 2 (two N-vectors, ATx / p-value sums): [digits(p_a) | digits(p_b)]; columns 0..7 / 8..15 = vector a / b.
//   MODE 3 (two M-vectors, Ax): the r' plane multiplies dig0 = [c_a | c_b], the miss plane dig1 = [e_a | e_b] (8 KiB):
//                  vector a = accX[:, 0:8] + accY[:, 0:8], vector b = accX[:, 8:16] + accY[:, 8:16].
// The LMMSE solve and the Onsager probe solve of one VAMP iteration are CG runs on the SAME operator (vamp.cpp:593-596
// and :884); MODE 2 / 3 let their matvecs share the stream over the genotype shard at the cost of one.
//
// Pipeline.  Stripes: three supertile register buffers, two K-blocks in flight from HBM per wave, three waves per SIMD.
// Digits: fetched once per WORKGROUP (16-32 B per thread), parked in a two-stage LDS ring, read back with one
// ds_read_b128 per dword-slot -- a wave fetching its own copy from L2 (the first version of this kernel) put 1.5-3x the
// HBM stream on the vector-memory path and cost 6 % (MODE 1) to 25 % (MODE 3).  One workgroup barrier per K-block: a
// bare s_barrier behind an LDS-only fence, so the stripe prefetches (vmcnt) stay in flight across it.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ABuf { u32x4 t[4]; };

__device__ __forceinline__ void load_a(ABuf& a, const u32x4* __restrict__ ap) {
#pragma unroll
    for (int i = 0; i < 4; i++) a.t[i] = __builtin_nontemporal_load(ap + i * 64);
}

// MODE 5 / 6 (kernel mode 2, "two-level fixed point"): the shapes of MODE 2 / 3 with the two vector slots holding the HEAD digits
// and the RESIDUAL digits of ONE vector, and with planes that give a missing genotype an exact zero:
//   first plane  a' = a if present else 0   (byte LUT r' -> {0,1,2,0}),   second plane  b = 1 if present else 0  (r' -> {1,1,1,0})
// so that  Ax : out = sum_i a'_ni c_i - sum_i b_ni (mave_i c_i)      ATx : sum a p = a'.p,  sum b p = b.p
// hold term by term as in the reference's table (data.cpp:951-988) -- no "3 c + (mave - 3) c - mave c" that cancels only to rounding.
template <int MODE> struct MD {
    static constexpr bool two_ax = MODE == 3 || MODE == 6;       // dig0 | dig1 hold one operand each of two slots
    static constexpr bool two_atx = MODE == 2 || MODE == 5;
    static constexpr int planes = MODE == 4 ? 1 : (MODE >= 5 ? 2 : 0);   // 0: r' / missing, 1: a^2 / missing, 2: a' / present
};
template <int PK> __device__ __forceinline__ uint32_t plane_x(uint32_t e) {
    return PK == 0 ? e : __builtin_amdgcn_perm(PK == 1 ? 0x00040100u : 0x00020100u, PK == 1 ? 0x00040100u : 0x00020100u, e);
}
template <int PK> __device__ __forceinline__ uint32_t plane_y(uint32_t e) {
    return __builtin_amdgcn_perm(PK == 2 ? 0x00010101u : 0x01000000u, PK == 2 ? 0x00010101u : 0x01000000u, e);
}

template <int MODE>
struct BFrag { u32x4 d[4]; u32x4 e[MD<MODE>::two_ax ? 4 : 1]; };

template <int MODE>
__device__ __forceinline__ void lds_read_b(BFrag<MODE>& b, const u32x4* sb, int bofs) {
    constexpr int DS = (MODE == 0) ? 32 : 64;
#pragma unroll
    for (int d = 0; d < 4; d++) b.d[d] = sb[d * DS + bofs];
    if (MD<MODE>::two_ax) {
#pragma unroll
        for (int d = 0; d < 4; d++) b.e[d] = sb[256 + d * 64 + bofs];
    }
}

template <int MODE>
__device__ __forceinline__ void compute_step(const ABuf& a, const BFrag<MODE>& b, v4i (&accX)[4], v4i (&accY)[4]) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const v4i BX = {(int)b.d[d].x, (int)b.d[d].y, (int)b.d[d].z, (int)b.d[d].w};
        const u32x4 be = MD<MODE>::two_ax ? b.e[d] : b.d[d];
        const v4i BY = {(int)be.x, (int)be.y, (int)be.z, (int)be.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t w = a.t[i][d];
            const uint32_t e0 = w & 0x03030303u, e1 = (w >> 2) & 0x03030303u, e2 = (w >> 4) & 0x03030303u,
                           e3 = (w >> 6) & 0x03030303u;
            // MODE 4: the first plane is a^2 instead of r' (byte LUT r' -> {0,1,4,0}: code 3 = missing -> 0), for sums that are
            // quadratic in the genotype (compute_people_statistics, data.cpp:608-624); MODE 5 / 6: a' and the present plane
            constexpr int PK = MD<MODE>::planes;
            const v4i X = {(int)plane_x<PK>(e0), (int)plane_x<PK>(e1), (int)plane_x<PK>(e2), (int)plane_x<PK>(e3)};
            accX[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(X, BX, accX[i], 0, 0, 0);
            const v4i Y = {(int)plane_y<PK>(e0), (int)plane_y<PK>(e1), (int)plane_y<PK>(e2), (int)plane_y<PK>(e3)};
            accY[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Y, BY, accY[i], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void wg_barrier_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Work decomposition.  The work is the grid of (quad = 4 row groups, K-block) cells, linearised quad-major:
// u = quad * nkb + kb.
//   uniform (skL == 0): grid = ceil(nrg / 4) * ksplit workgroups; workgroup (quad, ks) walks the ks-th of ksplit equal
//       K-segments of its quad.  All workgroups resident on an XCD walk the same K range in step.
//   balanced (skL > 0): workgroup w walks cells [w skL, (w+1) skL) whatever quads they fall in, flushing its accumulators
//       where the range crosses into the next quad.  With a grid of exactly the resident workgroups (768 = 256 CUs x 3)
//       every workgroup gets the same number of cells and the launch has no ragged last round -- worth 10-17 % when
//       ceil(nrg / 4) * ksplit is just above a multiple of 768 for every admissible ksplit (e.g. N = 50k: 782 quads).
// partial layout (int32): [(piece * P + plane) * rows_p + row] * 8 + digit, rows_p = 64 * nrg; P = 2 (MODE 0: sum r' p, sum miss p),
// 4 (MODE 2: planes 2v, 2v+1 belong to vector v), 1 (MODE 1, 4: the two products already added) or 2 (MODE 3: plane v = vector v); piece = ks (uniform) or w - floor(quad nkb / skL) (balanced: the
// workgroups that touch a quad are consecutive).  pieces_of() below is the count the epilogue kernels sum over.
//   hybrid (skL > 0, piv > 0): the first piv quads -- a whole number of rounds of the resident workgroups -- go one quad per
//       workgroup over the whole K range (uniform, ks = 1: co-resident workgroups walk K in step, a digit block is fetched once
//       per XCD), and only the remaining nq - piv quads are cut into balanced ranges of skL cells.  For shards whose quad count
//       is just above a multiple of 768: neither the ragged last round of a uniform split nor the scattered K offsets of a
//       fully balanced grid (1.13 x HBM over-fetch of digit blocks on the two-vector Ax at N = 400k x M = 125k).  The
//       remainder's workgroups take the first block indices (they are dispatched first and are short).
__device__ __forceinline__ int pieces_of(int64_t row, int ksplit, int64_t nkb, int64_t skL, int qshift = 8, int64_t piv = 0) {
    if (skL <= 0) return ksplit;
    int64_t q = row >> qshift;   // rows per quad of row groups: 256 (64-row groups) or 1024 (tile layout, Ax side)
    if (q < piv) return 1;
    q -= piv;
    return (int)(((q + 1) * nkb - 1) / skL - (q * nkb) / skL + 1);
}

#ifdef GV_WGTIME
__device__ unsigned long long g_wgt[4 * 16384];   // development build only: per-workgroup start / end clock, XCC id, CU id
#endif
// K-block boundaries of the segments of a uniform split (b[0] = 0 ... b[ks] = nkb; ks <= GV_MAX_KS)
constexpr int GV_MAX_KS = 64;
// [c]: the boundaries of a quad of parity c = q & 1 (Decomp::xskew: segment j of quad q is run by a workgroup whose block index has
// the parity of q + j; odd block indices land on the four XCDs that stream 4-6 % faster -- their segments are made that much longer)
struct KBounds { uint32_t b[2][GV_MAX_KS + 1]; };

// GO: the instantiation the device-resident CG loop launches (cg_run_device) -- it alone carries the `go` test, and it shows up
// under its own name in kernel traces, so that the ~3 us launches of a dropped CG step do not dilute the statistics of the
// streaming kernel proper.
template <int MODE, bool SK, bool GO>
__global__ __launch_bounds__(256, 3) void k_mfma_matvec(const u32x4* __restrict__ stripes, const u32x4* __restrict__ dig0,
                                                 const u32x4* __restrict__ dig1, int64_t nrg, int64_t nkb, int ksplit,
                                                 int64_t skL, int prio, KBounds kbnd, int32_t* __restrict__ partial,
                                                 const int* __restrict__ go, uint32_t piv) {
    // device-resident CG: a step enqueued before the host knew that every system had converged is dropped here
    if (GO && __builtin_nontemporal_load(go) == 0) return;
    constexpr int KBS = (MODE == 0) ? 128 : 256;   // u32x4 per K-block of one digit buffer
    constexpr int SS = MD<MODE>::two_ax ? 512 : 256;    // u32x4 per LDS stage (MODE 0 uses the first 128; the rest is a dummy target)
    __shared__ u32x4 sB[2][SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t nq = (nrg + 3) >> 2;
    const int c = lane & 15, g = lane >> 4;
    const int bofs = (MODE == 0) ? g * 8 + (c & 7) : lane;
    const int dofs = (MODE == 0) ? (tid & 127) : tid;   // MODE 0: the block is 128 pieces; threads 128..255 copy pieces 0..127
    // again into the unused half of the stage, so that the copy has no divergent branch (the loop body stays one basic block)
    // cells of this workgroup: uniform over the workgroup, kept in SGPRs (32-bit: nq * nkb = M N / 65536 cells)
    uint32_t u, uend;
    const uint32_t nkb32 = (uint32_t)nkb, skL32 = (uint32_t)skL;
    if (SK) {
        const uint32_t U = (uint32_t)nq * nkb32;
        const uint32_t nrem = gridDim.x - piv;      // hybrid: workgroups of the balanced remainder (piv == 0: all of them)
        if (blockIdx.x < nrem) {
            u = piv * nkb32 + blockIdx.x * skL32;
            uend = u + skL32 < U ? u + skL32 : U;
        } else {                                    // one whole quad
            u = (blockIdx.x - nrem) * nkb32;
            uend = u + nkb32;
        }
    } else {
        // workgroups are dealt to the eight XCDs round-robin by block index, so the parity of b decides whether a workgroup runs on one of
        // the four faster or the four slower XCDs.  Segment ks0 of quad q0: with an odd number of quads b = q0 + nq ks0 has the parity
        // of q0 + ks0; with an even one the quads are rotated by one per segment row so that it has, too -- a quad's segments
        // alternate between the two kinds of XCD either way
        const uint32_t ks0 = blockIdx.x / (uint32_t)nq;
        uint32_t q0 = blockIdx.x % (uint32_t)nq;
        if (!(nq & 1)) q0 = (q0 + ks0) % (uint32_t)nq;
        u = q0 * nkb32 + kbnd.b[q0 & 1][ks0];
        uend = q0 * nkb32 + kbnd.b[q0 & 1][ks0 + 1];
    }
    // Wave priority by remaining work (prio != 0).  The instruction arbiter favours the oldest wave of a SIMD, so the three
    // workgroups of a CU finish one after the other (measured: 245 / 320 / 385 us for equal work) and a launch ends on a
    // long, thinly occupied tail; a workgroup that is ahead of its neighbours now yields to them.  Four levels, lowered at
    // each quarter of the workgroup's cells.  Whether it pays depends on the shape (many short rounds: yes; few long
    // rounds: no), so it is one of the things autotune_ks measures.
    const uint32_t p_quarter = (uend - u + 3) / 4;
    uint32_t p_thr = prio ? u + p_quarter : 0xffffffffu, p_lvl = 3;
    if (prio) __builtin_amdgcn_s_setprio(3);
    ABuf a0, a1, a2;          // carried across the segments of a balanced range: a0 / a1 enter a segment holding its cells 0 / 1
    u32x4 r0, r1;
    bool primed = false;      // the previous segment already fetched this one's first two supertiles and first digit block
    uint32_t par = 0;         // LDS stage of this segment's cell 0 (steps done so far, mod 2)
    if (u >= uend) return;
#ifdef GV_WGTIME
    if (tid == 0 && blockIdx.x < 16384) {
        g_wgt[4 * blockIdx.x] = wall_clock64();
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_wgt[4 * blockIdx.x + 2] = (xcc & 0xf) | ((unsigned long long)(uend - u) << 8);     // XCC id | cells of this workgroup
        g_wgt[4 * blockIdx.x + 3] = hw;
    }
#endif
#pragma unroll 1
  do {
    const uint32_t q32 = __builtin_amdgcn_readfirstlane(u / nkb32);
    const int64_t q = q32, kb0 = u - q32 * nkb32;
    const uint32_t seg = nkb32 - (uint32_t)kb0 < uend - u ? nkb32 - (uint32_t)kb0 : uend - u;   // to the end of the quad or of the range
    const int64_t nsteps = seg;
    const int ks = SK ? (q32 < piv ? 0 : (int)__builtin_amdgcn_readfirstlane(blockIdx.x - ((q32 - piv) * nkb32) / skL32))
                      : (int)(blockIdx.x / (uint32_t)nq);
    const uint32_t useg0 = u;   // first cell of this segment
    u += seg;
    int64_t rg = q * 4 + (tid >> 6);
    const bool live = rg < nrg;      // a dead wave still loads digits and meets the barriers; it re-reads the last row group
    if (!live) rg = nrg - 1;

    v4i accX[4], accY[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        accX[i] = (v4i){0, 0, 0, 0};
        accY[i] = (v4i){0, 0, 0, 0};
    }
    const int64_t last = nsteps - 1;
    const u32x4* ap = stripes + (rg * nkb + kb0) * 256 + lane;
    const u32x4* g0 = dig0 + kb0 * KBS + dofs;
    const u32x4* g1 = MD<MODE>::two_ax ? dig1 + kb0 * KBS + tid : nullptr;
    // Look-ahead past the end of the segment.  Uniform K-split, or last segment of a balanced range: the last cell again
    // (clamped: no branches around loads).  Balanced range with another segment to come: that segment's first cells -- it
    // starts at K-block 0 of quad q + 1 -- so the stream of loads never drains at a quad boundary.
    const bool has_next = SK && u < uend && nkb32 >= 2;
    int64_t rgn = (q + 1) * 4 + (tid >> 6);
    if (rgn >= nrg) rgn = nrg - 1;
    const u32x4* apn = has_next ? stripes + rgn * nkb * 256 + lane : ap + last * 256;
    const u32x4* g0n = has_next ? dig0 + dofs : g0 + last * KBS;
    const u32x4* g1n = MD<MODE>::two_ax ? (has_next ? dig1 + tid : g1 + last * KBS) : nullptr;
    const int64_t nlastn = has_next ? (int64_t)(uend - u < nkb32 ? uend - u : nkb32) - 1 : 0;
#define GV_A_AT(I) ((I) <= last ? ap + (I) * 256 : apn + ((I) - nsteps < nlastn ? (I) - nsteps : nlastn) * 256)
#define GV_D_AT(G, GN, I) ((I) <= last ? (G) + (I) * KBS : (GN) + ((I) - nsteps < nlastn ? (I) - nsteps : nlastn) * KBS)

    if (!primed) {
        r0 = g0[0];
        if (MD<MODE>::two_ax) r1 = g1[0];
        load_a(a0, ap);
        load_a(a1, GV_A_AT((int64_t)1));
        sB[par][tid] = r0;
        if (MD<MODE>::two_ax) sB[par][256 + tid] = r1;
        wg_barrier_lds();
    }

    // One step: prefetch the digits of cell S+1 and the stripes of cell S+2, multiply cell S out of CUR and its LDS stage,
    // park the digits in the other stage, meet the workgroup.
    // vmcnt retires in order: the digit load is issued BEFORE the 4 stripe loads so that the ds_write waits for it only.
#define GV_WG_STEP(CUR, NXT, S)                                                        \
    {                                                                                  \
        const int64_t sv = (S);                                                        \
        const int64_t n2 = sv + 2, n1 = sv + 1;                                        \
        r0 = *GV_D_AT(g0, g0n, n1);                                                    \
        if (MD<MODE>::two_ax) r1 = *GV_D_AT(g1, g1n, n1);                              \
        load_a(NXT, GV_A_AT(n2));                                                      \
        BFrag<MODE> bf;                                                                \
        lds_read_b<MODE>(bf, sB[(sv + par) & 1], bofs);                                \
        compute_step<MODE>(CUR, bf, accX, accY);                                     \
        u32x4* sw = sB[(sv + 1 + par) & 1];                                            \
        sw[tid] = r0;                                                                  \
        if (MD<MODE>::two_ax) sw[256 + tid] = r1;                                      \
        wg_barrier_lds();                                                              \
    }
    // nsteps % 3 leading steps rotate the three supertile buffers by register moves; the main loop rotates them by NAME
    // (three steps per trip, one basic block), so a stripe load is only waited for two steps after it was issued.
    int64_t st = 0;
    const int64_t rem = nsteps % 3;
#pragma unroll 1
    for (; st < rem; st++) {
        GV_WG_STEP(a0, a2, st)
        a0 = a1;
        a1 = a2;
    }
#pragma unroll 1
    for (; st < nsteps; st += 3) {
        if (useg0 + (uint32_t)st >= p_thr) {
            p_thr += p_quarter;
            p_lvl = p_lvl > 0 ? p_lvl - 1 : 0;
            if (p_lvl == 2) __builtin_amdgcn_s_setprio(2);
            else if (p_lvl == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        GV_WG_STEP(a0, a2, st)
        GV_WG_STEP(a1, a0, st + 1)
        GV_WG_STEP(a2, a1, st + 2)
    }
#undef GV_WG_STEP
#undef GV_A_AT
#undef GV_D_AT
    par = (par + (uint32_t)nsteps) & 1u;
    primed = has_next;
    // (every wave has passed the barrier of the last step: nobody still reads the LDS ring when the next segment refills it)
    if (!live) continue;
    const int64_t rows_p = nrg * 64;
    const int cd = c & 7;
    if (MD<MODE>::two_ax) {
        // vector v = c >> 3 needs r'.c_v + miss.e_v only: both sit in the same lanes, so one plane per vector is stored
        // (|sum| <= 512 K per digit, the bound gv_set_dims sizes the K-segments for)
        const int pv = c >> 3;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int64_t row = rg * 64 + 16 * i + 4 * g + reg;
                partial[(((int64_t)ks * 2 + pv) * rows_p + row) * 8 + cd] = accX[i][reg] + accY[i][reg];
            }
        }
        continue;
    }
    if (MD<MODE>::two_atx) {
        const int pv = (c >> 3) * 2;
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int64_t row = rg * 64 + 16 * i + 4 * g + reg;
                partial[(((int64_t)ks * 4 + pv + 0) * rows_p + row) * 8 + cd] = accX[i][reg];
                partial[(((int64_t)ks * 4 + pv + 1) * rows_p + row) * 8 + cd] = accY[i][reg];
            }
        }
        continue;
    }
    if (MODE == 1 || MODE == 4) {
        // r'.c sits in lanes c < 8 of accX, miss.e in lanes c >= 8 of accY: bring the latter 8 lanes down within each row of
        // 16 (DPP row_shl:8) and store their sum as the one plane of this piece (|sum| <= 512 K per digit, as for MODE 3)
#pragma unroll
        for (int i = 0; i < 4; i++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int64_t row = rg * 64 + 16 * i + 4 * g + reg;
                const int ye = __builtin_amdgcn_mov_dpp(accY[i][reg], 0x108 /* row_shl:8 */, 0xf, 0xf, true);
                if (c < 8) partial[((int64_t)ks * rows_p + row) * 8 + cd] = accX[i][reg] + ye;
            }
        }
        continue;
    }
    // MODE 0: sum r' p and sum miss p are needed separately (the row's mean multiplies the second), two planes
#pragma unroll
    for (int i = 0; i < 4; i++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int64_t row = rg * 64 + 16 * i + 4 * g + reg;
            if (c < 8) {
                partial[(((int64_t)ks * 2 + 0) * rows_p + row) * 8 + cd] = accX[i][reg];
                partial[(((int64_t)ks * 2 + 1) * rows_p + row) * 8 + cd] = accY[i][reg];
            }
        }
    }
  } while (SK && u < uend);
#ifdef GV_WGTIME
    if (tid == 0 && blockIdx.x < 16384) g_wgt[4 * blockIdx.x + 1] = wall_clock64();
#endif
}
#ifdef GV_WGTIME
}  // namespace
extern "C" int gv_debug_wgtime(unsigned long long* out, int n) {
    if (!out) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_wgt)) != hipSuccess) return 1;
        return (int)hipMemset(p, 0, sizeof(unsigned long long) * 4 * 16384);
    }
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wgt), sizeof(unsigned long long) * 4 * (size_t)n);
}
namespace {
#endif

// =====================================================================================================================
// ONE resident layout for both products ("tile" layout, kernel mode 2): M N / 4 bytes instead of 2 x M N / 4.
//
// The MFMA contracts over the bytes a lane holds, so a layout serves a product only if, after the 2-bit -> byte expansion,
// the bytes of an operand register run along that product's K.  The expansion (w >> 2s) & 0x03030303 turns bit pair s of
// every byte into a byte: bytes keep their meaning, bit pairs become the register index.  Pack a dword as a 4 x 4 block
//     D(q, j) : byte t <-> marker 4q + t,  bit pair s <-> individual 4j + s
// and  * Ax  (K = markers)     is native: (D >> 2s) & mask = 4 consecutive markers (bytes = K) of individual 4j + s;
//      * ATx (K = individuals) needs bytes along individuals: a 4 x 4 BYTE transpose of the four dwords D(q, 4J..4J+3) a lane
//        holds (8 v_perm_b32) gives T[t] = marker 4q + t, byte d <-> individuals 16J + 4d + (bit pair), which is exactly the
//        1 x 16 packing the two-layout kernel expands.  +2 VALU per dword (13 instead of 11), nothing else.
// Memory: super-block SB(RG, KB) = 64 markers x 256 individuals = 4 KiB at ((RG nkb + KB) 4 KiB): 16 marker quads
// (q_local) x 16 individual groups of 16 (J_local), as four 1 KiB blocks JB = J_local >> 2, each ordered
// [q_local][J_local & 3] in 16-byte pieces P(q, J) = D(q, 4J .. 4J+3).
//   ATx wave, load i = block JB = i: lane (r, g) takes P(q_local = r, J_local = 4i + g)      -- the whole KiB, lanes permuted
//   Ax  wave, load i: lane (rho, gam) takes P(q_local = 4i + gam, J_local = rho)            -- 256 contiguous bytes of each block
// so both directions consume a super-block with four wave-wide 16-byte loads that touch only whole 128-byte lines.
// ATx: 4 row tiles t (marker 4 q + t) x K = 64 individuals per load, digits in the order of k_quant (element g*64 + i*16 + c).
// Ax: 16 row tiles (d, s) (individual 16 rho + 4 d + s) x K = 64 markers per super-block; operand register i comes from
// load i; the r' plane multiplies dig0 = [c_a | c_b], the missing plane dig1 = [e_a | e_b], both into ONE accumulator per
// tile (64 accumulator registers); digits in the order of k_quant_t.  A one-vector Ax is the two-vector pass with nv = 1.
// All sums are the same integers as in the two-layout kernels: results are bit-identical.
struct ABufT { u32x4 t[4]; };

template <int DIR>   // 0: ATx side (rows = markers), 1: Ax side (rows = individuals)
__device__ __forceinline__ void load_t(ABufT& a, const u32x4* __restrict__ sb, int lane) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int off = (DIR == 0) ? i * 64 + r * 4 + g : (r >> 2) * 64 + (4 * i + g) * 4 + (r & 3);
        a.t[i] = __builtin_nontemporal_load(sb + off);
    }
}

// ATx side, one K-block (256 individuals): per load i a byte transpose, then 4 tiles x (r' plane, missing plane)
template <int MODE>   // 0: one vector (8 digit columns, lanes c >= 8 alias), 2: two vectors, 5: head | residual of one vector (MD)
__device__ __forceinline__ void compute_atx_t(const ABufT& a, const u32x4* sb, int lane, v4i (&accX)[4], v4i (&accY)[4]) {
    const int c = lane & 15, g = lane >> 4;
    constexpr int PK = MD<MODE>::planes;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const u32x4 bq = (MODE == 0) ? sb[g * 32 + i * 8 + (c & 7)] : sb[g * 64 + i * 16 + c];
        const v4i B = {(int)bq.x, (int)bq.y, (int)bq.z, (int)bq.w};
        const uint32_t L0 = a.t[i].x, L1 = a.t[i].y, L2 = a.t[i].z, L3 = a.t[i].w;
        const uint32_t a0 = __builtin_amdgcn_perm(L1, L0, 0x05010400u), a1 = __builtin_amdgcn_perm(L1, L0, 0x07030602u);
        const uint32_t b0 = __builtin_amdgcn_perm(L3, L2, 0x05010400u), b1 = __builtin_amdgcn_perm(L3, L2, 0x07030602u);
        const uint32_t T[4] = {__builtin_amdgcn_perm(b0, a0, 0x05040100u), __builtin_amdgcn_perm(b0, a0, 0x07060302u),
                               __builtin_amdgcn_perm(b1, a1, 0x05040100u), __builtin_amdgcn_perm(b1, a1, 0x07060302u)};
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint32_t w = T[t];
            const uint32_t e0 = w & 0x03030303u, e1 = (w >> 2) & 0x03030303u, e2 = (w >> 4) & 0x03030303u,
                           e3 = (w >> 6) & 0x03030303u;
            const v4i X = {(int)plane_x<PK>(e0), (int)plane_x<PK>(e1), (int)plane_x<PK>(e2), (int)plane_x<PK>(e3)};
            accX[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(X, B, accX[t], 0, 0, 0);
            const v4i Y = {(int)plane_y<PK>(e0), (int)plane_y<PK>(e1), (int)plane_y<PK>(e2), (int)plane_y<PK>(e3)};
            accY[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Y, B, accY[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // (no barrier, or one MFMA per seven vector instructions through sched_group_barrier: 15.53-15.61 ms
                                                // against 15.54-15.67 at the headline, round 4 -- the vector ALU is busy, not badly ordered)
    }
}

// Ax side, one K-step (64 markers): 16 tiles (d, s), operand register i from load i
template <int PK>   // planes (MD::planes): 0 r' / missing, 1 a^2 / missing (compute_people_statistics), 2 a' / present (two-level fixed point)
__device__ __forceinline__ void compute_ax_t(const ABufT& a, const u32x4* sb, int lane, v4i (&acc)[16]) {
    const u32x4 q0 = sb[lane], q1 = sb[64 + lane];
    const v4i B0 = {(int)q0.x, (int)q0.y, (int)q0.z, (int)q0.w};
    const v4i B1 = {(int)q1.x, (int)q1.y, (int)q1.z, (int)q1.w};
#pragma unroll
    for (int d = 0; d < 4; d++) {
#pragma unroll
        for (int s2 = 0; s2 < 4; s2++) {
            const uint32_t e0 = (a.t[0][d] >> (2 * s2)) & 0x03030303u, e1 = (a.t[1][d] >> (2 * s2)) & 0x03030303u,
                           e2 = (a.t[2][d] >> (2 * s2)) & 0x03030303u, e3 = (a.t[3][d] >> (2 * s2)) & 0x03030303u;
            const v4i X = {(int)plane_x<PK>(e0), (int)plane_x<PK>(e1), (int)plane_x<PK>(e2), (int)plane_x<PK>(e3)};
            const v4i Y = {(int)plane_y<PK>(e0), (int)plane_y<PK>(e1), (int)plane_y<PK>(e2), (int)plane_y<PK>(e3)};
            acc[d * 4 + s2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(X, B0, acc[d * 4 + s2], 0, 0, 0);
            acc[d * 4 + s2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Y, B1, acc[d * 4 + s2], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// The streaming kernel on the tile layout.  Same skeleton as k_mfma_matvec: a workgroup = 4 waves = 4 consecutive row
// groups over the same K range (digits once per workgroup through a two-stage LDS ring, one bare s_barrier per step), three
// super-block register buffers rotating by name, uniform / tapered K-split or balanced decomposition, progress-based wave
// priority.  DIR 0 (ATx): row group = 64 markers (nrg = ceil(M/64)), K-step = 256 individuals (nkb = ceil(N/256)), digit
// block 128 (MODE 0) or 256 (MODE 2) pieces.  DIR 1 (Ax): row group = 256 individuals (nrg = ceil(N/256)), K-step = 64
// markers (nkb = ceil(M/64)), digit block = 64 pieces of dig0 + 64 of dig1.  Super-block of (row group rg, K-step kb):
// DIR 0: (rg nkb + kb), DIR 1: (kb nrg + rg) -- the layout is stored marker-group-major.
// partial layout as k_mfma_matvec: [(piece * P + plane) * rows_p + row] * 8 + digit; DIR 0: rows_p = 64 nrg, P = 2 (MODE 0)
// or 4 (MODE 2); DIR 1: rows_p = 256 nrg, P = nv.
template <int DIR, int MODE, bool SK, bool GO>
__global__ __launch_bounds__(256, 3) void k_mfma_tile(const u32x4* __restrict__ stripes, const u32x4* __restrict__ dig0,
                                                      const u32x4* __restrict__ dig1, int64_t nrg, int64_t nkb, int ksplit,
                                                      int64_t skL, int prio, KBounds kbnd, int32_t* __restrict__ partial, int nv,
                                                      const int* __restrict__ go, int64_t rstride, uint32_t piv) {
    // rstride (DIR 1): row groups per K-step in memory -- nrg, or more when this launch covers a sub-range of the row groups
    if (GO && __builtin_nontemporal_load(go) == 0) return;
    constexpr int KBS = (DIR == 1) ? 64 : ((MODE == 0) ? 128 : 256);   // u32x4 per K-step of one digit buffer
    constexpr int SS = (DIR == 1) ? 128 : 256;                         // u32x4 per LDS stage
    constexpr int ROWS = (DIR == 1) ? 256 : 64;                        // rows per row group
    __shared__ u32x4 sB[2][SS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t nq = (nrg + 3) >> 2;
    const int c = lane & 15, g = lane >> 4;
    // the digit copy: DIR 0: thread -> piece (MODE 0: pieces 0..127, threads 128..255 copy them again into the unused half);
    // DIR 1: threads 0..63 -> dig0, 64..127 -> dig1, 128..255 repeat (no divergent branch in the loop body)
    const int dofs = (DIR == 1) ? (tid & 63) : ((MODE == 0) ? (tid & 127) : tid);
    const u32x4* digsel = (DIR == 1 && (tid & 64)) ? dig1 : dig0;
    const int sofs = (DIR == 1) ? (tid & 127) : tid;
    uint32_t u, uend;
    const uint32_t nkb32 = (uint32_t)nkb, skL32 = (uint32_t)skL;
    if (SK) {
        const uint32_t U = (uint32_t)nq * nkb32;
        const uint32_t nrem = gridDim.x - piv;      // hybrid: workgroups of the balanced remainder (piv == 0: all of them)
        if (blockIdx.x < nrem) {
            u = piv * nkb32 + blockIdx.x * skL32;
            uend = u + skL32 < U ? u + skL32 : U;
        } else {                                    // one whole quad
            u = (blockIdx.x - nrem) * nkb32;
            uend = u + nkb32;
        }
    } else {
        // workgroups are dealt to the eight XCDs round-robin by block index, so the parity of b decides whether a workgroup runs on one of
        // the four faster or the four slower XCDs.  Segment ks0 of quad q0: with an odd number of quads b = q0 + nq ks0 has the parity
        // of q0 + ks0; with an even one the quads are rotated by one per segment row so that it has, too -- a quad's segments
        // alternate between the two kinds of XCD either way
        const uint32_t ks0 = blockIdx.x / (uint32_t)nq;
        uint32_t q0 = blockIdx.x % (uint32_t)nq;
        if (!(nq & 1)) q0 = (q0 + ks0) % (uint32_t)nq;
        u = q0 * nkb32 + kbnd.b[q0 & 1][ks0];
        uend = q0 * nkb32 + kbnd.b[q0 & 1][ks0 + 1];
    }
    const uint32_t p_quarter = (uend - u + 3) / 4;
    uint32_t p_thr = prio ? u + p_quarter : 0xffffffffu, p_lvl = 3;
    if (prio) __builtin_amdgcn_s_setprio(3);
    ABufT a0, a1, a2;
    u32x4 r0;
    bool primed = false;
    uint32_t par = 0;
    if (u >= uend) return;
#ifdef GV_WGTIME
    if (tid == 0 && blockIdx.x < 16384) {
        g_wgt[4 * blockIdx.x] = wall_clock64();
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_wgt[4 * blockIdx.x + 2] = (xcc & 0xf) | ((unsigned long long)(uend - u) << 8);     // XCC id | cells of this workgroup
        g_wgt[4 * blockIdx.x + 3] = hw;
    }
#endif
    // super-block (row group rg_, K-step kb_) in u32x4 units
    const int64_t kstride = (DIR == 1) ? rstride * 256 : 256;   // from one K-step to the next
#pragma unroll 1
  do {
    const uint32_t q32 = __builtin_amdgcn_readfirstlane(u / nkb32);
    const int64_t q = q32, kb0 = u - q32 * nkb32;
    const uint32_t seg = nkb32 - (uint32_t)kb0 < uend - u ? nkb32 - (uint32_t)kb0 : uend - u;
    const int64_t nsteps = seg;
    const int ks = SK ? (q32 < piv ? 0 : (int)__builtin_amdgcn_readfirstlane(blockIdx.x - ((q32 - piv) * nkb32) / skL32))
                      : (int)(blockIdx.x / (uint32_t)nq);
    const uint32_t useg0 = u;
    u += seg;
    int64_t rg = q * 4 + (tid >> 6);
    const bool live = rg < nrg;
    if (!live) rg = nrg - 1;

    v4i accX[(DIR == 1) ? 16 : 4], accY[(DIR == 1) ? 1 : 4];
#pragma unroll
    for (int i = 0; i < ((DIR == 1) ? 16 : 4); i++) accX[i] = (v4i){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < ((DIR == 1) ? 1 : 4); i++) accY[i] = (v4i){0, 0, 0, 0};
    const int64_t last = nsteps - 1;
    const u32x4* ap = stripes + ((DIR == 1) ? (kb0 * rstride + rg) * 256 : (rg * nkb + kb0) * 256);
    const u32x4* g0 = digsel + kb0 * KBS + dofs;
    const bool has_next = SK && u < uend && nkb32 >= 2;
    int64_t rgn = (q + 1) * 4 + (tid >> 6);
    if (rgn >= nrg) rgn = nrg - 1;
    const u32x4* apn = has_next ? stripes + ((DIR == 1) ? rgn * 256 : rgn * nkb * 256) : ap + last * kstride;
    const u32x4* g0n = has_next ? digsel + dofs : g0 + last * KBS;
    const int64_t nlastn = has_next ? (int64_t)(uend - u < nkb32 ? uend - u : nkb32) - 1 : 0;
#define GV_A_AT(I) ((I) <= last ? ap + (I) * kstride : apn + ((I) - nsteps < nlastn ? (I) - nsteps : nlastn) * kstride)
#define GV_D_AT(I) ((I) <= last ? g0 + (I) * KBS : g0n + ((I) - nsteps < nlastn ? (I) - nsteps : nlastn) * KBS)

    if (!primed) {
        r0 = g0[0];
        load_t<DIR>(a0, ap, lane);
        load_t<DIR>(a1, GV_A_AT((int64_t)1), lane);
        sB[par][sofs] = r0;
        wg_barrier_lds();
    }
#define GV_T_STEP(CUR, NXT, S)                                                         \
    {                                                                                  \
        const int64_t sv = (S);                                                        \
        const int64_t n2 = sv + 2, n1 = sv + 1;                                        \
        r0 = *GV_D_AT(n1);                                                             \
        load_t<DIR>(NXT, GV_A_AT(n2), lane);                                           \
        if constexpr (DIR == 1)                                                        \
            compute_ax_t<MD<MODE>::planes>(CUR, sB[(sv + par) & 1], lane, accX);       \
        else                                                                           \
            compute_atx_t<MODE>(CUR, sB[(sv + par) & 1], lane, accX, accY);            \
        sB[(sv + 1 + par) & 1][sofs] = r0;                                             \
        wg_barrier_lds();                                                              \
    }
    int64_t st = 0;
    const int64_t rem = nsteps % 3;
#pragma unroll 1
    for (; st < rem; st++) {
        GV_T_STEP(a0, a2, st)
        a0 = a1;
        a1 = a2;
    }
#pragma unroll 1
    for (; st < nsteps; st += 3) {
        if (useg0 + (uint32_t)st >= p_thr) {
            p_thr += p_quarter;
            p_lvl = p_lvl > 0 ? p_lvl - 1 : 0;
            if (p_lvl == 2) __builtin_amdgcn_s_setprio(2);
            else if (p_lvl == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
        GV_T_STEP(a0, a2, st)
        GV_T_STEP(a1, a0, st + 1)
        GV_T_STEP(a2, a1, st + 2)
    }
#undef GV_T_STEP
#undef GV_A_AT
#undef GV_D_AT
    par = (par + (uint32_t)nsteps) & 1u;
    primed = has_next;
    if (!live) continue;
    const int64_t rows_p = nrg * ROWS;
    const int cd = c & 7;
    if constexpr (DIR == 1) {
        // tile (d, s2): D[row 4g + reg][col c] <-> individual 256 rg + 16 (4g + reg) + 4d + s2; vector c >> 3, digit c & 7
        const int pv = c >> 3;
        if (pv < nv) {
#pragma unroll
            for (int t = 0; t < 16; t++) {
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int64_t row = rg * 256 + 16 * (4 * g + reg) + t;   // t = 4 d + s2
                    partial[(((int64_t)ks * nv + pv) * rows_p + row) * 8 + cd] = accX[t][reg];
                }
            }
        }
    } else if constexpr (MD<MODE>::two_atx) {
    // DIR 0: tile t, D[row 4g + reg][col c] <-> marker 64 rg + 4 (4g + reg) + t
        const int pv = (c >> 3) * 2;
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const int64_t row = rg * 64 + 4 * (4 * g + reg) + t;
                partial[(((int64_t)ks * 4 + pv + 0) * rows_p + row) * 8 + cd] = accX[t][reg];
                partial[(((int64_t)ks * 4 + pv + 1) * rows_p + row) * 8 + cd] = accY[t][reg];
            }
        }
    } else {
#pragma unroll
    for (int t = 0; t < 4; t++) {
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int64_t row = rg * 64 + 4 * (4 * g + reg) + t;
            if (c < 8) {
                partial[(((int64_t)ks * 2 + 0) * rows_p + row) * 8 + cd] = accX[t][reg];
                partial[(((int64_t)ks * 2 + 1) * rows_p + row) * 8 + cd] = accY[t][reg];
            }
        }
    }
    }
  } while (SK && u < uend);
#ifdef GV_WGTIME
    if (tid == 0 && blockIdx.x < 16384) g_wgt[4 * blockIdx.x + 1] = wall_clock64();
#endif
}

// ---- tile layout from a raw chunk: markers [m0, m0 + mc) of the shard, m0 % 64 == 0.  block = super-block (rg_local, kb);
// thread tau writes piece tau = JB * 64 + q_local * 4 + (J_local & 3): four 16-entry words (one per marker of the quad),
// recoded, then byte-transposed into the four dwords D(q, 4J + d).
__device__ __forceinline__ void transpose4x4_bytes(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, uint32_t (&o)[4]) {
    const uint32_t a0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), a1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
    const uint32_t b0 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), b1 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(b0, a0, 0x05040100u);
    o[1] = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(b1, a1, 0x05040100u);
    o[3] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
}
__global__ __launch_bounds__(256) void k_tile_build(const uint8_t* __restrict__ raw, int64_t pitch, int64_t mc, int64_t N,
                                                    uint4* __restrict__ tiles, int64_t rg0, int64_t nkb) {
    const int tau = threadIdx.x, JB = tau >> 6, ql = (tau >> 2) & 15, Jl = JB * 4 + (tau & 3);
    const int64_t kb = blockIdx.x, rgl = blockIdx.y;
    const int64_t J = kb * 16 + Jl;                 // group of 16 individuals = one 32-bit word of a raw row
    uint32_t w[4] = {0, 0, 0, 0};
    if (J * 4 < pitch) {
        const int nvalid = (int)min((int64_t)16, max((int64_t)0, N - J * 16));
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int64_t m = rgl * 64 + 4 * ql + t;
            if (m < mc) w[t] = keep_lo(recode(*reinterpret_cast<const uint32_t*>(raw + m * pitch + J * 4)), nvalid);
        }
    }
    uint32_t o[4];
    transpose4x4_bytes(w[0], w[1], w[2], w[3], o);   // o[d] byte t = w[t] byte d
    tiles[((rg0 + rgl) * nkb + kb) * 256 + tau] = make_uint4(o[0], o[1], o[2], o[3]);
}

// compute_markers_statistics (data.cpp:451-484) from the tile layout: block = 4 waves = 4 marker groups; lane (r, g) walks the
// pieces P(q_local = r, J_local = 4i + g) of its group, transposes them back to one word per marker and counts as k_stats_stripes
__global__ __launch_bounds__(256) void k_stats_tile(const uint4* __restrict__ tiles, const uint32_t* __restrict__ mask2,
                                                    int64_t M, int64_t nrg, int64_t nkb, int64_t P4, double nonas,
                                                    double alpha_scale, double* __restrict__ mave, double* __restrict__ msig,
                                                    uint32_t* __restrict__ counts) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int64_t rg = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (rg >= nrg) return;
    // counts by inclusion-exclusion as in k_stats_stripes: nh (high bit: a = 2 or missing), nl (low bit: a = 1 or missing), nb (both:
    // missing) among the individuals with a phenotype, np = those individuals in this lane's K slices
    uint32_t nh[4] = {0, 0, 0, 0}, nl[4] = {0, 0, 0, 0}, nb[4] = {0, 0, 0, 0}, np = 0;
    typedef unsigned int v4u __attribute__((ext_vector_type(4)));
    const v4u* src = reinterpret_cast<const v4u*>(tiles) + rg * nkb * 256 + r * 4 + g;
    auto count = [&](const v4u (&v)[4], int64_t kb) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int64_t J = kb * 16 + 4 * i + g;
            const uint32_t pm = (J < P4) ? mask2[J] : 0u;       // both bits of a pair are set for an individual with a phenotype
            np += __popc(pm & 0x55555555u);
            uint32_t T[4];
            transpose4x4_bytes(v[i].x, v[i].y, v[i].z, v[i].w, T);   // T[t] = marker 4q + t, 16 individuals of group J
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const uint32_t wm = T[t] & pm;
                nh[t] += __popc(wm & 0xAAAAAAAAu);
                nl[t] += __popc(wm & 0x55555555u);
                nb[t] += __popc(wm & (wm >> 1) & 0x55555555u);
            }
        }
    };
    int64_t kb = 0;
    for (; kb + 2 <= nkb; kb += 2) {       // two super-blocks (8 KiB) of non-temporal loads in flight per wave
        v4u a[4], b[4];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = __builtin_nontemporal_load(src + kb * 256 + i * 64);
#pragma unroll
        for (int i = 0; i < 4; i++) b[i] = __builtin_nontemporal_load(src + (kb + 1) * 256 + i * 64);
        count(a, kb);
        count(b, kb + 1);
    }
    for (; kb < nkb; kb++) {
        v4u a[4];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = __builtin_nontemporal_load(src + kb * 256 + i * 64);
        count(a, kb);
    }
    uint32_t cp = np;
    cp += __shfl_xor(cp, 16, 64); cp += __shfl_xor(cp, 32, 64);
    uint32_t n2[4], n1[4], n0[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        uint32_t ch = nh[t], cl = nl[t], cb = nb[t];
        ch += __shfl_xor(ch, 16, 64); ch += __shfl_xor(ch, 32, 64);
        cl += __shfl_xor(cl, 16, 64); cl += __shfl_xor(cl, 32, 64);
        cb += __shfl_xor(cb, 16, 64); cb += __shfl_xor(cb, 32, 64);
        n2[t] = ch - cb; n1[t] = cl - cb; n0[t] = cp - ch - cl + cb;
    }
    if (g != 0) return;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t m = rg * 64 + 4 * r + t;
        if (m >= M) continue;
        double suma = 2.0 * n2[t] + 1.0 * n1[t], sumb = (double)n0[t] + (double)n1[t] + (double)n2[t];
        double mu = (sumb != 0) ? suma / sumb : 0.0;
        double sumsqr = n2[t] * ((2.0 - mu) * (2.0 - mu)) + n1[t] * ((1.0 - mu) * (1.0 - mu)) + n0[t] * (mu * mu);
        double sg;
        if (sumsqr != 0) {
            if (alpha_scale == 1.0) sg = 1.0 / sqrt(sumsqr / (nonas - 1.0));
            else sg = 1.0 / pow(sqrt(sumsqr / (nonas - 1.0)), alpha_scale);
        } else
            sg = 1.0;
        mave[m] = mu;
        msig[m] = sg;
        if (counts) {
            counts[3 * m] = n2[t];
            counts[3 * m + 1] = n1[t];
            counts[3 * m + 2] = n0[t];
        }
    }
}

// digits of an M-vector in the B-operand order of the tile layout's Ax side: K-step = 64 markers; entry
// k = 64 kstep + 16 i + 4 gam + t is byte t of dword i of the 16-byte element (kstep, gam, column): dword index
// kstep * 256 + (gam * 16 + col0 + c) * 4 + i.  Thread = (kstep, i, gam): four consecutive entries.
template <bool WIDE>   // WIDE: head digits in columns col0 .., residual digits in col0 + 8 .. (k_quant)
__global__ __launch_bounds__(256) void k_quant_t(QuantArgs a, int64_t n, int64_t nsteps) {
    const double* __restrict__ v = a.v[blockIdx.y];
    uint32_t* __restrict__ out = a.out[blockIdx.y];
    const int col0 = a.col0[blockIdx.y];
    const double mult = quant_scale(a);
    const int64_t tid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tid >= nsteps * 16) return;
    const int gam = tid & 3, i = (tid >> 2) & 3;
    const int64_t kstep = tid >> 4;
    const double mult2 = WIDE ? wide_mult2(mult) : 0.0, sinv = (WIDE && mult > 0.0) ? 1.0 / mult : 0.0;
    uint32_t dig[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dig2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int64_t k = 4 * tid + t;                       // = 64 kstep + 16 i + 4 gam + t
        double val = (k < n) ? v[k] : 0.0;
        const double qd = rint(val * mult);
        digits_of((long long)qd, t, dig);
        if (WIDE && mult2 > 0.0) digits_of((long long)rint(fma(-qd, sinv, val) * mult2), t, dig2);
    }
    uint32_t* o = out + kstep * 256 + (gam * 16 + col0) * 4 + i;
#pragma unroll
    for (int c = 0; c < 8; c++) o[c * 4] = dig[c];
    if (WIDE) {
#pragma unroll
        for (int c = 0; c < 8; c++) o[(8 + c) * 4] = dig2[c];
    }
}

__device__ __forceinline__ void combine(const long long (&s)[7], long long& hi, long long& lo);

// Digit sums of P consecutive planes (p0 ...) of one row over its np pieces.  Four pieces are fetched before any is added: the
// loads of a piece do not depend on the previous one, but a loop of unknown length waits for each piece in turn -- 25 round trips
// for a row of the hybrid decomposition's remainder.  (DEEP = false: the plain loop -- k_fin_atx_dot, whose many one-piece rows lost
// more to the registers of the prefetch than its few long rows gained: 17 -> 20 us at M = 200k; it starts its long rows first instead.)
template <int P, bool DEEP = true>
__device__ __forceinline__ void gather_pieces(const int32_t* __restrict__ partial, int np, int ppk, int p0, int64_t rows_p, int64_t row,
                                              long long (&s)[P][7]) {
#pragma unroll
    for (int pl = 0; pl < P; pl++)
#pragma unroll
        for (int c = 0; c < 7; c++) s[pl][c] = 0;
    const int64_t pstride = (int64_t)ppk * rows_p * 8;                       // int32 from one piece to the next
    const int32_t* base = partial + ((int64_t)p0 * rows_p + row) * 8;
    int ks = 0;
    for (; DEEP && ks + 4 <= np; ks += 4) {
        int4 a[4][P], b[4][P];
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int pl = 0; pl < P; pl++) {
                const int4* q = reinterpret_cast<const int4*>(base + (int64_t)(ks + j) * pstride + (int64_t)pl * rows_p * 8);
                a[j][pl] = q[0];
                b[j][pl] = q[1];
            }
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int pl = 0; pl < P; pl++) {
                s[pl][0] += a[j][pl].x; s[pl][1] += a[j][pl].y; s[pl][2] += a[j][pl].z; s[pl][3] += a[j][pl].w;
                s[pl][4] += b[j][pl].x; s[pl][5] += b[j][pl].y; s[pl][6] += b[j][pl].z;
            }
    }
    for (; ks < np; ks++)
#pragma unroll
        for (int pl = 0; pl < P; pl++) {
            const int4* q = reinterpret_cast<const int4*>(base + (int64_t)ks * pstride + (int64_t)pl * rows_p * 8);
            const int4 x0 = q[0], x1 = q[1];
            s[pl][0] += x0.x; s[pl][1] += x0.y; s[pl][2] += x0.z; s[pl][3] += x0.w;
            s[pl][4] += x1.x; s[pl][5] += x1.y; s[pl][6] += x1.z;
        }
}

// The p-value pass (data::pvals_calc, data.cpp:1150-1226; pvals_calc_LOCO, :1290-1353): ONE pass over stripes_m for the two
// N-vectors p = y_mod and p^2 (MODE 2), and in its epilogue, per marker, the four raw sums
//   {sum_n a p, sum_n b p, sum_n a p^2, sum_n b p^2}     (a, b of dotp_lut.hpp; no mean / scale / 1/sqrt(N))
// recombined from the digit planes and handed straight to the regression test (gvp::marker_pval) -- the sums never go through
// memory (32 MB written and read back at M = 1M when the test was a launch of its own).
struct FinPvals { const uint32_t* cnt; const double* mave; const double* msig; const double* xself; double self_scale;
                  const int* chrom; int ch; double* pvals; };
__global__ __launch_bounds__(128) void k_fin_pvals(const int32_t* __restrict__ partial, int ksplit, int64_t rows_p, int64_t M,
                                                   const double* __restrict__ scal1, const double* __restrict__ scal2, FinPvals a,
                                                   int64_t nkb, int64_t skL, int64_t piv) {
    const int64_t m = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (m >= M) return;
    if (a.chrom && a.chrom[m] != a.ch) return;          // LOCO: only the markers of chromosome ch are tested and written
    long long s[4][7];
    gather_pieces<4>(partial, pieces_of(m, ksplit, nkb, skL, 8, piv), 4, 0, rows_p, m, s);
    double s4[4];
#pragma unroll
    for (int v = 0; v < 2; v++) {
        long long xh, xl, yh, yl;
        combine(s[2 * v], xh, xl);
        combine(s[2 * v + 1], yh, yl);
        const double* sc = v ? scal2 : scal1;
        const double sa = ((double)(xh - 3 * yh) * 4294967296.0 + (double)(xl - 3 * yl)) * sc[3];
        const double sm = ((double)yh * 4294967296.0 + (double)yl) * sc[3];
        s4[2 * v] = sa;
        s4[2 * v + 1] = sc[1] - sm;
    }
    const double cself = a.xself ? a.xself[m] * a.self_scale : 0.0;
    a.pvals[m] = gvp::marker_pval(a.cnt[3 * m], a.cnt[3 * m + 1], a.cnt[3 * m + 2], a.mave[m], a.msig[m], s4, cself);
}

// exact digit recombination: sum_c s_c 256^c as (hi, lo) with value = hi * 2^32 + lo
__device__ __forceinline__ void combine(const long long (&s)[7], long long& hi, long long& lo) {
    lo = s[0] + (s[1] << 8) + (s[2] << 16) + (s[3] << 24);
    hi = s[4] + (s[5] << 8) + (s[6] << 16);
}

// data::ATx epilogue (data.cpp:779, :825-832): out[m] = msig (sum a p - mave sum b p) / sqrt(N),
// sum a p = (X - 3Y) scale, sum b p = P - Y scale.
struct FinAtx { double* out[2]; const double* addx[2]; };   // addx != NULL: out = tau * ATx + gam2 * addx (lmmse_mult, vamp.cpp:1112-1116)
__global__ __launch_bounds__(256) void k_fin_atx(const int32_t* __restrict__ partial, int ksplit, int64_t rows_p, int64_t M,
                                                 const double* __restrict__ scal_base, const double* __restrict__ mave,
                                                 const double* __restrict__ msig, double inv_sqrt_n, FinAtx a, double tau,
                                                 double gam2, int ppk, int64_t nkb, int64_t skL, int64_t piv) {
    // ppk = planes per K-split in `partial` (2, or 4 for the two-vector kernels); vector v = blockIdx.y owns planes 2v, 2v+1
    const int v = blockIdx.y, p0 = 2 * v;
    const double* __restrict__ scal = scal_base + 4 * v;
    double* __restrict__ out = a.out[v];
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    long long sxy[2][7];
    gather_pieces<2>(partial, pieces_of(m, ksplit, nkb, skL, 8, piv), ppk, p0, rows_p, m, sxy);
    long long xh, xl, yh, yl;
    combine(sxy[0], xh, xl);
    combine(sxy[1], yh, yl);
    const double scale = scal[3], P = scal[1];
    const double sa = ((double)(xh - 3 * yh) * 4294967296.0 + (double)(xl - 3 * yl)) * scale;
    const double sm = ((double)yh * 4294967296.0 + (double)yl) * scale;
    const double r = msig[m] * fma(-mave[m], P - sm, sa) * inv_sqrt_n;
    out[m] = a.addx[v] ? fma(tau, r, gam2 * a.addx[v][m]) : r;   // (explicit: k_fin_atx_dot rounds the same way)
}

// The same epilogue as one half of a device-resident CG step: out = d = tau ATx + gam2 p (addx = p is the search direction)
// and, in the same sweep, the block partials of <d, p> in the summation order of gvk::dots (grid-stride, block tree);
// gvk::finalize adds them up in the fixed order of k_finalize and leaves <d, p> in dot_out[v][0].
// A slot whose system has finished is skipped (its d is not needed and its scalars must not move).
struct FinAtxDot { double* out[2]; const double* addx[2]; const double* st[2]; double* part[2]; double* dot_out[2]; int pv[2]; };
__global__ __launch_bounds__(256) void k_fin_atx_dot(const int32_t* __restrict__ partial, int ksplit, int64_t rows_p, int64_t M,
                                                     const double* __restrict__ scal_base, const double* __restrict__ mave,
                                                     const double* __restrict__ msig, double inv_sqrt_n, FinAtxDot a, double tau,
                                                     double gam2, int ppk, int64_t nkb, int64_t skL, int64_t piv) {
    __shared__ double sh[4];
    const int v = blockIdx.y, p0 = 2 * a.pv[v];
    const double* st = a.st[v];
    if (st && st[gvm::ST_ACTIVE] == 0.0) return;
    const double* __restrict__ scal = scal_base + 4 * a.pv[v];
    double* __restrict__ out = a.out[v];
    const double* __restrict__ addx = a.addx[v];
    const double scale = scal[3], P = scal[1];
    double s = 0.0;
    // Block b of the reduction is run by the workgroup dispatched (gridDim.x - 1 - b)-th: the rows of the hybrid decomposition's
    // remainder -- the LAST rows, up to 25 pieces each where the others have one -- start first instead of last (the partial sums
    // keep their slots, so nothing changes for whoever adds them up)
    const int64_t bx = (int64_t)gridDim.x - 1 - blockIdx.x;
    for (int64_t m = bx * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
        long long sxy[2][7];
        gather_pieces<2, false>(partial, pieces_of(m, ksplit, nkb, skL, 8, piv), ppk, p0, rows_p, m, sxy);
        long long xh, xl, yh, yl;
        combine(sxy[0], xh, xl);
        combine(sxy[1], yh, yl);
        const double sa = ((double)(xh - 3 * yh) * 4294967296.0 + (double)(xl - 3 * yl)) * scale;
        const double sm = ((double)yh * 4294967296.0 + (double)yl) * scale;
        const double r = msig[m] * fma(-mave[m], P - sm, sa) * inv_sqrt_n;
        if (addx) {                                  // a CG slot: d = tau r + gam2 p and <d, p>
            const double pm = addx[m];
            const double d = fma(tau, r, gam2 * pm);
            out[m] = d;
            s = fma(d, pm, s);
        } else
            out[m] = r;                              // a plain product sharing the pass (gv_cg_solve_aat2: A^T p of the N-space system)
    }
    s = wave_sum_d(s);   // the butterfly of gvk::dots (bit-identical sums)
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0 && a.part[v]) a.part[v][bx] = sh[0] + sh[1] + sh[2] + sh[3];
    // (the block partials are added up by gvk::finalize, launched behind this kernel: a last-block ticket would serialise
    // ~1000 atomics on one address, 50 ns each -- measured 74 us for this kernel at M = 500k against 16 us without)
}

// data::Ax epilogue (data.cpp:972, :998-1005): out[n] = mask (T scale - K0) * post, post = 1/sqrt(N) or 1 (multi-rank)
// ride / st / ride_out (gvm::CgHook::ride_out): the slot the rider took on the way in (k_prep_ax, same test on the same flags)
// delivers its product straight into ride_out
// dq_p / dq_part (gvm::CgHook::dq_p): slot v closes an application of tau A A^T + gam2 I to p (gv_cg_solve_aat2w, one rank): while
// its system is running, out = tau * product + gam2 * p and the block partials of <out, p> -- what gvk::aat_step's k_aat_dq does in
// a launch of its own, bit for bit (one entry per thread there too when the vector fits RED_BLOCKS blocks, which the caller checks)
struct FinAx { double* out[2]; const int* ride; const double* st[2]; double* ride_out; const double* dq_p[2]; double* dq_part[2];
               double tau, gam2; };
__global__ __launch_bounds__(256) void k_fin_ax(const int32_t* __restrict__ partial, int ksplit, int64_t rows_p,
                                                int64_t npad, const double* __restrict__ scal_base,
                                                const uint32_t* __restrict__ mask2, double post, FinAx a, int ppk,
                                                int64_t nkb, int64_t skL, int qshift, int64_t piv) {
    // the streaming kernel has already added the r'.c and miss.e products: one plane per vector and piece, ppk = number of
    // vectors of the pass (1: MODE 1, 4; 2: MODE 3)
    __shared__ double sh[4];
    const int v = blockIdx.y;
    const double* __restrict__ scal = scal_base + 4 * v;
    double* __restrict__ out = a.out[v];
    const bool sys_done = a.st[v] && a.st[v][gvm::ST_ACTIVE] == 0.0;
    if (a.ride_out && a.ride && *a.ride == 1 && sys_done && a.st[1 - v] && a.st[1 - v][gvm::ST_ACTIVE] != 0.0) out = a.ride_out;
    const bool dq = a.dq_p[v] && !sys_done;          // (uniform over the launch slice)
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double val = 0.0;
    if (n < npad) {
        const uint32_t present = (mask2[n >> 4] >> (2 * (n & 15))) & 1u;
        if (present && n < rows_p) {
            long long sx[1][7];
            gather_pieces<1>(partial, pieces_of(n, ksplit, nkb, skL, qshift, piv), ppk, v, rows_p, n, sx);
            long long xh, xl;
            combine(sx[0], xh, xl);
            const double T = ((double)xh * 4294967296.0 + (double)xl) * scal[3];
            val = (T - scal[1]) * post;
        }
    }
    if (!dq) {
        if (n < npad) out[n] = val;
        return;
    }
    double s = 0.0;
    if (n < npad) {
        const double pi = a.dq_p[v][n];
        const double di = fma(a.tau, val, a.gam2 * pi);
        out[n] = di;
        s = fma(di, pi, s);
    }
    s = wave_sum_d(s);   // gvk's block_sum_256: wave butterflies, then the four wave sums in order
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) a.dq_part[v][blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}

// ---- epilogues of the two-level passes (kernel mode 2): planes of the HEAD slot and of the RESIDUAL slot, each an exact integer,
// each converted with one rounding and scaled by its own power of two; the second is 2^-54 of the first.
__device__ __forceinline__ double two_level(const long long (&h)[7], const long long (&r)[7], double s1) {
    long long hh, hl, rh, rl;
    combine(h, hh, hl);
    combine(r, rh, rl);
    const double head = ((double)hh * 4294967296.0 + (double)hl) * s1;
    const double res = ((double)rh * 4294967296.0 + (double)rl) * (s1 * 0x1p-54);
    return head + res;
}
// data::Ax: out[n] = mask (sum_i a'_ni c_i - sum_i b_ni mave_i c_i) post; planes 0 / 1 of every piece = head / residual
__global__ __launch_bounds__(256) void k_fin_ax_wide(const int32_t* __restrict__ partial, int ksplit, int64_t rows_p, int64_t npad,
                                                     const double* __restrict__ scal, const uint32_t* __restrict__ mask2, double post,
                                                     double* __restrict__ out, int64_t nkb, int64_t skL, int qshift, int64_t piv) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= npad) return;
    double val = 0.0;
    const uint32_t present = (mask2[n >> 4] >> (2 * (n & 15))) & 1u;
    if (present && n < rows_p) {
        long long sx[2][7];
        gather_pieces<2>(partial, pieces_of(n, ksplit, nkb, skL, qshift, piv), 2, 0, rows_p, n, sx);
        val = two_level(sx[0], sx[1], scal[3]) * post;
    }
    out[n] = val;
}
// data::ATx: out[m] = msig (sum a p - mave sum b p) / sqrt(N); planes 0..3 of every piece = a'.head, b.head, a'.residual, b.residual
__global__ __launch_bounds__(256) void k_fin_atx_wide(const int32_t* __restrict__ partial, int ksplit, int64_t rows_p, int64_t M,
                                                      const double* __restrict__ scal, const double* __restrict__ mave,
                                                      const double* __restrict__ msig, double inv_sqrt_n, double* __restrict__ out,
                                                      const double* __restrict__ addx, double tau, double gam2, int64_t nkb, int64_t skL,
                                                      int64_t piv) {
    const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    long long sx[4][7];
    gather_pieces<4>(partial, pieces_of(m, ksplit, nkb, skL, 8, piv), 4, 0, rows_p, m, sx);
    const double sa = two_level(sx[0], sx[2], scal[3]), sb = two_level(sx[1], sx[3], scal[3]);
    const double r = msig[m] * fma(-mave[m], sb, sa) * inv_sqrt_n;
    out[m] = addx ? fma(tau, r, gam2 * addx[m]) : r;
}

inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }

// launch of the streaming kernel of one matvec (HIP events around it when the roofline timing is on)
static KBounds make_bounds(const gvm::Decomp& d, int64_t nkb);
// the streaming kernel on the tile layout (DIR 0: ATx side, MODE 0 / 2; DIR 1: Ax side, MODE 3 / 4, nv vectors)
template <int DIR, int MODE>
void launch_tile(hipStream_t s, const gvm::Plan& pl, const void* dig0, const void* dig1, int64_t nrg, int64_t nkb,
                 const gvm::Decomp& d, int nv, const int* go) {
    if (pl.ev0) (void)hipEventRecord(pl.ev0, s);
    const int64_t nq = (nrg + 3) / 4;
    const int64_t grid = gvm::grid_of(d, nq, nkb);
    if (grid <= 0) return;          // an empty shard: nothing to stream (a zero-size grid is an invalid launch)
    const KBounds kb = make_bounds(d, nkb);
    const int64_t rstride = pl.rstride_n > 0 ? pl.rstride_n : nrg;
    const unsigned lds_pad = gvm::lds_pad_of(d);     // Decomp::occ: two workgroups per CU instead of three
#define GV_LAUNCH_T(SKV, GOV)                                                                                              \
    hipLaunchKernelGGL((k_mfma_tile<DIR, MODE, SKV, GOV>), dim3((unsigned)grid), dim3(256), lds_pad, s, (const u32x4*)pl.tiles,     \
                       (const u32x4*)dig0, (const u32x4*)dig1, nrg, nkb, d.ks, d.skL, d.prio, kb, pl.partial, nv, go, rstride, \
                       (uint32_t)gvm::piv_of(d, nq))
    if (MODE >= 4 || !go) {             // (the people-statistics plane and the two-level passes are never part of a device-resident CG step)
        if (d.skL > 0) GV_LAUNCH_T(true, false); else GV_LAUNCH_T(false, false);
    } else if constexpr (MODE < 4) {
        if (d.skL > 0) GV_LAUNCH_T(true, true); else GV_LAUNCH_T(false, true);
    }
#undef GV_LAUNCH_T
    if (pl.ev1) (void)hipEventRecord(pl.ev1, s);
}
static KBounds make_bounds(const gvm::Decomp& d, int64_t nkb) {
    KBounds kb{};
    if (d.skL <= 0) {
        const int ks = d.ks < 1 ? 1 : (d.ks > GV_MAX_KS ? GV_MAX_KS : d.ks);
        // cumulative segment lengths, every segment at least one K-block (ks <= nkb): geometric (big first), tapered, or equal;
        // then, per quad parity c, the segments whose workgroup has an ODD block index ((c + j) odd: the faster XCDs) stretched by
        // 1 + xskew, the others shrunk by 1 - xskew
        double w[GV_MAX_KS];
        for (int j = 0; j < ks; j++) {
            if (d.geo > 0.f && ks > 1) w[j] = j ? w[j - 1] * (double)d.geo : 1.0;
            else w[j] = ks > 1 ? 1.0 + (double)d.taper * (double)(ks - 1 - 2 * j) / (double)(ks - 1) : 1.0;
        }
        for (int c = 0; c < 2; c++) {
            double tot = 0.0, acc = 0.0, v[GV_MAX_KS];
            for (int j = 0; j < ks; j++) {
                v[j] = w[j] * (((c + j) & 1) ? 1.0 + (double)d.xskew : 1.0 - (double)d.xskew);      // (c + j) odd <-> odd block index
                tot += v[j];
            }
            kb.b[c][0] = 0;
            for (int j = 0; j < ks; j++) {
                acc += v[j];
                int64_t e = (int64_t)((double)nkb * acc / tot + 0.5);
                const int64_t lo = (int64_t)kb.b[c][j] + 1, hi = nkb - (ks - 1 - j);
                e = e < lo ? lo : (e > hi ? hi : e);
                kb.b[c][j + 1] = (uint32_t)e;
            }
            kb.b[c][ks] = (uint32_t)nkb;
        }
    }
    return kb;
}

template <int MODE>
void launch_stream(hipStream_t s, const gvm::Plan& pl, const void* stripes, const void* dig0, const void* dig1, int64_t nrg,
                   int64_t nkb, const gvm::Decomp& d, const int* go = nullptr) {
    if (pl.layout == 1) {   // one resident layout: the same products on the tile kernels
        if (MODE == 0) return launch_tile<0, 0>(s, pl, dig0, dig1, nrg, nkb, d, 1, go);
        if (MODE == 2) return launch_tile<0, 2>(s, pl, dig0, dig1, nrg, nkb, d, 2, go);
        if (MODE == 1) return launch_tile<1, 3>(s, pl, dig0, dig1, nrg, nkb, d, 1, go);
        if (MODE == 3) return launch_tile<1, 3>(s, pl, dig0, dig1, nrg, nkb, d, 2, go);
        if (MODE == 5) return launch_tile<0, 5>(s, pl, dig0, dig1, nrg, nkb, d, 2, go);
        if (MODE == 6) return launch_tile<1, 6>(s, pl, dig0, dig1, nrg, nkb, d, 2, go);
        return launch_tile<1, 4>(s, pl, dig0, dig1, nrg, nkb, d, 1, go);
    }
    if (pl.ev0) (void)hipEventRecord(pl.ev0, s);
    const int64_t nq = (nrg + 3) / 4;
    const int64_t grid = gvm::grid_of(d, nq, nkb);
    if (grid <= 0) return;          // an empty shard: nothing to stream (a zero-size grid is an invalid launch)
    const KBounds kb = make_bounds(d, nkb);
    const unsigned lds_pad = gvm::lds_pad_of(d);
#define GV_LAUNCH_MV(SKV, GOV)                                                                                             \
    hipLaunchKernelGGL((k_mfma_matvec<MODE, SKV, GOV>), dim3((unsigned)grid), dim3(256), lds_pad, s, (const u32x4*)stripes,   \
                       (const u32x4*)dig0, (const u32x4*)dig1, nrg, nkb, d.ks, d.skL, d.prio, kb, pl.partial, go,        \
                       (uint32_t)gvm::piv_of(d, nq))
    if (MODE >= 4 || !go) {            // (the people-statistics plane and the two-level passes are never part of a device-resident CG step)
        if (d.skL > 0) GV_LAUNCH_MV(true, false); else GV_LAUNCH_MV(false, false);
    } else if constexpr (MODE < 4) {
        if (d.skL > 0) GV_LAUNCH_MV(true, true); else GV_LAUNCH_MV(false, true);
    }
#undef GV_LAUNCH_MV
    if (pl.ev1) (void)hipEventRecord(pl.ev1, s);
}

}  // namespace

namespace gvm {

void stripes_m_chunk(hipStream_t s, const uint8_t* raw, int64_t pitch, int64_t mc, int64_t N, void* stripes,
                     int64_t rg0, int64_t nkb) {
    if (mc <= 0) return;
    hipLaunchKernelGGL(k_stripes_m, dim3((unsigned)nkb, (unsigned)((mc + 63) / 64)), dim3(256), 0, s, raw, pitch, mc, N,
                       (uint4*)stripes, rg0, nkb);
}
void stripes_n_chunk(hipStream_t s, const uint8_t* raw, int64_t pitch, int64_t mc, int64_t N, void* stripes,
                     int64_t kb0, int64_t nkb, int64_t nrg_n) {
    if (mc <= 0) return;
    hipLaunchKernelGGL(k_stripes_n, dim3((unsigned)((mc + 255) / 256), (unsigned)nrg_n), dim3(256), 0, s, raw, pitch, mc,
                       N, (uint4*)stripes, kb0, nkb);
}
void stats_from_stripes(hipStream_t s, const void* stripes_m, const uint32_t* mask2, int64_t M, int64_t nkb,
                        int64_t P4, double nonas, double alpha_scale, double* mave, double* msig, uint32_t* counts) {
    if (M <= 0) return;
    // K-blocks in flight per wave: 4 (16 KiB) measured best on the 100 GB shard (GV_STATS_UNROLL = 2 / 3 / 4: development)
    static const int unr = getenv("GV_STATS_UNROLL") ? atoi(getenv("GV_STATS_UNROLL")) : 4;
#define GV_STATS(U) hipLaunchKernelGGL(k_stats_stripes<U>, dim3(nblk((M + 63) / 64, 4)), dim3(256), 0, s, (const uint4*)stripes_m, mask2, M, \
                                       nkb, P4, nonas, alpha_scale, mave, msig, counts)
    if (unr == 2) GV_STATS(2); else if (unr == 3) GV_STATS(3); else GV_STATS(4);
#undef GV_STATS
}

// Blocks of a prep launch (every block of the quantisation launch behind it adds their partials up: prep_scalars)
constexpr int PREP_BLOCKS = 256;
static int prep_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > PREP_BLOCKS ? PREP_BLOCKS : b));
}

// vector preparation of one (nv = 1) or two (nv = 2) N-vectors: scal[4v..] and the digit columns 8v.. of dig0
static void prep_quant_atx(hipStream_t s, const Plan& pl, const double* pa, const double* pb, int64_t npad, double* red_partial,
                           const CgHook* cg = nullptr) {
    const int nv = pb ? 2 : 1, nb = prep_blocks(npad);
    PrepAtx pa_{{pa, pb}, cg ? cg->go : nullptr, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    if (cg)
        for (int v = 0; v < nv; v++)
            if (cg->pn[v]) { pa_.st[v] = cg->state[v]; pa_.pw[v] = cg->pn[v]; pa_.z[v] = cg->zn[v]; }
    hipLaunchKernelGGL(k_prep_atx, dim3(nb, nv), dim3(256), 0, s, pa_, npad, red_partial);
    QuantArgs q{};
    q.nblocks = nb;
    q.v[0] = pa; q.scal[0] = pl.scal; q.out[0] = (uint32_t*)pl.dig0; q.col0[0] = 0; q.part[0] = red_partial; q.wr[0] = 1;
    q.v[1] = pb; q.scal[1] = pl.scal + 4; q.out[1] = (uint32_t*)pl.dig0; q.col0[1] = 8; q.part[1] = red_partial + PREP_STRIDE; q.wr[1] = 1;
    hipLaunchKernelGGL(k_quant<false>, dim3(nblk(pl.nkb_m * 64, 256), nv), dim3(256), 0, s, q, npad, pl.nkb_m, nv == 2 ? 16 : 8);
}

// red_blocks of gvk::dots: the block partials of the fused <d, p> are taken in the same order
static int dot_blocks(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > RED_BLOCKS ? RED_BLOCKS : b));
}
static void fin_atx_cg(hipStream_t s, const Plan& pl, const Decomp& d, int nv, const double* mave, const double* msig,
                       double inv_sqrt_n, double* const* out, const double* const* addx, double tau, double gam2,
                       const CgHook& cg) {
    FinAtxDot f{};
    for (int v = 0; v < nv; v++) {
        f.out[v] = out[v]; f.addx[v] = addx[v]; f.st[v] = cg.state[v]; f.part[v] = cg.dot_part[v]; f.dot_out[v] = cg.dot_out[v];
        f.pv[v] = v;
    }
    const int nb = dot_blocks(pl.M);
    hipLaunchKernelGGL(k_fin_atx_dot, dim3(nb, nv), dim3(256), 0, s, pl.partial, d.ks, pl.nrg_m * 64, pl.M,
                       pl.scal, mave, msig, inv_sqrt_n, f, tau, gam2, nv == 2 ? 4 : 2, pl.nkb_m, d.skL, piv_of(d, (pl.nrg_m + 3) / 4));
    if (!cg.dot_self)
        for (int v = 0; v < nv; v++)
            if (cg.dot_out[v]) gvk::finalize(s, cg.dot_part[v], nb, 1, cg.dot_out[v]);
}
// blocks of the <d, p> partials k_fin_atx_dot leaves (for a consumer that adds them up itself: CgHook::dot_self)
int atx_dot_blocks(const Plan& pl) { return dot_blocks(pl.M); }

void atx(hipStream_t s, const Plan& pl, const double* p, int64_t npad, const double* mave, const double* msig,
         double inv_sqrt_n, double* red_partial, double* out, const double* addx, double tau, double gam2, const CgHook* cg) {
    prep_quant_atx(s, pl, p, nullptr, npad, red_partial, cg);
    launch_stream<0>(s, pl, pl.stripes_m, pl.dig0, nullptr, pl.nrg_m, pl.nkb_m, pl.dm[0], cg ? cg->go : nullptr);
    if (cg && cg->dot_out[0]) {
        double* o[2] = {out, nullptr};
        const double* ax_[2] = {addx, nullptr};
        fin_atx_cg(s, pl, pl.dm[0], 1, mave, msig, inv_sqrt_n, o, ax_, tau, gam2, *cg);
        return;
    }
    FinAtx f{{out, nullptr}, {addx, nullptr}};
    hipLaunchKernelGGL(k_fin_atx, dim3(nblk(pl.M, 256), 1), dim3(256), 0, s, pl.partial, pl.dm[0].ks, pl.nrg_m * 64, pl.M,
                       pl.scal, mave, msig, inv_sqrt_n, f, tau, gam2, 2, pl.nkb_m, pl.dm[0].skL, piv_of(pl.dm[0], (pl.nrg_m + 3) / 4));
}

// data::ATx of TWO N-vectors in one pass over stripes_m
void atx2(hipStream_t s, const Plan& pl, const double* pa, const double* pb, int64_t npad, const double* mave,
          const double* msig, double inv_sqrt_n, double* red_partial, double* outa, double* outb, const double* addxa,
          const double* addxb, double tau, double gam2, const CgHook* cg) {
    prep_quant_atx(s, pl, pa, pb, npad, red_partial, cg);
    launch_stream<2>(s, pl, pl.stripes_m, pl.dig0, nullptr, pl.nrg_m, pl.nkb_m, pl.dm[1], cg ? cg->go : nullptr);
    if (cg && (cg->dot_out[0] || cg->dot_out[1])) {
        double* o[2] = {outa, outb};
        const double* ax_[2] = {addxa, addxb};
        fin_atx_cg(s, pl, pl.dm[1], 2, mave, msig, inv_sqrt_n, o, ax_, tau, gam2, *cg);
        return;
    }
    FinAtx f{{outa, outb}, {addxa, addxb}};
    hipLaunchKernelGGL(k_fin_atx, dim3(nblk(pl.M, 256), 2), dim3(256), 0, s, pl.partial, pl.dm[1].ks, pl.nrg_m * 64, pl.M,
                       pl.scal, mave, msig, inv_sqrt_n, f, tau, gam2, 4, pl.nkb_m, pl.dm[1].skL, piv_of(pl.dm[1], (pl.nrg_m + 3) / 4));
}

// the p-value pass: operands (k_prep_pv), digits, ONE pass over stripes_m, per-marker test in the epilogue (k_fin_pvals)
void marker_pvals(hipStream_t s, const Plan& pl, const double* y, const double* z1, const double* add, const uint32_t* mask2,
                  int64_t npad, const double* mave, const double* msig, double* pa, double* pb, double* red_partial, const PvArgs& a,
                  double* pvals) {
    const int nb = prep_blocks(npad);
    hipLaunchKernelGGL(k_prep_pv, dim3(nb), dim3(256), 0, s, y, z1, add, mask2, npad, pa, pb, red_partial);
    QuantArgs q{};
    q.nblocks = nb;
    q.v[0] = pa; q.scal[0] = pl.scal; q.out[0] = (uint32_t*)pl.dig0; q.col0[0] = 0; q.part[0] = red_partial; q.wr[0] = 1;
    q.v[1] = pb; q.scal[1] = pl.scal + 4; q.out[1] = (uint32_t*)pl.dig0; q.col0[1] = 8; q.part[1] = red_partial + PREP_STRIDE; q.wr[1] = 1;
    hipLaunchKernelGGL(k_quant<false>, dim3(nblk(pl.nkb_m * 64, 256), 2), dim3(256), 0, s, q, npad, pl.nkb_m, 16);
    {
        gvm::Plan q2 = pl;      // no roofline events around the p-value pass
        q2.ev0 = q2.ev1 = nullptr;
        launch_stream<2>(s, q2, pl.stripes_m, pl.dig0, nullptr, pl.nrg_m, pl.nkb_m, pl.dm[1]);
    }
    FinPvals f{a.cnt, mave, msig, a.xself, a.self_scale, a.chrom, a.ch, pvals};
    hipLaunchKernelGGL(k_fin_pvals, dim3(nblk(pl.M, 128)), dim3(128), 0, s, pl.partial, pl.dm[1].ks, pl.nrg_m * 64, pl.M, pl.scal,
                       pl.scal + 4, f, pl.nkb_m, pl.dm[1].skL, piv_of(pl.dm[1], (pl.nrg_m + 3) / 4));
}

// operands cv / ev (and cv2 / ev2) -> digit buffers of the Ax side.  Two stripe sets: one vector dig0 = [c | e], two vectors
// dig0 = [c_a | c_b], dig1 = [e_a | e_b].  Tile layout: always the second form (a one-vector pass leaves columns 8..15 unused).
static void quant_ax(hipStream_t s, const Plan& pl, int nv, const double* red_partial, int nb) {
    QuantArgs q{};
    q.nblocks = nb;
    q.part[0] = q.part[1] = red_partial;                      // slots 0, 1: c and e of vector a; 2, 3: of vector b
    q.part[2] = q.part[3] = red_partial + PREP_STRIDE;
    q.wr[0] = q.wr[2] = 1;
    if (pl.layout == 1 || nv == 2) {
        q.v[0] = pl.cv;  q.scal[0] = pl.scal;     q.out[0] = (uint32_t*)pl.dig0; q.col0[0] = 0;
        q.v[1] = pl.ev;  q.scal[1] = pl.scal;     q.out[1] = (uint32_t*)pl.dig1; q.col0[1] = 0;
        q.v[2] = pl.cv2; q.scal[2] = pl.scal + 4; q.out[2] = (uint32_t*)pl.dig0; q.col0[2] = 8;
        q.v[3] = pl.ev2; q.scal[3] = pl.scal + 4; q.out[3] = (uint32_t*)pl.dig1; q.col0[3] = 8;
        if (pl.layout == 1)
            hipLaunchKernelGGL(k_quant_t<false>, dim3(nblk(pl.nkb_n * 16, 256), 2 * nv), dim3(256), 0, s, q, pl.M, pl.nkb_n);
        else
            hipLaunchKernelGGL(k_quant<false>, dim3(nblk(pl.nkb_n * 64, 256), 4), dim3(256), 0, s, q, pl.M, pl.nkb_n, 16);
        return;
    }
    q.v[0] = pl.cv; q.scal[0] = pl.scal; q.out[0] = (uint32_t*)pl.dig0; q.col0[0] = 0;
    q.v[1] = pl.ev; q.scal[1] = pl.scal; q.out[1] = (uint32_t*)pl.dig0; q.col0[1] = 8;
    hipLaunchKernelGGL(k_quant<false>, dim3(nblk(pl.nkb_n * 64, 256), 2), dim3(256), 0, s, q, pl.M, pl.nkb_n, 16);
}
static void fin_ax(hipStream_t s, const Plan& pl, const Decomp& d, int nv, int64_t npad, const uint32_t* mask2, double post,
                   double* outa, double* outb, const CgHook* cg = nullptr, int64_t n0 = 0) {
    FinAx f{{outa, outb}, nullptr, {nullptr, nullptr}, nullptr, {nullptr, nullptr}, {nullptr, nullptr}, 0.0, 0.0};
    if (cg && cg->ride_out && nv == 2) { f.ride = cg->ride; f.st[0] = cg->state[0]; f.st[1] = cg->state[1]; f.ride_out = cg->ride_out + n0; }
    if (cg && n0 == 0)
        for (int v = 0; v < nv; v++)
            if (cg->dq_p[v]) {     // (whole-vector launches only: the block partials are those of a launch over all of npad)
                f.st[v] = cg->state[v]; f.dq_p[v] = cg->dq_p[v]; f.dq_part[v] = cg->dq_part[v]; f.tau = cg->dq_tau; f.gam2 = cg->dq_gam2;
            }
    hipLaunchKernelGGL(k_fin_ax, dim3(nblk(npad, 256), nv), dim3(256), 0, s, pl.partial, d.ks, pl.nrg_n * pl.rows_n, npad,
                       pl.scal, mask2, post, f, nv, pl.nkb_n, d.skL, pl.rows_n == 256 ? 10 : 8, piv_of(d, (pl.nrg_n + 3) / 4));
}

// ---- data::Ax in two stages: the operands of the whole vector (ax_prep), then the product for a range of row groups
// (ax_rows: stream kernel + epilogue) -- the whole range in one go, or individual-range chunks whose cross-rank all-reduce
// can run behind the next chunk (GV_OVERLAP, gv_capi.hip).  A chunk is a smaller problem of the same shape: the layouts are
// row-group-major on this side (tile layout: with the row stride of the whole shard), so it only takes pointer offsets.
void ax_prep(hipStream_t s, const Plan& pl, const double* xa, const double* xb, const double* mave, const double* msig,
             double* red_partial, const CgHook* cg) {
    const int nv = xb ? 2 : 1, nb = prep_blocks(pl.M);
    PrepAx pa{{xa, xb}, {pl.cv, pl.cv2}, {pl.ev, pl.ev2}, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}, nullptr, nullptr,
              cg ? cg->go : nullptr};
    if (cg) {
        for (int v = 0; v < nv; v++) { pa.st[v] = cg->state[v]; pa.pw[v] = cg->p[v]; pa.z[v] = cg->z[v]; }
        if (nv == 2) { pa.ride = cg->ride; pa.alt_x = cg->alt_x; }
    }
    hipLaunchKernelGGL(k_prep_ax, dim3(nb, nv), dim3(256), 0, s, pa, mave, msig, pl.M, red_partial);
    quant_ax(s, pl, nv, red_partial, nb);
}
void ax_rows(hipStream_t s, const Plan& pl, int nv, int64_t rg0, int64_t rg1, const uint32_t* mask2, int64_t npad, double post,
             double* outa, double* outb, const CgHook* cg) {
    if (rg1 <= rg0) return;
    Plan q = pl;
    const int64_t n0 = rg0 * pl.rows_n;
    int64_t nrows = (rg1 - rg0) * pl.rows_n;
    if (n0 + nrows > npad) nrows = npad - n0;
    q.nrg_n = rg1 - rg0;
    if (pl.layout == 1) {
        q.tiles = (char*)pl.tiles + (size_t)rg0 * 4096;
        q.rstride_n = pl.rstride_n > 0 ? pl.rstride_n : pl.nrg_n;
    } else
        q.stripes_n = (char*)pl.stripes_n + (size_t)rg0 * pl.nkb_n * 4096;
    const Decomp& d = nv == 2 ? pl.dn[1] : pl.dn[0];
    if (nv == 2) launch_stream<3>(s, q, q.stripes_n, pl.dig0, pl.dig1, q.nrg_n, pl.nkb_n, d, cg ? cg->go : nullptr);
    else         launch_stream<1>(s, q, q.stripes_n, pl.dig0, pl.dig1, q.nrg_n, pl.nkb_n, d, cg ? cg->go : nullptr);
    fin_ax(s, q, d, nv, nrows, mask2 + n0 / 16, post, outa + n0, outb ? outb + n0 : nullptr, cg, n0);
}

void ax(hipStream_t s, const Plan& pl, const double* x, const double* mave, const double* msig, const uint32_t* mask2,
        int64_t npad, double post, double* red_partial, double* out, const CgHook* cg) {
    ax_prep(s, pl, x, nullptr, mave, msig, red_partial, cg);
    ax_rows(s, pl, 1, 0, pl.nrg_n, mask2, npad, post, out, nullptr, cg);
}

// one of the three per-individual sums of compute_people_statistics from the Ax-side layout (k_prep_people): out[n] = mask * sum
void ax_people(hipStream_t s, const Plan& pl, int kind, const double* mave, const double* msig, const uint32_t* mask2,
               int64_t npad, double* red_partial, double* out) {
    const int nb = prep_blocks(pl.M);
    hipLaunchKernelGGL(k_prep_people, dim3(nb), dim3(256), 0, s, kind, mave, msig, pl.M, pl.cv, pl.ev, red_partial);
    quant_ax(s, pl, 1, red_partial, nb);
    gvm::Plan pq = pl;      // no roofline events around the statistics passes
    pq.ev0 = pq.ev1 = nullptr;
    if (kind == 1) launch_stream<4>(s, pq, pl.stripes_n, pl.dig0, pl.dig1, pl.nrg_n, pl.nkb_n, pl.dn[0]);
    else           launch_stream<1>(s, pq, pl.stripes_n, pl.dig0, pl.dig1, pl.nrg_n, pl.nkb_n, pl.dn[0]);
    fin_ax(s, pl, pl.dn[0], 1, npad, mask2, 1.0, out, nullptr);
}

// data::Ax of TWO M-vectors in one pass: dig0 = [c_a | c_b] (r' plane), dig1 = [e_a | e_b] (miss plane)
void ax2(hipStream_t s, const Plan& pl, const double* xa, const double* xb, const double* mave, const double* msig,
         const uint32_t* mask2, int64_t npad, double post, double* red_partial, double* outa, double* outb, const CgHook* cg) {
    ax_prep(s, pl, xa, xb, mave, msig, red_partial, cg);
    ax_rows(s, pl, 2, 0, pl.nrg_n, mask2, npad, post, outa, outb, cg);
}

// ---- kernel mode 2: one-vector products as a two-vector pass whose slots are the head and the residual of the vector ---------
void ax_wide(hipStream_t s, const Plan& pl, const double* x, const double* mave, const double* msig, const uint32_t* mask2,
             int64_t npad, double post, double* red_partial, double* out) {
    const int nb = prep_blocks(pl.M);
    hipLaunchKernelGGL(k_prep_ax_wide, dim3(nb), dim3(256), 0, s, x, mave, msig, pl.M, pl.cv, pl.ev, red_partial);
    QuantArgs q{};
    q.nblocks = nb;
    q.part[0] = q.part[1] = red_partial;                      // c and e on ONE scale
    q.wr[0] = 1;
    q.v[0] = pl.cv; q.scal[0] = pl.scal; q.out[0] = (uint32_t*)pl.dig0; q.col0[0] = 0;     // a' plane: [c head | c residual]
    q.v[1] = pl.ev; q.scal[1] = pl.scal; q.out[1] = (uint32_t*)pl.dig1; q.col0[1] = 0;     // present plane: [e head | e residual]
    if (pl.layout == 1)
        hipLaunchKernelGGL(k_quant_t<true>, dim3(nblk(pl.nkb_n * 16, 256), 2), dim3(256), 0, s, q, pl.M, pl.nkb_n);
    else
        hipLaunchKernelGGL(k_quant<true>, dim3(nblk(pl.nkb_n * 64, 256), 2), dim3(256), 0, s, q, pl.M, pl.nkb_n, 16);
    const Decomp& d = pl.dn[1];
    launch_stream<6>(s, pl, pl.stripes_n, pl.dig0, pl.dig1, pl.nrg_n, pl.nkb_n, d);
    hipLaunchKernelGGL(k_fin_ax_wide, dim3(nblk(npad, 256)), dim3(256), 0, s, pl.partial, d.ks, pl.nrg_n * pl.rows_n, npad, pl.scal, mask2,
                       post, out, pl.nkb_n, d.skL, pl.rows_n == 256 ? 10 : 8, piv_of(d, (pl.nrg_n + 3) / 4));
}
void atx_wide(hipStream_t s, const Plan& pl, const double* p, int64_t npad, const double* mave, const double* msig, double inv_sqrt_n,
              double* red_partial, double* out, const double* addx, double tau, double gam2) {
    const int nb = prep_blocks(npad);
    PrepAtx pa_{{p, nullptr}, nullptr, {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}};
    hipLaunchKernelGGL(k_prep_atx, dim3(nb, 1), dim3(256), 0, s, pa_, npad, red_partial);
    QuantArgs q{};
    q.nblocks = nb;
    q.v[0] = p; q.scal[0] = pl.scal; q.out[0] = (uint32_t*)pl.dig0; q.col0[0] = 0; q.part[0] = red_partial; q.wr[0] = 1;
    hipLaunchKernelGGL(k_quant<true>, dim3(nblk(pl.nkb_m * 64, 256), 1), dim3(256), 0, s, q, npad, pl.nkb_m, 16);
    const Decomp& d = pl.dm[1];
    launch_stream<5>(s, pl, pl.stripes_m, pl.dig0, nullptr, pl.nrg_m, pl.nkb_m, d);
    hipLaunchKernelGGL(k_fin_atx_wide, dim3(nblk(pl.M, 256)), dim3(256), 0, s, pl.partial, d.ks, pl.nrg_m * 64, pl.M, pl.scal, mave, msig,
                       inv_sqrt_n, out, addx, tau, gam2, pl.nkb_m, d.skL, piv_of(d, (pl.nrg_m + 3) / 4));
}

void tile_chunk(hipStream_t s, const uint8_t* raw, int64_t pitch, int64_t mc, int64_t N, void* tiles, int64_t rg0, int64_t nkb) {
    if (mc <= 0) return;
    hipLaunchKernelGGL(k_tile_build, dim3((unsigned)nkb, (unsigned)((mc + 63) / 64)), dim3(256), 0, s, raw, pitch, mc, N,
                       (uint4*)tiles, rg0, nkb);
}
void stats_from_tiles(hipStream_t s, const void* tiles, const uint32_t* mask2, int64_t M, int64_t nrg, int64_t nkb, int64_t P4,
                      double nonas, double alpha_scale, double* mave, double* msig, uint32_t* counts) {
    if (M <= 0) return;
    hipLaunchKernelGGL(k_stats_tile, dim3(nblk(nrg, 4)), dim3(256), 0, s, (const uint4*)tiles, mask2, M, nrg, nkb, P4, nonas,
                       alpha_scale, mave, msig, counts);
}

}  // namespace gvm
