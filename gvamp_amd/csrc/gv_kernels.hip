// gv_kernels.hip -- gfx950 kernels of libgvamp (fp64 VALU family + element-wise / reduction kernels).
//
// HBM layout of the genotype shard: marker-major rows of `pitch` bytes (pitch % 64 == 0, pad bytes 0), PLINK
// 2-bit codes, individual 4j+k in bits 2k..2k+1 of byte j (data.cpp:201-234 keeps the same bytes, unpadded).
// A 32-bit word of a row therefore holds 16 consecutive individuals, individual q at bits 2q..2q+1.
// Decode (dotp_lut.hpp:3,1030): code 00 -> a=2, 10 -> a=1, 11 -> a=0, 01 -> missing (a=0, b=0).
#include <cstdlib>
#include "gv_internal.h"
#include "gv_pval_dev.h"

namespace {

constexpr int WAVE = 64;

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
    return v;
}

// block of 256 threads: sum of v over the block, valid in thread 0.  sh: 4 doubles.
__device__ __forceinline__ double block_sum_256(double v, double* sh) {
    v = wave_sum(v);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}

// scalar k of nblocks block partials (K per block), added up by a block of 256 threads for itself in the order of k_finalize
// (strided partial sums, then a fixed tree): the consumer of a reduction takes the place of a k_finalize launch -- same bits,
// one launch less per reduction (a CG step has two or three; profiles/r3_cfg5_gaps.txt).  sh: 256 doubles.  Valid in every thread.
__device__ __forceinline__ double sum_partials_256(const double* __restrict__ partial, int nblocks, int K, int k, double* sh) {
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * K + k];
    __syncthreads();
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    return sh[0];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// ---- synthetic shard (SURVEY 8d recipe, integer-only so the host generator matches bit for bit) ---------
// marker g (global index): maf = (3277 + h % 29491) / 65536 in [0.05, 0.5); genotype ~ Binomial(2, maf);
// missing with probability miss_thr / 2^32.  Codes: 2 -> 00, 1 -> 10, 0 -> 11, missing -> 01.
// Linkage disequilibrium (ld_block > 0): the markers of a block of ld_block consecutive markers share, per individual, one
// latent uniform; each genotype takes it with probability ld_thr / 2^32 instead of its own draw and pushes it through the
// marker's own allele-frequency thresholds -- co-monotone genotypes inside a block, independent blocks (block-correlated
// columns: what makes the CG of the LMMSE step run tens of iterations on real genotypes).
__global__ void k_synth_bed(uint32_t* bed, int64_t M, int64_t S, int64_t N, int64_t P4, uint64_t seed,
                            uint32_t miss_thr, uint32_t ld_block, uint32_t ld_thr) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= M * P4) return;
    int64_t m = idx / P4, j = idx % P4;
    uint64_t g = (uint64_t)(S + m);
    uint64_t hm = splitmix64(seed ^ (g * 0xD1342543DE82EF95ull));
    uint32_t maf = 3277u + (uint32_t)(hm % 29491ull);
    uint32_t qv = 65536u - maf;
    uint32_t p0 = qv * qv;              // P(geno 0) * 2^32  (qv <= 62259 -> fits)
    uint32_t p1 = 2u * maf * qv;        // P(geno 1) * 2^32
    uint64_t base = splitmix64(hm + 0x632BE59BD9B4E019ull);
    const uint64_t lbase = ld_block ? splitmix64(seed ^ ((g / ld_block) * 0xA24BAED4963EE407ull) ^ 0x5851F42D4C957F2Dull) : 0;
    uint32_t w = 0;
    for (int q = 0; q < 16; q++) {
        int64_t n = 16 * j + q;
        if (n >= N) break;
        uint64_t r = splitmix64(base + (uint64_t)n);
        uint32_t u = (uint32_t)(r >> 32), um = (uint32_t)r;
        if (ld_block) {
            const uint64_t rs = splitmix64(r ^ 0x9FB21C651E98DF25ull);
            if ((uint32_t)(rs >> 32) < ld_thr) u = (uint32_t)(splitmix64(lbase + (uint64_t)n) >> 32);
        }
        uint32_t code;
        if (um < miss_thr) code = 1u;
        else if (u < p0) code = 3u;
        else if (u - p0 < p1) code = 2u;
        else code = 0u;
        w |= code << (2 * q);
    }
    bed[idx] = w;
}

// ---- compute_markers_statistics (data.cpp:451-484), count form ------------------------------------------
// One wave per marker.  n2/n1/n0 = number of present individuals with a = 2/1/0 (b = 1); the masked sums of
// the reference are then exact integers: suma = 2 n2 + n1, sumb = n0 + n1 + n2,
// sumsqr = n2 (2-mu)^2 + n1 (1-mu)^2 + n0 mu^2.
__global__ __launch_bounds__(256) void k_marker_stats(const uint32_t* __restrict__ bed,
                                                      const uint32_t* __restrict__ mask2, int64_t M, int64_t P4,
                                                      double nonas, double alpha_scale, double* __restrict__ mave,
                                                      double* __restrict__ msig, uint32_t* __restrict__ counts) {
    int lane = threadIdx.x & 63;
    int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const uint32_t* row = bed + m * P4;
    uint32_t n2 = 0, n1 = 0, n0 = 0;
    for (int64_t j = lane; j < P4; j += WAVE) {
        uint32_t w = row[j], pm = mask2[j] & 0x55555555u;
        uint32_t lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
        n2 += __popc(~lo & ~hi & pm);
        n1 += __popc(hi & ~lo & pm);
        n0 += __popc(hi & lo & pm);
    }
    n2 = wave_sum_u32(n2);
    n1 = wave_sum_u32(n1);
    n0 = wave_sum_u32(n0);
    if (lane == 0) {
        double suma = 2.0 * n2 + 1.0 * n1, sumb = (double)n0 + (double)n1 + (double)n2;
        double mu = (sumb != 0) ? suma / sumb : 0.0;
        double sumsqr = n2 * ((2.0 - mu) * (2.0 - mu)) + n1 * ((1.0 - mu) * (1.0 - mu)) + n0 * (mu * mu);
        double sg;
        if (sumsqr != 0) {
            if (alpha_scale == 1.0) sg = 1.0 / sqrt(sumsqr / (nonas - 1.0));
            else sg = 1.0 / pow(sqrt(sumsqr / (nonas - 1.0)), alpha_scale);
        } else
            sg = 1.0;
        mave[m] = mu;
        msig[m] = sg;
        if (counts) {
            counts[3 * m] = n2;
            counts[3 * m + 1] = n1;
            counts[3 * m + 2] = n0;
        }
    }
}

// ---- data::Ax (data.cpp:951-1007), fp64 VALU family ------------------------------------------------------
// per-marker table: c = msig*x ; t = {(2-mu)c, (1-mu)c, (0-mu)c} = the three non-zero values of
// (a - ave) * sig_phen_i * b of data.cpp:972.
__global__ void k_ax_table(const double* __restrict__ x, const double* __restrict__ mave,
                           const double* __restrict__ msig, int64_t M, double* __restrict__ t3) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    double c = msig[i] * x[i], mu = mave[i];
    t3[3 * i + 0] = (2.0 - mu) * c;
    t3[3 * i + 1] = (1.0 - mu) * c;
    t3[3 * i + 2] = (0.0 - mu) * c;
}

// A thread owns one 32-bit column (16 individuals, accumulators in registers) and walks a chunk of markers;
// consecutive threads read consecutive words of the row (coalesced 1 KiB per block per marker).
// grid = (column tiles, marker chunks); partial sums per chunk are combined by k_ax_reduce in chunk order.
__global__ __launch_bounds__(256) void k_ax_f64(const uint32_t* __restrict__ bed, int64_t M, int64_t P4,
                                                const double* __restrict__ t3, int64_t mpc,
                                                double* __restrict__ partial, int64_t npad) {
    int64_t col = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t m0 = (int64_t)blockIdx.y * mpc, m1 = m0 + mpc < M ? m0 + mpc : M;
    if (col >= P4) return;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; q++) acc[q] = 0.0;
    const uint32_t* colp = bed + col;
#pragma unroll 4
    for (int64_t m = m0; m < m1; ++m) {
        uint32_t w = colp[m * P4];
        double t2 = t3[3 * m], t1 = t3[3 * m + 1], t0 = t3[3 * m + 2];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            uint32_t code = (w >> (2 * q)) & 3u;
            double v = (code == 0u) ? t2 : (code == 2u) ? t1 : (code == 3u) ? t0 : 0.0;
            acc[q] += v;
        }
    }
    double2* dst = reinterpret_cast<double2*>(partial + (int64_t)blockIdx.y * npad + col * 16);
#pragma unroll
    for (int q = 0; q < 8; q++) dst[q] = make_double2(acc[2 * q], acc[2 * q + 1]);
}

// out[n] = mask[n] * scale * sum_chunks partial[chunk][n]   (na_lut factor of data.cpp:972; scale :998-1005
// is applied here only when no cross-rank all-reduce follows, otherwise scale = 1 and k_scale runs after it)
__global__ void k_ax_reduce(const double* __restrict__ partial, int chunks, int64_t npad,
                            const uint32_t* __restrict__ mask2, double scale, double* __restrict__ out) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= npad) return;
    double s = 0.0;
    for (int c = 0; c < chunks; c++) s += partial[(int64_t)c * npad + n];
    uint32_t present = (mask2[n >> 4] >> (2 * (n & 15))) & 1u;
    out[n] = present ? s * scale : 0.0;
}

__global__ void k_scale(double* v, int64_t n, double a) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] *= a;
}

// ---- data::ATx (data.cpp:810-835 over dot_product :758-779), fp64 VALU family ---------------------------------
// A wave handles MW markers over the whole row: the 16 phenotype values of a lane's word are loaded once and
// used for MW markers; per-marker partial sums stay in registers until one wave reduction at the end.
template <int MW>
__global__ __launch_bounds__(256) void k_atx_f64(const uint32_t* __restrict__ bed, int64_t M, int64_t P4,
                                                 const double* __restrict__ p, const double* __restrict__ mave,
                                                 const double* __restrict__ msig, double scale,
                                                 double* __restrict__ out) {
    int lane = threadIdx.x & 63;
    int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * MW;
    if (m0 >= M) return;
    double sa[MW], sb[MW];
#pragma unroll
    for (int k = 0; k < MW; k++) sa[k] = sb[k] = 0.0;
    for (int64_t j = lane; j < P4; j += WAVE) {
        uint32_t w[MW];
#pragma unroll
        for (int k = 0; k < MW; k++) w[k] = (m0 + k < M) ? bed[(m0 + k) * P4 + j] : 0x55555555u;
        const double2* pp = reinterpret_cast<const double2*>(p + j * 16);
#pragma unroll
        for (int h = 0; h < 8; h++) {
            double2 pv = pp[h];
#pragma unroll
            for (int k = 0; k < MW; k++) {
                uint32_t c0 = (w[k] >> (4 * h)) & 3u, c1 = (w[k] >> (4 * h + 2)) & 3u;
                sa[k] += (c0 == 0u) ? 2.0 * pv.x : (c0 == 2u) ? pv.x : 0.0;
                sb[k] += (c0 == 1u) ? 0.0 : pv.x;
                sa[k] += (c1 == 0u) ? 2.0 * pv.y : (c1 == 2u) ? pv.y : 0.0;
                sb[k] += (c1 == 1u) ? 0.0 : pv.y;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MW; k++) {
        double a = wave_sum(sa[k]), b = wave_sum(sb[k]);
        if (lane == 0 && m0 + k < M) out[m0 + k] = msig[m0 + k] * (a - mave[m0 + k] * b) * scale;
    }
}

// Raw per-marker sums of two N-vectors (fp64 family of gvm::marker_sums2): out4[4m..] = {sum a p1, sum b p1,
// sum a p2, sum b p2} -- the ingredients of data::pvals_calc (data.cpp:1150-1170).  A wave handles 4 markers.
__global__ __launch_bounds__(256) void k_marker_sums2_f64(const uint32_t* __restrict__ bed, int64_t M, int64_t P4,
                                                          const double* __restrict__ p1, const double* __restrict__ p2,
                                                          double* __restrict__ out4) {
    constexpr int MW = 4;
    int lane = threadIdx.x & 63;
    int64_t m0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * MW;
    if (m0 >= M) return;
    double s[MW][4];
#pragma unroll
    for (int k = 0; k < MW; k++) s[k][0] = s[k][1] = s[k][2] = s[k][3] = 0.0;
    for (int64_t j = lane; j < P4; j += WAVE) {
        uint32_t w[MW];
#pragma unroll
        for (int k = 0; k < MW; k++) w[k] = (m0 + k < M) ? bed[(m0 + k) * P4 + j] : 0x55555555u;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const double v1 = p1[j * 16 + q], v2 = p2[j * 16 + q];
#pragma unroll
            for (int k = 0; k < MW; k++) {
                uint32_t c = (w[k] >> (2 * q)) & 3u;
                double a = (c == 0u) ? 2.0 : (c == 2u) ? 1.0 : 0.0, b = (c == 1u) ? 0.0 : 1.0;
                s[k][0] += a * v1;
                s[k][1] += b * v1;
                s[k][2] += a * v2;
                s[k][3] += b * v2;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < MW; k++) {
        double r0 = wave_sum(s[k][0]), r1 = wave_sum(s[k][1]), r2 = wave_sum(s[k][2]), r3 = wave_sum(s[k][3]);
        if (lane == 0 && m0 + k < M) {
            double* o = out4 + 4 * (m0 + k);
            o[0] = r0; o[1] = r1; o[2] = r2; o[3] = r3;
        }
    }
}

// ---- data::compute_people_statistics (data.cpp:558-716): per-marker tables for three k_ax_f64 passes -----------------
// kind 0: value = (a - mu) sigma          -> {(2-mu)s, (1-mu)s, (0-mu)s}
// kind 1: b (count of non-missing)        -> {1, 1, 1}
// kind 2: value^2
__global__ void k_people_table(const double* __restrict__ mave, const double* __restrict__ msig, int64_t M, int kind,
                               double* __restrict__ t3) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    const double mu = mave[i], s = msig[i];
    double v2 = (2.0 - mu) * s, v1 = (1.0 - mu) * s, v0 = (0.0 - mu) * s;
    if (kind == 1) v2 = v1 = v0 = 1.0;
    if (kind == 2) { v2 *= v2; v1 *= v1; v0 *= v0; }
    t3[3 * i] = v2;
    t3[3 * i + 1] = v1;
    t3[3 * i + 2] = v0;
}
// mean = S1 / cnt ; msig = sqrt((cnt - 1) / (S2 - cnt mean^2)) for individuals with a phenotype, else 0 (data.cpp:608-624)
__global__ void k_people_finish(double* __restrict__ s1, double* __restrict__ s2, const double* __restrict__ cnt,
                                const uint32_t* __restrict__ mask2, int64_t N, int64_t npad) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= npad) return;
    const uint32_t present = (mask2[n >> 4] >> (2 * (n & 15))) & 1u;
    if (n < N && present) {
        const double c = cnt[n], mean = s1[n] / c;
        s1[n] = mean;
        s2[n] = sqrt((c - 1) / (s2[n] - c * mean * mean));
    } else {
        s1[n] = 0.0;
        s2[n] = 0.0;
    }
}
// preconditioner of CG_solverAAT (denoiserXXT.cpp:59-66)
__global__ void k_aat_diag(const double* __restrict__ mave_p, const double* __restrict__ msig_p,
                           const double* __restrict__ numb_p, double tau, double gam2, double Nd, int64_t npad,
                           double* __restrict__ diag) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= npad) return;
    diag[n] = tau * ((numb_p[n] - 1) / msig_p[n] / msig_p[n] + mave_p[n] * mave_p[n] * numb_p[n]) / Nd + gam2;
}
// denoiserXXT.cpp:100-118: r -= alpha d ; z = r / diag[n] ; partials of <r,z>, <r,r>.  alpha == 0: initialisation pass.
__global__ __launch_bounds__(256) void k_cg_b_diag(double* __restrict__ r, const double* __restrict__ d, double alpha,
                                                   const double* __restrict__ diag, double* __restrict__ z, int64_t n,
                                                   double* __restrict__ partial) {
    __shared__ double sh[4];
    int64_t stride = (int64_t)gridDim.x * 256;
    double s0 = 0, s1 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double ri = r[i];
        if (alpha != 0.0) ri -= d[i] * alpha;
        double zi = ri / diag[i];
        r[i] = ri;
        z[i] = zi;
        s0 += ri * zi;
        s1 += ri * ri;
    }
    s0 = block_sum_256(s0, sh);
    s1 = block_sum_256(s1, sh);
    if (threadIdx.x == 0) {
        partial[(int64_t)blockIdx.x * 2] = s0;
        partial[(int64_t)blockIdx.x * 2 + 1] = s1;
    }
}

// ---- probit (--model bin_class): vamp::g1_bin_class / g1d_bin_class over the N individuals (vamp_probit.cpp:335-352,
// :661-705); m_cov (may be NULL) is the per-individual covariate offset.  z1 = posterior mean of z given y in {0,1} and the cavity N(p1, 1/tau1); partials of
// sum g1d and sum (z1 - p1)^2.  Entries n >= N of z1 are zeroed.
__global__ __launch_bounds__(256) void k_probit_denoise(const double* __restrict__ p1, const double* __restrict__ y,
                                                        const double* __restrict__ m_cov, int64_t N, int64_t npad,
                                                        double tau1, double probit_var, double* __restrict__ z1,
                                                        double* __restrict__ partial) {
    __shared__ double sh[4];
    const double inv_s = 1.0 / sqrt(probit_var + 1.0 / tau1);
    int64_t stride = (int64_t)gridDim.x * 256;
    double s0 = 0, s1 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npad; i += stride) {
        double out = 0.0;
        if (i < N) {
            const double p = p1[i], sgn = 2 * y[i] - 1;
            const double c = (m_cov ? p + m_cov[i] : p) * inv_s;   // m_cov = <Z[i], cov_eff> (vamp_probit.cpp:347,:364)
            const double ratio = 2.0 / sqrt(2 * M_PI) / erfcx(-sgn * c / sqrt(2.0));   // pdf / cdf at sgn * c
            out = p + sgn * ratio / tau1 * inv_s;
            const double gd = 1 - ratio / (1 + tau1 * probit_var) * (sgn * c + ratio);
            s0 += gd;
            s1 += (out - p) * (out - p);
        }
        z1[i] = out;
    }
    s0 = block_sum_256(s0, sh);
    s1 = block_sum_256(s1, sh);
    if (threadIdx.x == 0) {
        partial[(int64_t)blockIdx.x * 2] = s0;
        partial[(int64_t)blockIdx.x * 2 + 1] = s1;
    }
}

// out = x * y (element-wise) and out = a where mask (chrom[i] == ch) else 0 helpers of the p-value passes
__global__ void k_mul(double* out, const double* x, const double* y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = x[i] * y[i];
}
__global__ void k_select_eq(double* out, const double* x, const int* key, int value, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (key[i] == value) ? x[i] : 0.0;
}

// ---- element-wise ------------------------------------------------------------------------------------------
__global__ void k_fill(double* v, int64_t n, double a) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = a;
}
__global__ void k_axpby(double* out, double a, const double* x, double b, const double* y, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (y ? a * x[i] + b * y[i] : a * x[i]);
}
// p = z + beta p (vamp.cpp:1209-1210), one rounding -- the same fma as the device-resident CG (k_prep_ax)
__global__ void k_vcopy(double* __restrict__ dst, const double* __restrict__ src, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void k_p_update(double* p, const double* z, double beta, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = fma(beta, p[i], z[i]);
}
__global__ void k_mask_copy(double* out, const double* in, const uint32_t* mask2, int64_t npad) {
    int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= npad) return;
    uint32_t present = (mask2[n >> 4] >> (2 * (n & 15))) & 1u;
    out[n] = present ? in[n] : 0.0;
}
// ---- loop-back transport (gv_debug_force_multi): a one-rank all-reduce that really moves the message.  k_loop_out takes the
// buffer to scratch and POISONS it (a quiet NaN with a recognisable payload), k_loop_wait holds the stream for a bounded number
// of microseconds, k_loop_in brings it back.  Whoever reads the buffer without being ordered behind the exchange reads NaNs.
__global__ void k_loop_out(double* __restrict__ buf, double* __restrict__ scratch, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    scratch[i] = buf[i];
    buf[i] = __longlong_as_double(0x7ff8000000c0ffeeLL);
}
__global__ void k_loop_wait(int us) {
    // wall_clock64 ticks at 100 MHz on gfx950; bounded by the iteration count as well (a stuck clock must not hang the stream)
    const unsigned long long t0 = wall_clock64(), ticks = (unsigned long long)us * 100ull;
    for (int it = 0; it < (1 << 22) && wall_clock64() - t0 < ticks; it++) __builtin_amdgcn_s_sleep(8);
}
__global__ void k_loop_in(double* __restrict__ buf, double* __restrict__ scratch, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    buf[i] = scratch[i];
    scratch[i] = __longlong_as_double(0x7ff8000000c0ffeeLL);
}
__global__ void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, int64_t n2) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) dst[i] = src[i];
}
// read-only stream probe with the access shape of the matvec kernels: every wave walks its own contiguous run of
// 4 KiB blocks (one coalesced, non-temporal KiB per instruction), two blocks in flight, nothing but an XOR per load.
// What it reaches is the ceiling a streaming kernel can be held against on this part.
typedef unsigned int rd_u32x4 __attribute__((ext_vector_type(4)));
// perm: which 16-byte piece of a 4 KiB block a lane takes in load i -- 0: lane-linear (the stripe kernels), 1: the ATx side of
// the tile layout (KiB i, piece (lane & 15) * 4 + (lane >> 4)), 2: its Ax side (256 contiguous bytes of each KiB)
__global__ __launch_bounds__(256, 3) void k_read_stream(const rd_u32x4* __restrict__ src, int64_t blocks_per_wave,
                                                        int64_t nwaves, unsigned int* __restrict__ sink, int perm) {
    const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
    const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= nwaves) return;
    const rd_u32x4* p = src + w * blocks_per_wave * 256;
    int off[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        off[i] = perm == 1 ? i * 64 + r * 4 + g : (perm == 2 ? (r >> 2) * 64 + (4 * i + g) * 4 + (r & 3) : i * 64 + lane);
    rd_u32x4 acc = {0, 0, 0, 0};
    rd_u32x4 a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = __builtin_nontemporal_load(p + off[i]);
    for (int64_t k = 1; k < blocks_per_wave; k++) {
#pragma unroll
        for (int i = 0; i < 4; i++) b[i] = __builtin_nontemporal_load(p + k * 256 + off[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) acc ^= a[i];
#pragma unroll
        for (int i = 0; i < 4; i++) a[i] = b[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) acc ^= a[i];
    const unsigned int v = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (v == 0x9E3779B9u) sink[0] = v;   // practically never: keeps the loads alive without a store stream
}

// Scalars for the host without a copy engine or a stream synchronisation: the K values go straight into mapped,
// coherent host memory, then -- after a system-scope fence -- the sequence number the host is spinning on.
__global__ void k_publish(const double* __restrict__ src, int K, double* mailbox, unsigned long long* flag,
                          unsigned long long seq) {
    if ((int)threadIdx.x < K) mailbox[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- reductions: block partials -> one ordered final pass (deterministic; SURVEY 7 parity trap d) ------------
__global__ __launch_bounds__(256) void k_finalize(const double* __restrict__ partial, int nblocks, int K,
                                                  double* __restrict__ out) {
    // one block per scalar k = blockIdx.x (grid = K): strided partial sums, then a fixed tree -- the order of additions
    // depends on nothing but nblocks, so every run (and every K) gives the same bits
    __shared__ double sh[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * K + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = sh[0];
}

// k_finalize that also publishes (read_scalars without a launch of its own): scalar k goes to out[k] AND to the host mailbox;
// the block that takes the last ticket releases the sequence number the host spins on.
__global__ __launch_bounds__(256) void k_finalize_pub(const double* __restrict__ partial, int nblocks, int K,
                                                      double* __restrict__ out, double* mailbox, unsigned long long* flag,
                                                      unsigned long long seq, unsigned int* __restrict__ counter) {
    __shared__ double sh[256];
    const int k = blockIdx.x;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 256) s += partial[(int64_t)b * K + k];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (threadIdx.x < off) sh[threadIdx.x] += sh[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[k] = sh[0];
        mailbox[k] = sh[0];
        __threadfence_system();
        if (atomicAdd(counter, 1u) == (unsigned int)K - 1) {
            *counter = 0;
            __threadfence_system();
            __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

struct DotArgs {
    const double* x[8];
    const double* y[8];
};
__global__ __launch_bounds__(256) void k_dots(DotArgs a, int K, int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    int64_t stride = (int64_t)gridDim.x * 256;
    for (int k = 0; k < K; k++) {
        const double *x = a.x[k], *y = a.y[k];
        double s = 0.0;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) s = fma(x[i], y[i], s);   // (explicit:
        // the fused <d,p> of k_fin_atx_dot must add the same roundings)
        s = block_sum_256(s, sh);
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * K + k] = s;
    }
}

// Several dots of one VAMP iteration in ONE launch and one read-back (gv_vec_dots_ex): pair k is <xa - xb, ya - yb> over n[k]
// entries (xb / yb may be NULL).  Pair k is summed exactly as a k_dots launch of its own over n[k] entries would sum it -- the
// first nb[k] = red_blocks(n[k]) blocks stride over it by nb[k] * 256, the others leave +0 partials, which add nothing in
// k_finalize -- and a difference is rounded once, as k_axpby(1, ., -1, .) rounds it: the batched scalars are bit-identical to
// those of the axpby + dots launches they replace.
struct DotExArgs { const double* xa[8]; const double* xb[8]; const double* ya[8]; const double* yb[8]; long long n[8]; int nb[8]; };
__global__ __launch_bounds__(256) void k_dots_ex(DotExArgs a, int K, double* __restrict__ partial) {
    __shared__ double sh[4];
    for (int k = 0; k < K; k++) {
        double s = 0.0;
        if ((int)blockIdx.x < a.nb[k]) {
            const double *xa = a.xa[k], *xb = a.xb[k], *ya = a.ya[k], *yb = a.yb[k];
            const bool square = xa == ya && xb == yb;
            const int64_t n = a.n[k], stride = (int64_t)a.nb[k] * 256;
            for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
                double x = xa[i];
                if (xb) x = x - xb[i];
                double y = x;
                if (!square) {
                    y = ya[i];
                    if (yb) y = y - yb[i];
                }
                s = fma(x, y, s);
            }
        }
        s = block_sum_256(s, sh);
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * K + k] = s;
    }
}

// vamp.cpp:1169-1176 : mu += alpha p ; partial of <v, mu>
__global__ __launch_bounds__(256) void k_cg_a(double* __restrict__ mu, const double* __restrict__ p, double alpha,
                                              const double* __restrict__ v, int64_t n,
                                              double* __restrict__ partial) {
    __shared__ double sh[4];
    int64_t stride = (int64_t)gridDim.x * 256;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double m = fma(alpha, p[i], mu[i]);        // explicit fma / mul in k_cg_a, k_cg_b and k_cgx_ab: the host-driven and
        mu[i] = m;                                 // the device-resident CG step must round identically
        s = fma(v[i], m, s);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// vamp.cpp:1195-1216 : r -= alpha d ; z = r / diag ; partials of <r,z>, <z,z>, <r,r>, <mu,mu>
__global__ __launch_bounds__(256) void k_cg_b(double* __restrict__ r, const double* __restrict__ d, double alpha,
                                              double diag, double* __restrict__ z, const double* __restrict__ mu,
                                              int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    int64_t stride = (int64_t)gridDim.x * 256;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double ri = r[i];
        if (alpha != 0.0) ri = fma(-d[i], alpha, ri);   // alpha == 0: initialisation pass (z = r/diag, norms), d unread
        double zi = ri / diag;
        r[i] = ri;
        z[i] = zi;
        double m = mu[i];
        s0 = fma(ri, zi, s0);
        s1 = fma(zi, zi, s1);
        s2 = fma(ri, ri, s2);
        s3 = fma(m, m, s3);
    }
    s0 = block_sum_256(s0, sh);
    s1 = block_sum_256(s1, sh);
    s2 = block_sum_256(s2, sh);
    s3 = block_sum_256(s3, sh);
    if (threadIdx.x == 0) {
        double* o = partial + (int64_t)blockIdx.x * 4;
        o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
    }
}

// ---- device-resident CG step (gv_solvers.hip: cg_run_device) ------------------------------------------------------------
// One launch for every still-active system (blockIdx.y): alpha = <r,z> / <d,p> from the system's state block and the reduced
// <d,p> the ATx epilogue left, then k_cg_a and k_cg_b in one sweep -- mu += alpha p, r -= alpha d, z = r / diag -- with the
// block partials of <v,mu>, <r,z>, <r,r>, added up by gvk::finalize in the fixed order of k_finalize (red[v][0..2]).  (The reference leaves r one step behind on an Onsager-rule exit, vamp.cpp:1176-1193; here r is always
// current, which is what the by-products of gv_cg_solve2x want and nobody else reads.)
struct CgxAB {
    double* st[2]; double* mu[2]; const double* p[2]; const double* v[2]; double* r[2]; const double* d[2]; double* z[2];
    const double* dp[2]; double* part[2]; double* red[2];
    const double* dpp[2]; int dp_nb;      // dpp[v] != NULL: <d,p> still in dp_nb block partials (one rank: no all-reduce, no k_finalize)
    double* az[2]; const double* aw[2]; long long npad;   // az[v] != NULL: A mu += alpha A p over npad entries (gv_cg_extras.a_mu_a)
};
__global__ __launch_bounds__(256) void k_cgx_ab(CgxAB a, double diag, int64_t n) {
    __shared__ double sh[4];
    __shared__ double shr[256];
    const int v = blockIdx.y;
    double* st = a.st[v];
    if (st[gvm::ST_ACTIVE] == 0.0) return;
    const double dp = a.dpp[v] ? sum_partials_256(a.dpp[v], a.dp_nb, 1, 0, shr) : a.dp[v][0];
    const double alpha = st[gvm::ST_RZ] / dp;                            // vamp.cpp:1167
    double* __restrict__ mu = a.mu[v];
    const double* __restrict__ p = a.p[v];
    const double* __restrict__ vv = a.v[v];
    double* __restrict__ r = a.r[v];
    const double* __restrict__ d = a.d[v];
    double* __restrict__ z = a.z[v];
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (a.az[v]) {      // A mu accumulated from the A p_k (what k_axpy_st did in a launch of its own: the same expression, the same bits)
        double* __restrict__ y = a.az[v];
        const double* __restrict__ x = a.aw[v];
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.npad; i += stride) y[i] += alpha * x[i];
    }
    double s_vmu = 0, s_rz = 0, s_rr = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        double m = fma(alpha, p[i], mu[i]);
        mu[i] = m;
        s_vmu = fma(vv[i], m, s_vmu);
        double ri = fma(-d[i], alpha, r[i]);
        double zi = ri / diag;
        r[i] = ri;
        z[i] = zi;
        s_rz = fma(ri, zi, s_rz);
        s_rr = fma(ri, ri, s_rr);
    }
    s_vmu = block_sum_256(s_vmu, sh);
    s_rz = block_sum_256(s_rz, sh);
    s_rr = block_sum_256(s_rr, sh);
    if (threadIdx.x == 0) {
        double* o = a.part[v] + (int64_t)blockIdx.x * 3;
        o[0] = s_vmu; o[1] = s_rz; o[2] = s_rr;
        if (blockIdx.x == 0) st[gvm::ST_ALPHA] = alpha;
    }
    // (block partials -> gvk::finalize behind this kernel; see k_fin_atx_dot for why not a last-block ticket)
}
// ---- vamp::CG_solverAAT (denoiserXXT.cpp:52-130), the N-space CG of --use-XXT-denoiser 1, with its scalars on the device ----
// d = tau d + gam2 p (the rest of lmmse_multAAT, denoiserXXT.cpp:30-33) and the block partials of <d, p> (:88)
__global__ __launch_bounds__(256) void k_aat_dq(double* __restrict__ d, const double* __restrict__ p, double tau, double gam2,
                                                int64_t n, double* __restrict__ partial) {
    __shared__ double sh[4];
    const int64_t stride = (int64_t)gridDim.x * 256;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double pi = p[i];
        const double di = fma(tau, d[i], gam2 * pi);
        d[i] = di;
        s = fma(di, pi, s);
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// alpha = <r,z> / <d,p> (:88); mu += alpha p (:90-93); r -= alpha d; z = r / diag[n] (:95-105); partials of <r,z>, <r,r>
// <d,p> arrives as the dp_nb block partials of k_aat_dq (every block adds them up for itself: sum_partials_256); at_acc (may be
// NULL): A^T mu += alpha A^T p over m entries, what k_axpy_st did in a launch of its own
__global__ __launch_bounds__(256) void k_aat_ab(double* __restrict__ st, double* __restrict__ mu, const double* __restrict__ p,
                                                double* __restrict__ r, const double* __restrict__ d, double* __restrict__ z,
                                                const double* __restrict__ diag, const double* __restrict__ dp_part, int dp_nb,
                                                int64_t n, double* __restrict__ partial, double* __restrict__ at_acc,
                                                const double* __restrict__ at_p, int64_t m) {
    __shared__ double sh[4];
    __shared__ double shr[256];
    if (st[gvm::ST_ACTIVE] == 0.0) return;
    const double alpha = st[gvm::ST_RZ] / sum_partials_256(dp_part, dp_nb, 1, 0, shr);
    const int64_t stride = (int64_t)gridDim.x * 256;
    if (at_acc)
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < m; i += stride) at_acc[i] += alpha * at_p[i];
    double s0 = 0, s1 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        mu[i] = fma(alpha, p[i], mu[i]);
        const double ri = fma(-d[i], alpha, r[i]);
        const double zi = ri / diag[i];
        r[i] = ri;
        z[i] = zi;
        s0 = fma(ri, zi, s0);
        s1 = fma(ri, ri, s1);
    }
    s0 = block_sum_256(s0, sh);
    s1 = block_sum_256(s1, sh);
    if (threadIdx.x == 0) {
        partial[(int64_t)blockIdx.x * 2] = s0;
        partial[(int64_t)blockIdx.x * 2 + 1] = s1;
        if (blockIdx.x == 0) st[gvm::ST_ALPHA] = alpha;
    }
}
// beta, <r,z>, ||r|| / ||v|| and its trace, the 1e-4 stopping rule (:98-120); status -> host mailbox (slot seq & 1, system 0).
// ST_NORMV holds ||v||^2 for this solver (the reference takes the square root of the ratio, :113-114).
// other_st / go (may be NULL): the joint solver's pipelined loop (gv_cg_solve_aat2w) -- *go = this system or the M-space system
// behind other_st is still running: passes enqueued after both have finished are dropped on the device.  ride (may be NULL):
// reported in status word 7 (1 rider pending, 2 rode).
// part / nb: the block partials k_aat_ab left (<r,z>, <r,r>), added up here (one block of 256 threads) instead of by k_finalize
__global__ __launch_bounds__(256) void k_aat_decide(double* __restrict__ st, const double* __restrict__ part, int nb,
                                                    double* __restrict__ relres, int max_iter, double* mailbox,
                                                    unsigned long long* flag, unsigned long long seq, const double* other_st,
                                                    int* go, const int* ride, int* ride_mark) {
    __shared__ double shr[256];
    double red[2];
    red[0] = sum_partials_256(part, nb, 2, 0, shr);
    red[1] = sum_partials_256(part, nb, 2, 1, shr);
    if (threadIdx.x == 0) {
        // the rider rode in the Ax pass just made if it was pending and exactly one system had finished before it (flags unchanged yet)
        if (ride_mark && *ride_mark == 1 && other_st && ((st[gvm::ST_ACTIVE] != 0.0) != (other_st[gvm::ST_ACTIVE] != 0.0))) *ride_mark = 2;
        if (st[gvm::ST_ACTIVE] != 0.0) {
            const int iters = (int)st[gvm::ST_ITERS] + 1;
            st[gvm::ST_ITERS] = (double)iters;
            st[gvm::ST_BETA] = red[0] / st[gvm::ST_RZ];
            st[gvm::ST_RZ] = red[0];
            const double rel = sqrt(red[1] / st[gvm::ST_NORMV]);
            st[gvm::ST_RELERR] = rel;
            if (relres) relres[iters - 1] = rel;
            st[gvm::ST_NRELRES] = (double)iters;
            st[gvm::ST_STEPPED] = 1.0;
            if (rel < 1e-4) {
                st[gvm::ST_CONV] = 1.0;
                st[gvm::ST_ACTIVE] = 0.0;
            } else if (iters >= max_iter)
                st[gvm::ST_ACTIVE] = 0.0;
        } else
            st[gvm::ST_STEPPED] = 0.0;
        double* mb = mailbox + (seq & 1ull) * (2 * 8);
        mb[0] = st[gvm::ST_ACTIVE]; mb[1] = st[gvm::ST_ITERS]; mb[2] = st[gvm::ST_CONV]; mb[3] = st[gvm::ST_RELERR];
        mb[4] = 0.0; mb[5] = st[gvm::ST_NRELRES]; mb[6] = st[gvm::ST_STEPPED]; mb[7] = ride ? (double)*ride : 0.0;
        if (go) *go = (st[gvm::ST_ACTIVE] != 0.0 || (other_st && other_st[gvm::ST_ACTIVE] != 0.0)) ? 1 : 0;
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// p = z + beta p (:109-110) while the system is running, beta from its state block
__global__ void k_p_update_st(double* __restrict__ p, const double* __restrict__ z, const double* __restrict__ st, int64_t n) {
    if (st[gvm::ST_ACTIVE] == 0.0 || st[gvm::ST_STEPPED] == 0.0) return;
    const double beta = st[gvm::ST_BETA];
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = fma(beta, p[i], z[i]);
}

// initial state block of a CG system / the go and rider words, carried by the launch itself: no pinned staging block that a
// stream synchronisation would have to protect before the host may write the next one
struct StateInit { double q[gvm::ST_SIZE]; };
__global__ void k_state_init(double* __restrict__ dst, StateInit a) {
    if (threadIdx.x < gvm::ST_SIZE) dst[threadIdx.x] = a.q[threadIdx.x];
}
// ... or built on the device from the block partials of the solve's two opening reductions (one rank: nothing to all-reduce, so
// nothing for the host to read): ST_RZ = scalar k_rz of part_rz, ST_NORMV = ||v|| (sqrt_norm) or ||v||^2 from part_vv; the other
// entries as given.  Sums in k_finalize's order, what the host used to read back.
__global__ __launch_bounds__(256) void k_state_from_partials(double* __restrict__ dst, StateInit a, const double* __restrict__ part_rz,
                                                             int nb_rz, int K_rz, int k_rz, const double* __restrict__ part_vv,
                                                             int nb_vv, int sqrt_norm) {
    __shared__ double shr[256];
    const double rz = sum_partials_256(part_rz, nb_rz, K_rz, k_rz, shr);
    const double vv = sum_partials_256(part_vv, nb_vv, 1, 0, shr);
    if (threadIdx.x < gvm::ST_SIZE) {
        double q = a.q[threadIdx.x];
        if (threadIdx.x == gvm::ST_RZ) q = rz;
        if (threadIdx.x == gvm::ST_NORMV) q = sqrt_norm ? sqrt(vv) : vv;
        dst[threadIdx.x] = q;
    }
}
// ... or from scalars that have been summed over the ranks already (sharded jobs: finalize + in-stream all-reduce, no host)
__global__ void k_state_from_scalars(double* __restrict__ dst, StateInit a, const double* __restrict__ rz, const double* __restrict__ vv,
                                     int sqrt_norm) {
    if (threadIdx.x < gvm::ST_SIZE) {
        double q = a.q[threadIdx.x];
        if (threadIdx.x == gvm::ST_RZ) q = rz[0];
        if (threadIdx.x == gvm::ST_NORMV) q = sqrt_norm ? sqrt(vv[0]) : vv[0];
        dst[threadIdx.x] = q;
    }
}
__global__ void k_set_ints(int* __restrict__ dst, int a, int b) {
    if (threadIdx.x == 0) { dst[0] = a; dst[1] = b; }
}
// go word from the state blocks the device-side opening left (cg_open_device): a system its first step already ended must not
// be stepped again because the host, which has read no status yet, still believes it active
__global__ void k_go_from_states(int* __restrict__ dst, const double* __restrict__ st0, const double* __restrict__ st1, int b) {
    if (threadIdx.x == 0) {
        dst[0] = (st0[gvm::ST_ACTIVE] != 0.0 || st1[gvm::ST_ACTIVE] != 0.0) ? 1 : 0;
        dst[1] = b;
    }
}
// the rider's product (see gvm::CgHook::ride) out of the slot that carried it
__global__ void k_ride_copy(double* __restrict__ out, const double* __restrict__ w0, const double* __restrict__ w1,
                            const double* __restrict__ st0, const double* __restrict__ st1, const int* __restrict__ ride,
                            int64_t n) {
    if (*ride != 1) return;
    const bool a0 = st0[gvm::ST_ACTIVE] != 0.0, a1 = st1[gvm::ST_ACTIVE] != 0.0;
    if (a0 == a1) return;
    const double* __restrict__ src = a0 ? w1 : w0;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = src[i];
}
// the rider rode in the pass just made if it was pending and exactly one of the two systems had finished before that pass (the
// flags still say so: the step kernels that may change them come behind this launch)
__global__ void k_ride_mark(const double* __restrict__ st0, const double* __restrict__ st1, int* __restrict__ ride) {
    if (threadIdx.x == 0 && *ride == 1 && ((st0[gvm::ST_ACTIVE] != 0.0) != (st1[gvm::ST_ACTIVE] != 0.0))) *ride = 2;
}
// The scalar part of a CG step (vamp.cpp:1174-1223) for every system, on the device: iteration count, Onsager rule, beta,
// <r,z>, relative residual and its trace, stopping rules; then *go = some system is still running, and the status of every
// system goes to the host mailbox (slot = seq & 1: the host reads status s while the device may already write s + 1).
struct CgxDecide { double* st[2]; const double* red[2]; double* relres[2]; int nsys; int* ride;
                   const double* other_st; const int* ride_report;      // other_st: a system stepped elsewhere that also keeps *go up
                   const double* part[2]; int part_nb; };               // part[v] != NULL: <v,mu>, <r,z>, <r,r> still in block partials
constexpr int CGX_STATUS = 8;   // doubles per system in the mailbox: active, iters, converged, rel_err, onsager, n_relres, stepped
__global__ __launch_bounds__(256) void k_cgx_decide(CgxDecide a, double gam2, int max_iter, int* __restrict__ go, double* mailbox,
                                                    unsigned long long* flag, unsigned long long seq) {
    __shared__ int any;
    __shared__ double shr[3][256];
    __shared__ double qs[2][3];
    for (int v = 0; v < a.nsys; v++) {
        if (a.part[v]) {       // the three sums of system v side by side, each in the order of k_finalize (sum_partials_256)
            double s3[3] = {0.0, 0.0, 0.0};
            for (int b = threadIdx.x; b < a.part_nb; b += 256)
                for (int k = 0; k < 3; k++) s3[k] += a.part[v][(int64_t)b * 3 + k];
            __syncthreads();
            for (int k = 0; k < 3; k++) shr[k][threadIdx.x] = s3[k];
            __syncthreads();
            for (int off = 128; off > 0; off >>= 1) {
                if (threadIdx.x < off)
                    for (int k = 0; k < 3; k++) shr[k][threadIdx.x] += shr[k][threadIdx.x + off];
                __syncthreads();
            }
            if (threadIdx.x == 0)
                for (int k = 0; k < 3; k++) qs[v][k] = shr[k][0];
        } else if (threadIdx.x == 0)
            for (int k = 0; k < 3; k++) qs[v][k] = a.red[v][k];
    }
    if (threadIdx.x == 0) {
        any = 0;
        // the rider rode in this step if it was pending and exactly one system had finished before the step (the flags
        // still say so: they are only changed below)
        if (a.ride && *a.ride == 1 && a.nsys == 2 &&
            ((a.st[0][gvm::ST_ACTIVE] != 0.0) != (a.st[1][gvm::ST_ACTIVE] != 0.0)))
            *a.ride = 2;
    }
    __syncthreads();
    const int v = threadIdx.x;
    if (v < a.nsys) {
        double* st = a.st[v];
        if (st[gvm::ST_ACTIVE] != 0.0) {
            const double* q = qs[v];                    // <v,mu>, <r,z>, <r,r>
            const int iters = (int)st[gvm::ST_ITERS] + 1;
            st[gvm::ST_ITERS] = (double)iters;
            bool stepping = true;
            if (st[gvm::ST_DENOISER] == 0.0) {          // :1174-1193
                const double ons = gam2 * q[0], prev = st[gvm::ST_PREV_ONS];
                const double oerr = (ons != 0) ? fabs((ons - prev) / ons) : 1.0;
                st[gvm::ST_ONS] = ons;
                if (oerr < 1e-8) {
                    st[gvm::ST_CONV] = 1.0;
                    st[gvm::ST_ACTIVE] = 0.0;
                    stepping = false;
                }
                st[gvm::ST_PREV_ONS] = ons;
            }
            if (stepping) {
                st[gvm::ST_BETA] = q[1] / st[gvm::ST_RZ];   // (1/<r,z>_old) * <r,z>_new (:1198,:1207)
                st[gvm::ST_RZ] = q[1];
                const double rel = sqrt(q[2]) / st[gvm::ST_NORMV];   // :1215
                st[gvm::ST_RELERR] = rel;
                if (a.relres[v]) a.relres[v][iters - 1] = rel;
                st[gvm::ST_NRELRES] = (double)iters;
                if (rel < 1e-5) {                       // :1217,:1222
                    st[gvm::ST_CONV] = 1.0;
                    st[gvm::ST_ACTIVE] = 0.0;
                } else if (iters >= max_iter)
                    st[gvm::ST_ACTIVE] = 0.0;
            }
            st[gvm::ST_STEPPED] = stepping ? 1.0 : 0.0;
        } else
            st[gvm::ST_STEPPED] = 0.0;
        if (st[gvm::ST_ACTIVE] != 0.0) atomicOr(&any, 1);
        double* mb = mailbox + (seq & 1ull) * (2 * CGX_STATUS) + v * CGX_STATUS;
        mb[0] = st[gvm::ST_ACTIVE]; mb[1] = st[gvm::ST_ITERS]; mb[2] = st[gvm::ST_CONV]; mb[3] = st[gvm::ST_RELERR];
        mb[4] = st[gvm::ST_ONS]; mb[5] = st[gvm::ST_NRELRES]; mb[6] = st[gvm::ST_STEPPED];
        mb[7] = a.ride ? (double)*a.ride : (a.ride_report ? (double)*a.ride_report : 0.0);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *go = (any || (a.other_st && a.other_st[gvm::ST_ACTIVE] != 0.0)) ? 1 : 0;
        __threadfence_system();
        __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// vamp.cpp:292-310 fused: x1 = g1(r1), d = g1d(r1), partials of sum d and sum (x1 - r1)^2.
// vamp::g1 / g1d (vamp.cpp:805-869) in the reference's operation order per element -- with gam1 = 1e-8 (the first iteration of a run)
// 1 + sigma (pkdd / pk - q^2) cancels ten digits, and only the same roundings give the reference's alpha1 (a re-associated form
// with the divisions folded into per-component constants was 2.5 x faster and moved x2 of such an iteration by 1.6e-3:
// tests/test_gpu_ld.py, tiny-variance prior).  What depends on the component only AND enters the element's arithmetic as the very
// same double is computed once per block: vs = var + sigma, eta_max - var, prob / sqrt(vs), prob / (vs sqrt(vs)) -- the two square
// roots and three of the seven divisions per (element, component).  Bit-identical to the formulas as written.
__global__ __launch_bounds__(256) void k_denoise(const double* __restrict__ r1, int64_t n, double gam1, gv_prior pr,
                                                 double* __restrict__ x1, double* __restrict__ dd,
                                                 double* __restrict__ partial) {
    __shared__ double sh[4];
    __shared__ double c_vs[GV_LMAX], c_dv[GV_LMAX], c_a[GV_LMAX], c_d[GV_LMAX];
    const double sigma = 1 / gam1;
    const bool identity = sigma < 1e-10 && sigma > -1e-10;      // vamp.cpp:813,844
    const int L = pr.L;
    if (threadIdx.x < L) c_vs[threadIdx.x] = pr.vars[threadIdx.x];
    __syncthreads();
    double eta_max = c_vs[0];
    for (int i = 1; i < L; i++) eta_max = fmax(eta_max, c_vs[i]);
    __syncthreads();
    if (threadIdx.x < L) {
        const double var = pr.vars[threadIdx.x], vs = var + sigma;
        c_vs[threadIdx.x] = vs;
        c_dv[threadIdx.x] = eta_max - var;
        c_a[threadIdx.x] = pr.probs[threadIdx.x] / sqrt(vs);
        c_d[threadIdx.x] = pr.probs[threadIdx.x] / (vs * sqrt(vs));
    }
    __syncthreads();
    const double es = eta_max + sigma;
    int64_t stride = (int64_t)gridDim.x * 256;
    double s0 = 0, s1 = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const double y = r1[i];
        double g = y, gd = 1;
        if (!identity) {
            double pk = 0, pkd = 0, pkdd = 0;
            const double h = -0.5 * (y * y);
            for (int j = 0; j < L; j++) {
                const double vs = c_vs[j];
                const double e = exp(h * c_dv[j] / vs / es);
                double z = c_a[j] * e;
                pk = pk + z;
                z = z / vs * y;
                pkd = pkd - z;
                const double z2 = z / vs * y;
                pkdd = pkdd - c_d[j] * e + z2;
            }
            g = y + sigma * pkd / pk;
            const double q = pkd / pk;
            gd = 1 + sigma * (pkdd / pk - q * q);
        }
        x1[i] = g;
        if (dd) dd[i] = gd;
        s0 += gd;
        s1 += (g - y) * (g - y);
    }
    s0 = block_sum_256(s0, sh);
    s1 = block_sum_256(s1, sh);
    if (threadIdx.x == 0) {
        partial[(int64_t)blockIdx.x * 2] = s0;
        partial[(int64_t)blockIdx.x * 2 + 1] = s1;
    }
}

// vamp::updatePrior E-step (vamp.cpp:953-1013).  pr.probs[j] holds omegas[j] (j >= 1), pr.vars[j] the variances.
// Everything that depends on the component only (the exponent coefficient, the Gaussian prefactor, the posterior mean gain and
// variance) is computed once per block, so an element costs one exp and a few FMAs per component instead of the nine divisions /
// square roots of the formulas as written (same values up to the rounding of the re-association).
// Shape: the arithmetic is small (200k markers x 15 components: ~3 us of the chip's fp64 rate) and the kernel took 19 us, because
// its three register arrays of LC doubles with LC unrolled exps in flight filled 256 VGPRs -- one-wave blocks, one or two waves per
// SIMD, three elements per thread one after the other and 47 wave reductions at the end, nothing to hide any of it behind.  Now the
// responsibilities' numerators wait in an LDS column per thread between the two loops over the components (the exp loop is
// rolled), only the 2 LC accumulators stay in registers, and 256-thread blocks run four (LC <= 16), three (<= 24) or two waves per
// SIMD: a thread takes one element up to 262k markers.
template <int LC>
__global__ __launch_bounds__(256, LC <= 16 ? 4 : (LC <= 24 ? 3 : 2)) void k_prior_estep_reg(const double* __restrict__ r1, int64_t n,
                                                                                            double gam1, double lambda, gv_prior pr,
                                                                                            double* __restrict__ partial) {
    __shared__ double c_exp[GV_LMAX - 1], c_pre[GV_LMAX - 1], c_gain[GV_LMAX - 1], c_var[GV_LMAX - 1];
    __shared__ double numL[LC][260];          // [component][thread] (+ 4: the row sums at the end read eight rows at once); re-used for those sums
    const int t = threadIdx.x, Lm1 = pr.L - 1, K = 1 + 2 * Lm1;
    const double noise_var = 1 / gam1;
    // (the largest variance: a loop over pr.vars is pr.L dependent scalar loads from the kernel arguments, ~2 us at the head of every
    // block; one load per thread and a pass over LDS instead)
    if (t < pr.L) numL[0][t] = pr.vars[t];
    __syncthreads();
    double max_sigma = numL[0][0];
    for (int j = 1; j < pr.L; j++) max_sigma = fmax(max_sigma, numL[0][j]);
    __syncthreads();
    if (t < Lm1) {
        const double v = pr.vars[t + 1];
        c_exp[t] = 0.5 * (max_sigma - v) / (v + noise_var) / (max_sigma + noise_var);
        c_pre[t] = lambda * pr.probs[t + 1] / sqrt(v + noise_var) / sqrt(2 * M_PI);
        c_gain[t] = gam1 / (1 / v + gam1);
        c_var[t] = 1.0 / (1.0 / v + gam1);
    }
    const double c0 = (1 - lambda) / sqrt(2 * M_PI * noise_var);
    const double e0 = 0.5 * max_sigma / noise_var / (noise_var + max_sigma);
    double accR[LC], accG[LC];
#pragma unroll
    for (int j = 0; j < LC; j++) accR[j] = accG[j] = 0.0;
    __syncthreads();
    double acc_pin = 0.0;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + t; i < n; i += stride) {
        const double r = r1[i], r2 = r * r;
        double sum_of_elems = 0.0;
        __asm__ volatile("" ::: "memory");                 // the per-component constants are read where they are used, not hoisted into 4 LC registers
#pragma unroll 4
        for (int j = 0; j < Lm1; j++) {
            const double v = c_pre[j] * exp(-r2 * c_exp[j]);
            numL[j][t] = v;
            sum_of_elems += v;
        }
        const double inv = 1 / sum_of_elems;
        const double pin = 1 / (1 + c0 * exp(-r2 * e0) * inv);
        acc_pin += pin;
        const double w = inv * pin;
#pragma unroll
        for (int j = 0; j < LC; j++)
            if (j < Lm1) {
                const double bp = numL[j][t] * w;          // beta * pin
                const double gm = c_gain[j] * r;
                accR[j] += bp;
                accG[j] = fma(bp, fma(gm, gm, c_var[j]), accG[j]);
            }
    }
    // the sums over the block, through LDS: 2 LC + 1 wave reductions by shuffles are 6 x 2 ds_bpermute each -- 10 us of LDS traffic per
    // CU with sixteen waves at it.  Instead every thread leaves its accumulator j in row j of the numerator array (free now), and
    // eight threads per row add 32 entries each in a fixed order, then one another's sums
    auto rows_sum = [&](const double (&acc)[LC], int slot0) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < LC; j++)
            if (j < Lm1) numL[j][t] = acc[j];
        __syncthreads();
        const int row = t >> 3, part = t & 7;
        double v = 0.0;
        if (row < Lm1) {
#pragma unroll 8
            for (int i = 0; i < 32; i++) v += numL[row][i * 8 + part];
        }
        v += __shfl_down(v, 4, 8);
        v += __shfl_down(v, 2, 8);
        v += __shfl_down(v, 1, 8);
        if (row < Lm1 && part == 0) partial[(int64_t)blockIdx.x * K + slot0 + 2 * row] = v;
    };
    rows_sum(accR, 1);
    rows_sum(accG, 2);
    __shared__ double sh[4];
    const double s = block_sum_256(acc_pin, sh);
    if (t == 0) partial[(int64_t)blockIdx.x * K] = s;
}

// ---- p-values: the per-marker regression test (device functions in gv_pval_dev.h) ----------------------------------------
// One marker per thread.  sums4[4k..] = {sum a p, sum b p, sum a p^2, sum b p^2} of the phenotype residual p over the
// marker's column (one two-vector pass); the column's own sums are exact, from the genotype counts.  xself != NULL
// (leave-one-out): the marker's own effect c = xself[k] * self_scale is added back analytically (data.cpp:1145-1148).
// chrom != NULL (LOCO): only the markers of chromosome ch are tested and written.
__global__ __launch_bounds__(128) void k_pvals_test(const uint32_t* __restrict__ cnt, const double* __restrict__ mave,
                                                    const double* __restrict__ msig, const double* __restrict__ sums4,
                                                    const double* __restrict__ xself, double self_scale,
                                                    const int* __restrict__ chrom, int ch, int64_t M,
                                                    double* __restrict__ pvals) {
    const int64_t k = (int64_t)blockIdx.x * 128 + threadIdx.x;
    if (k >= M) return;
    if (chrom && chrom[k] != ch) return;
    const double cself = xself ? xself[k] * self_scale : 0.0;
    const double s4[4] = {sums4[4 * k], sums4[4 * k + 1], sums4[4 * k + 2], sums4[4 * k + 3]};
    pvals[k] = gvp::marker_pval(cnt[3 * k], cnt[3 * k + 1], cnt[3 * k + 2], mave[k], msig[k], s4, cself);
}

inline int nblk(int64_t n, int bs) { return (int)((n + bs - 1) / bs); }
inline int red_blocks(int64_t n, int bs) {
    int64_t b = (n + bs - 1) / bs;
    return (int)(b < 1 ? 1 : (b > RED_BLOCKS ? RED_BLOCKS : b));
}

}  // namespace

namespace {
// armed by the caller that is about to read the scalars of the NEXT reduction back (gvk::arm_publish): its finalisation
// publishes to the host mailbox itself
thread_local struct { double* mailbox; unsigned long long* flag; unsigned long long seq; unsigned int* counter; bool on; } t_arm = {};
void launch_finalize(hipStream_t s, const double* partial, int nb, int K, double* out) {
    if (t_arm.on) {
        t_arm.on = false;
        hipLaunchKernelGGL(k_finalize_pub, dim3(K), dim3(256), 0, s, partial, nb, K, out, t_arm.mailbox, t_arm.flag, t_arm.seq,
                           t_arm.counter);
    } else
        hipLaunchKernelGGL(k_finalize, dim3(K), dim3(256), 0, s, partial, nb, K, out);
}
}  // namespace

namespace gvk {

void arm_publish(double* mailbox, unsigned long long* flag, unsigned long long seq, unsigned int* counter) {
    t_arm.mailbox = mailbox; t_arm.flag = flag; t_arm.seq = seq; t_arm.counter = counter; t_arm.on = true;
}
void disarm_publish() { t_arm.on = false; }

void synth_bed(hipStream_t s, uint8_t* bed, int64_t M, int64_t S, int64_t N, int64_t pitch, uint64_t seed,
               uint32_t miss_thr, uint32_t ld_block, uint32_t ld_thr) {
    int64_t P4 = pitch / 4, tot = M * P4;
    if (tot == 0) return;
    hipLaunchKernelGGL(k_synth_bed, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, (uint32_t*)bed, M, S, N, P4,
                       seed, miss_thr, ld_block, ld_thr);
}

void marker_stats(hipStream_t s, const uint8_t* bed, const uint32_t* mask2, int64_t M, int64_t pitch, double nonas,
                  double alpha_scale, double* mave, double* msig, uint32_t* counts) {
    if (M == 0) return;
    hipLaunchKernelGGL(k_marker_stats, dim3(nblk(M, 4)), dim3(256), 0, s, (const uint32_t*)bed, mask2, M, pitch / 4,
                       nonas, alpha_scale, mave, msig, counts);
}

void marker_sums2_f64(hipStream_t s, const uint8_t* bed, int64_t M, int64_t pitch, const double* p1, const double* p2,
                      double* out4) {
    if (M == 0) return;
    hipLaunchKernelGGL(k_marker_sums2_f64, dim3(nblk(M, 16)), dim3(256), 0, s, (const uint32_t*)bed, M, pitch / 4, p1, p2,
                       out4);
}

void people_table(hipStream_t s, const double* mave, const double* msig, int64_t M, int kind, double* t3) {
    if (M == 0) return;
    hipLaunchKernelGGL(k_people_table, dim3(nblk(M, 256)), dim3(256), 0, s, mave, msig, M, kind, t3);
}
void people_finish(hipStream_t s, double* s1, double* s2, const double* cnt, const uint32_t* mask2, int64_t N,
                   int64_t npad) {
    hipLaunchKernelGGL(k_people_finish, dim3(nblk(npad, 256)), dim3(256), 0, s, s1, s2, cnt, mask2, N, npad);
}
void aat_diag(hipStream_t s, const double* mave_p, const double* msig_p, const double* numb_p, double tau, double gam2,
              double Nd, int64_t npad, double* diag) {
    hipLaunchKernelGGL(k_aat_diag, dim3(nblk(npad, 256)), dim3(256), 0, s, mave_p, msig_p, numb_p, tau, gam2, Nd, npad, diag);
}
void cg_step_b_diag(hipStream_t s, double* r, const double* d, double alpha, const double* diag, double* z, int64_t n,
                    double* partial, double* out) {
    int nb = red_blocks(n, 256);
    hipLaunchKernelGGL(k_cg_b_diag, dim3(nb), dim3(256), 0, s, r, d, alpha, diag, z, n, partial);
    if (out) launch_finalize(s, partial, nb, 2, out);      // (out == NULL: the consumer adds the block partials up itself)
}

void probit_denoise(hipStream_t s, const double* p1, const double* y, const double* m_cov, int64_t N, int64_t npad,
                    double tau1, double probit_var, double* z1, double* partial, double* out) {
    int nb = red_blocks(npad, 256);
    hipLaunchKernelGGL(k_probit_denoise, dim3(nb), dim3(256), 0, s, p1, y, m_cov, N, npad, tau1, probit_var, z1, partial);
    launch_finalize(s, partial, nb, 2, out);
}

void mul(hipStream_t s, double* out, const double* x, const double* y, int64_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_mul, dim3(nblk(n, 256)), dim3(256), 0, s, out, x, y, n);
}

void select_eq(hipStream_t s, double* out, const double* x, const int* key, int value, int64_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_select_eq, dim3(nblk(n, 256)), dim3(256), 0, s, out, x, key, value, n);
}

void ax_table(hipStream_t s, const double* x, const double* mave, const double* msig, int64_t M, double* t3) {
    if (M == 0) return;
    hipLaunchKernelGGL(k_ax_table, dim3(nblk(M, 256)), dim3(256), 0, s, x, mave, msig, M, t3);
}

void ax_f64(hipStream_t s, const uint8_t* bed, int64_t M, int64_t pitch, const double* t3, int chunks,
            double* partial, int64_t npad) {
    int64_t P4 = pitch / 4;
    int64_t mpc = (M + chunks - 1) / chunks;
    if (mpc < 1) mpc = 1;
    hipLaunchKernelGGL(k_ax_f64, dim3(nblk(P4, 256), chunks), dim3(256), 0, s, (const uint32_t*)bed, M, P4, t3, mpc,
                       partial, npad);
}

void ax_reduce(hipStream_t s, const double* partial, int chunks, int64_t npad, const uint32_t* mask2, double scale,
               double* out) {
    hipLaunchKernelGGL(k_ax_reduce, dim3(nblk(npad, 256)), dim3(256), 0, s, partial, chunks, npad, mask2, scale, out);
}

void scale_vec(hipStream_t s, double* v, int64_t n, double a) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_scale, dim3(nblk(n, 256)), dim3(256), 0, s, v, n, a);
}

void atx_f64(hipStream_t s, const uint8_t* bed, int64_t M, int64_t pitch, const double* p, const double* mave,
             const double* msig, double scale, double* out) {
    if (M == 0) return;
    constexpr int MW = 8;
    hipLaunchKernelGGL(k_atx_f64<MW>, dim3(nblk(M, 4 * MW)), dim3(256), 0, s, (const uint32_t*)bed, M, pitch / 4, p,
                       mave, msig, scale, out);
}

void publish(hipStream_t s, const double* src, int K, double* mailbox, unsigned long long* flag, unsigned long long seq) {
    hipLaunchKernelGGL(k_publish, dim3(1), dim3(128), 0, s, src, K, mailbox, flag, seq);
}

void fill(hipStream_t s, double* v, int64_t n, double a) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_fill, dim3(nblk(n, 256)), dim3(256), 0, s, v, n, a);
}
// pseudo-random entries in (-1, 1) with every mantissa bit in play (autotune_ks: a representative operand -- a constant vector has
// one non-zero digit plane, the matrix pipe multiplies zeros and the ranking of the decompositions changes)
__global__ void k_fill_hash(double* __restrict__ v, int64_t n, uint64_t seed) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t z = (uint64_t)i * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    v[i] = (double)(int64_t)(z >> 11) * (1.0 / 4503599627370496.0) - 1.0;      // 53 bits -> [-1, 1)
}
void fill_hash(hipStream_t s, double* v, int64_t n, uint64_t seed) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_fill_hash, dim3(nblk(n, 256)), dim3(256), 0, s, v, n, seed);
}

void axpby(hipStream_t s, double* out, double a, const double* x, double b, const double* y, int64_t n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_axpby, dim3(nblk(n, 256)), dim3(256), 0, s, out, a, x, b, y, n);
}

// dst <- src on the stream, as a kernel: a device-to-device hipMemcpyAsync costs the HOST 10-15 us per call on this runtime (the
// gaps around __amd_rocclr_copyBuffer in profiles/r3_cfg5_gaps.txt: 90 us per VAMP iteration at config 5), a launch ~3
void copy(hipStream_t s, double* dst, const double* src, int64_t n) {
    if (n <= 0 || dst == src) return;
    hipLaunchKernelGGL(k_vcopy, dim3(nblk(n, 256)), dim3(256), 0, s, dst, src, n);
}
void loopback(hipStream_t s, double* buf, double* scratch, int64_t n, int delay_us) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_loop_out, dim3(nblk(n, 256)), dim3(256), 0, s, buf, scratch, n);
    if (delay_us > 0) hipLaunchKernelGGL(k_loop_wait, dim3(1), dim3(1), 0, s, delay_us > 5000 ? 5000 : delay_us);
    hipLaunchKernelGGL(k_loop_in, dim3(nblk(n, 256)), dim3(256), 0, s, buf, scratch, n);
}
void p_update(hipStream_t s, double* p, const double* z, double beta, int64_t n) {
    if (n <= 0) return;            // an empty shard (M == 0) steps through the solver with empty vectors
    hipLaunchKernelGGL(k_p_update, dim3(nblk(n, 256)), dim3(256), 0, s, p, z, beta, n);
}
void mask_copy(hipStream_t s, double* out, const double* in, const uint32_t* mask2, int64_t npad) {
    if (npad <= 0) return;
    hipLaunchKernelGGL(k_mask_copy, dim3(nblk(npad, 256)), dim3(256), 0, s, out, in, mask2, npad);
}

void dots(hipStream_t s, int K, const double* const* x, const double* const* y, int64_t n, double* partial,
          double* out) {
    DotArgs a{};
    for (int k = 0; k < K; k++) {
        a.x[k] = x[k];
        a.y[k] = y[k];
    }
    int nb = red_blocks(n, 256);
    hipLaunchKernelGGL(k_dots, dim3(nb), dim3(256), 0, s, a, K, n, partial);
    if (out) launch_finalize(s, partial, nb, K, out);
}

void dots_ex(hipStream_t s, int K, const double* const* xa, const double* const* xb, const double* const* ya,
             const double* const* yb, const int64_t* n, double* partial, double* out) {
    DotExArgs a{};
    int nb = 1;
    for (int k = 0; k < K; k++) {
        a.xa[k] = xa[k]; a.xb[k] = xb[k]; a.ya[k] = ya[k]; a.yb[k] = yb[k];
        a.n[k] = n[k];
        a.nb[k] = red_blocks(n[k], 256);
        if (a.nb[k] > nb) nb = a.nb[k];
    }
    hipLaunchKernelGGL(k_dots_ex, dim3(nb), dim3(256), 0, s, a, K, partial);
    launch_finalize(s, partial, nb, K, out);
}

int cgx_ab(hipStream_t s, int nsys, double* const* st, double* const* mu, const double* const* p, const double* const* v,
           double* const* r, const double* const* d, double* const* z, const double* const* dp, double* const* part,
           double* const* red, double diag, int64_t n, const double* const* dp_part, int dp_nb, double* const* az,
           const double* const* aw, int64_t npad) {
    CgxAB a{};
    for (int k = 0; k < nsys; k++) {
        a.st[k] = st[k]; a.mu[k] = mu[k]; a.p[k] = p[k]; a.v[k] = v[k]; a.r[k] = r[k]; a.d[k] = d[k]; a.z[k] = z[k];
        a.dp[k] = dp[k]; a.part[k] = part[k]; a.red[k] = red[k];
        a.dpp[k] = dp_part ? dp_part[k] : nullptr;
        a.az[k] = az ? az[k] : nullptr;
        a.aw[k] = aw ? aw[k] : nullptr;
    }
    a.npad = npad;
    a.dp_nb = dp_nb;
    const int nb = red_blocks(n, 256);
    hipLaunchKernelGGL(k_cgx_ab, dim3(nb, nsys), dim3(256), 0, s, a, diag, n);
    // (an inactive system's kernel slice returns at once and its stale partials are summed into a slot nobody reads)
    // one rank (dp_part != NULL): k_cgx_decide adds the block partials up itself; sharded: the sums are all-reduced first
    if (!dp_part)
        for (int k = 0; k < nsys; k++) hipLaunchKernelGGL(k_finalize, dim3(3), dim3(256), 0, s, part[k], nb, 3, red[k]);
    return nb;
}
void finalize(hipStream_t s, const double* partial, int nb, int K, double* out) {
    hipLaunchKernelGGL(k_finalize, dim3(K), dim3(256), 0, s, partial, nb, K, out);
}
// one CG_solverAAT step after d = A (A^T p) has arrived: everything but the read-back
// Three launches (two when the Ax epilogue has taken k_aat_dq along, one more with the search-direction update): every reduction is added up by the kernel that consumes it
// (sum_partials_256), A^T mu += alpha A^T p (at_p = the first half of this application) rides in k_aat_ab -- before
// k_aat_decide may clear ST_ACTIVE -- and p <- z + beta p is left to the k_prep_atx of the next pass when p_update == false.
void aat_step(hipStream_t s, double* st, double* mu, double* p, double* r, double* d, double* z, const double* diag, double tau,
              double gam2, int64_t n, double* partial, double* relres, int max_iter, double* mailbox,
              unsigned long long* flag, unsigned long long seq, double* at_acc, const double* at_p, int64_t m, const double* other_st,
              int* go, const int* ride, int* ride_mark, bool p_update, bool dq_done) {
    const int nb = red_blocks(n, 256);
    double* part_dq = partial;                       // <d,p>: nb doubles
    double* part_ab = partial + 2 * RED_BLOCKS;      // <r,z>, <r,r>: 2 nb doubles (clear of the sums its own blocks still read)
    // (dq_done: the Ax epilogue of the pass has left d = tau d + gam2 p and these partials already -- gvm::CgHook::dq_p)
    if (!dq_done) hipLaunchKernelGGL(k_aat_dq, dim3(nb), dim3(256), 0, s, d, p, tau, gam2, n, part_dq);
    hipLaunchKernelGGL(k_aat_ab, dim3(nb), dim3(256), 0, s, st, mu, p, r, d, z, diag, part_dq, nb, n, part_ab,
                       (at_acc && m > 0) ? at_acc : nullptr, at_p, m);
    hipLaunchKernelGGL(k_aat_decide, dim3(1), dim3(256), 0, s, st, part_ab, nb, relres, max_iter, mailbox, flag, seq, other_st, go,
                       ride, ride_mark);
    if (p_update) hipLaunchKernelGGL(k_p_update_st, dim3(nblk(n, 256)), dim3(256), 0, s, p, z, st, n);
}
void p_update_st(hipStream_t s, double* p, const double* z, const double* st, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_p_update_st, dim3(nblk(n, 256)), dim3(256), 0, s, p, z, st, n);
}
void ride_copy(hipStream_t s, double* out, const double* w0, const double* w1, const double* st0, const double* st1,
               const int* ride, int64_t n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_ride_copy, dim3(nblk(n, 256)), dim3(256), 0, s, out, w0, w1, st0, st1, ride, n);
}
void state_init(hipStream_t s, double* dst, const double* q) {
    StateInit a;
    for (int i = 0; i < gvm::ST_SIZE; i++) a.q[i] = q[i];
    hipLaunchKernelGGL(k_state_init, dim3(1), dim3(64), 0, s, dst, a);
}
void state_from_partials(hipStream_t s, double* dst, const double* q, const double* part_rz, int K_rz, int k_rz, const double* part_vv,
                         int64_t n, bool sqrt_norm) {
    StateInit a;
    for (int i = 0; i < gvm::ST_SIZE; i++) a.q[i] = q[i];
    const int nb = red_blocks(n, 256);        // both reductions ran over n entries
    hipLaunchKernelGGL(k_state_from_partials, dim3(1), dim3(256), 0, s, dst, a, part_rz, nb, K_rz, k_rz, part_vv, nb, sqrt_norm ? 1 : 0);
}
void state_from_scalars(hipStream_t s, double* dst, const double* q, const double* rz, const double* vv, bool sqrt_norm) {
    StateInit a;
    for (int i = 0; i < gvm::ST_SIZE; i++) a.q[i] = q[i];
    hipLaunchKernelGGL(k_state_from_scalars, dim3(1), dim3(64), 0, s, dst, a, rz, vv, sqrt_norm ? 1 : 0);
}
void set_ints(hipStream_t s, int* dst, int a, int b) { hipLaunchKernelGGL(k_set_ints, dim3(1), dim3(64), 0, s, dst, a, b); }
void go_from_states(hipStream_t s, int* dst, const double* st0, const double* st1, int b) {
    hipLaunchKernelGGL(k_go_from_states, dim3(1), dim3(64), 0, s, dst, st0, st1, b);
}
void ride_mark(hipStream_t s, const double* st0, const double* st1, int* ride) {
    hipLaunchKernelGGL(k_ride_mark, dim3(1), dim3(64), 0, s, st0, st1, ride);
}
void cgx_decide(hipStream_t s, int nsys, double* const* st, const double* const* red, double* const* relres, double gam2,
                int max_iter, int* go, double* mailbox, unsigned long long* flag, unsigned long long seq, int* ride,
                const double* other_st, const int* ride_report, const double* const* part, int part_nb) {
    CgxDecide a{};
    a.nsys = nsys;
    a.ride = ride;
    a.other_st = other_st;
    a.ride_report = ride_report;
    a.part_nb = part_nb;
    for (int k = 0; k < nsys; k++) {
        a.st[k] = st[k]; a.red[k] = red[k]; a.relres[k] = relres[k];
        a.part[k] = part ? part[k] : nullptr;
    }
    hipLaunchKernelGGL(k_cgx_decide, dim3(1), dim3(256), 0, s, a, gam2, max_iter, go, mailbox, flag, seq);
}

void cg_step_a(hipStream_t s, double* mu, const double* p, double alpha, const double* v, int64_t n, double* partial,
               double* out) {
    int nb = red_blocks(n, 256);
    hipLaunchKernelGGL(k_cg_a, dim3(nb), dim3(256), 0, s, mu, p, alpha, v, n, partial);
    launch_finalize(s, partial, nb, 1, out);
}

void cg_step_b(hipStream_t s, double* r, const double* d, double alpha, double diag, double* z, const double* mu,
               int64_t n, double* partial, double* out) {
    int nb = red_blocks(n, 256);
    hipLaunchKernelGGL(k_cg_b, dim3(nb), dim3(256), 0, s, r, d, alpha, diag, z, mu, n, partial);
    if (out) launch_finalize(s, partial, nb, 4, out);
}

void denoise(hipStream_t s, const double* r1, int64_t n, double gam1, const gv_prior& pr, double* x1, double* dd,
             double* partial, double* out) {
    int nb = red_blocks(n, 256);
    hipLaunchKernelGGL(k_denoise, dim3(nb), dim3(256), 0, s, r1, n, gam1, pr, x1, dd, partial);
    launch_finalize(s, partial, nb, 2, out);
}

void prior_estep(hipStream_t s, const double* r1, int64_t n, double gam1, double lambda, const gv_prior& pr,
                 double* partial, double* out) {
    int nb = red_blocks(n, 256);
    int K = 1 + 2 * (pr.L - 1);
    const int Lm1 = pr.L - 1;
#define GV_ESTEP(LCV) hipLaunchKernelGGL(k_prior_estep_reg<LCV>, dim3(nb), dim3(256), 0, s, r1, n, gam1, lambda, pr, partial)
    if (Lm1 <= 4) GV_ESTEP(4);
    else if (Lm1 <= 8) GV_ESTEP(8);
    else if (Lm1 <= 16) GV_ESTEP(16);
    else if (Lm1 <= 24) GV_ESTEP(24);
    else GV_ESTEP(GV_LMAX - 1);
#undef GV_ESTEP
    launch_finalize(s, partial, nb, K, out);
}

void pvals_test(hipStream_t s, const uint32_t* cnt, const double* mave, const double* msig, const double* sums4,
                const double* xself, double self_scale, const int* chrom, int ch, int64_t M, double* pvals) {
    if (M == 0) return;
    hipLaunchKernelGGL(k_pvals_test, dim3(nblk(M, 128)), dim3(128), 0, s, cnt, mave, msig, sums4, xself, self_scale, chrom,
                       ch, M, pvals);
}

void copy_bw(hipStream_t s, const double* src, double* dst, int64_t n) {
    hipLaunchKernelGGL(k_copy, dim3(256 * 16), dim3(256), 0, s, (const double2*)src, (double2*)dst, n / 2);
}
void read_bw(hipStream_t s, const void* src, int64_t blocks_per_wave, int64_t nwaves, unsigned int* sink, int perm) {
    hipLaunchKernelGGL(k_read_stream, dim3((unsigned)((nwaves + 3) / 4)), dim3(256), 0, s, (const rd_u32x4*)src,
                       blocks_per_wave, nwaves, sink, perm);
}

}  // namespace gvk
