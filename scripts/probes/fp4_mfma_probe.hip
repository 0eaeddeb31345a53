// Probe (development, not product): can the 2-bit -> operand expansion of the streaming kernels be made cheaper by feeding the
// genotypes to the matrix pipe as FP4 (E2M1) instead of bytes?   v_mfma_scale_f32_16x16x128_f8f6f4, A = fp4 genotype planes,
// B = fp8 (E4M3) digits in [-8, 8] (base-16 balanced digits: the largest digit set E4M3 holds exactly), fp32 accumulation (exact
// below 2^24), scales 2^0.
//   (1) exactness and operand layout: one 16 x 128 tile of codes against the integer reference;
//   (2) what the expansion costs: expand_fp4() below turns the 32 codes a lane holds for one MFMA (2 dwords of 16 two-bit codes) into
//       the 4 + 4 dwords of two fp4 planes; the vector instructions it compiles to are counted from the disassembly by
//       scripts/probes/fp4_mfma_probe.sh and held against the byte expansion of the shipped kernels (gv_mfma.hip: 11-13 per dword).
// Build + run: bash scripts/probes/fp4_mfma_probe.sh   (on the GPU box)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));

// 16 two-bit codes (one dword) -> 16 nibbles (two dwords), nibble = code: the classic bit spread, three shift-or-mask stages per half
__device__ __forceinline__ void spread16(uint32_t w, uint32_t& lo, uint32_t& hi) {
    uint32_t a = w & 0xFFFFu, b = w >> 16;
    a = (a | (a << 8)) & 0x00FF00FFu;  b = (b | (b << 8)) & 0x00FF00FFu;
    a = (a | (a << 4)) & 0x0F0F0F0Fu;  b = (b | (b << 4)) & 0x0F0F0F0Fu;
    a = (a | (a << 2)) & 0x33333333u;  b = (b | (b << 2)) & 0x33333333u;
    lo = a; hi = b;
}
// Two fp4 planes of the codes f in {0,1,2,3}:  P2 = nibble f      -> E2M1 values {0, 0.5, 1, 1.5}
//                                               P1 = nibble f << 1 -> E2M1 values {0, 1, 2, 4}
// Any two independent planes serve (the operand vectors absorb the change of basis: a' = -3 P1 + 8 P2, present = 1 - P1 + 2 P2).
__device__ __forceinline__ void expand_fp4(uint32_t w0, uint32_t w1, uint32_t (&p1)[4], uint32_t (&p2)[4]) {
    spread16(w0, p2[0], p2[1]);
    spread16(w1, p2[2], p2[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) p1[i] = p2[i] << 1;
}

// (2) the expansion in a loop the compiler cannot fold away: reads code dwords, writes plane checksums
__global__ void k_expand_only(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int n) {
    uint32_t acc = 0;
    for (int i = threadIdx.x; i + 64 < n; i += 128) {
        uint32_t p1[4], p2[4];
        expand_fp4(in[i], in[i + 64], p1, p2);
        // (an MFMA would consume the eight dwords; here they are folded so that every one of them is live)
        acc ^= p1[0] + 3 * p1[1] + 5 * p1[2] + 7 * p1[3] + 11 * p2[0] + 13 * p2[1] + 17 * p2[2] + 19 * p2[3];
    }
    out[threadIdx.x] = acc;
}

// (1) one MFMA on raw per-lane operands (host-packed): A 4 dwords of fp4 per lane, B 8 dwords of fp8 per lane
__global__ void k_raw(const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, float* __restrict__ C, int nmfma) {
    const int l = threadIdx.x;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const int one = 0x7F7F7F7F;      // E8M0 scale 2^0 in every byte
    for (int m = 0; m < nmfma; m++) {
        v8i a = {(int)A[(m * 64 + l) * 4 + 0], (int)A[(m * 64 + l) * 4 + 1], (int)A[(m * 64 + l) * 4 + 2], (int)A[(m * 64 + l) * 4 + 3], 0, 0, 0, 0};
        v8i b;
        for (int d = 0; d < 8; d++) b[d] = (int)B[(m * 64 + l) * 8 + d];
        // cbsz = 4: A is fp4 (E2M1); blgp = 0: B is fp8 (E4M3)
        acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 4, 0, 0, one, 0, one);
    }
    for (int reg = 0; reg < 4; reg++) C[l * 4 + reg] = acc[reg];
}

static uint8_t e4m3_of_int(int v) {      // integers -8 .. 8, OCP e4m3fn
    static const uint8_t mag[9] = {0x00, 0x38, 0x40, 0x44, 0x48, 0x4A, 0x4C, 0x4E, 0x50};
    return (uint8_t)((v < 0 ? 0x80 : 0x00) | mag[v < 0 ? -v : v]);
}

static uint32_t *dA, *dB; static float* dC;
static std::vector<float> run_raw(const std::vector<uint32_t>& A, const std::vector<uint32_t>& B, int nmfma) {
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_raw, dim3(1), dim3(64), 0, 0, dA, dB, dC, nmfma);
    std::vector<float> C(256);
    if (hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("kernel failed\n"); exit(1); }
    return C;
}

int main() {
    hipMalloc(&dA, 2 * 64 * 4 * 4); hipMalloc(&dB, 2 * 64 * 8 * 4); hipMalloc(&dC, 256 * 4);
    // ---- layout discovery.  A: one nibble (lane LA, nibble ja) = 0x2 (E2M1 1.0); B: one byte (lane LB, byte jb) = 0x38 (E4M3 1.0).
    // The single product lands in C iff the two positions name the same k; the C slot gives (row of LA, column of LB).
    int kmapA[64][32], kmapB[64][32];      // (lane, element) -> k
    for (int l = 0; l < 64; l++) for (int j = 0; j < 32; j++) kmapA[l][j] = kmapB[l][j] = -1;
    {
        // step 1: B all ones -> C[row] = number of A ones in that row: where does lane LA's nibble land (row)?
        std::vector<uint32_t> A(64 * 4, 0), B(64 * 8, 0x38383838u);
        int rowA[64], colB[64];
        for (int LA = 0; LA < 64; LA++) {
            std::fill(A.begin(), A.end(), 0u);
            A[LA * 4] = 0x2u;
            auto C = run_raw(A, B, 1);
            rowA[LA] = -1;
            for (int i = 0; i < 256; i++) if (C[i] != 0.f) { int lane = i / 4, reg = i % 4; rowA[LA] = 4 * (lane >> 4) + reg; break; }
        }
        std::vector<uint32_t> A1(64 * 4, 0x22222222u);
        for (int LB = 0; LB < 64; LB++) {
            std::fill(B.begin(), B.end(), 0u);
            B[LB * 8] = 0x38u;
            auto C = run_raw(A1, B, 1);
            colB[LB] = -1;
            for (int i = 0; i < 256; i++) if (C[i] != 0.f) { colB[LB] = (i / 4) & 15; break; }
        }
        bool okr = true, okc = true;
        for (int l = 0; l < 64; l++) { okr &= rowA[l] == (l & 15); okc &= colB[l] == (l & 15); }
        printf("A: lane l holds row l & 15: %s;  B: lane l holds column l & 15: %s\n", okr ? "yes" : "NO", okc ? "yes" : "NO");
        // step 2: k of every A element against B's elements of lane group g (64 x 32 x ... too many launches: use the structure --
        // for A element (LA = 16 ga, ja) find the B element (LB = 16 gb, jb) that pairs with it)
        for (int ga = 0; ga < 4; ga++)
            for (int ja = 0; ja < 32; ja++) {
                std::fill(A.begin(), A.end(), 0u);
                A[(16 * ga) * 4 + ja / 8] = 0x2u << (4 * (ja % 8));
                // B: byte jb of lane group gb carries the value 1 + jb / 64 + gb / 4 ... simpler: B element (gb, jb) = distinct power of two
                // is impossible (128 elements); so: B = 1.0 everywhere in group gb only, then byte jb only
                int gb_hit = -1, jb_hit = -1;
                for (int gb = 0; gb < 4 && gb_hit < 0; gb++) {
                    std::fill(B.begin(), B.end(), 0u);
                    for (int d = 0; d < 8; d++) B[(16 * gb) * 8 + d] = 0x38383838u;
                    auto C = run_raw(A, B, 1);
                    if (C[0] != 0.f) gb_hit = gb;      // row 0 (lane 0, reg 0), column 0
                }
                for (int jb = 0; jb < 32 && gb_hit >= 0 && jb_hit < 0; jb++) {
                    std::fill(B.begin(), B.end(), 0u);
                    B[(16 * gb_hit) * 8 + jb / 4] = 0x38u << (8 * (jb % 4));
                    auto C = run_raw(A, B, 1);
                    if (C[0] != 0.f) jb_hit = jb;
                }
                kmapA[16 * ga][ja] = gb_hit * 32 + jb_hit;      // in units of "B's (group, byte)" = B's own k order, taken as the reference
            }
        bool ident = true;
        for (int ga = 0; ga < 4; ga++) for (int ja = 0; ja < 32; ja++) ident &= kmapA[16 * ga][ja] == 32 * ga + ja;
        printf("A nibble j of lane group g pairs with B byte j of lane group g (k = 32 g + j on both sides): %s\n", ident ? "yes" : "NO");
        if (!ident) {
            for (int ga = 0; ga < 4; ga++) {
                printf("  A group %d nibbles pair with B (group*32 + byte):", ga);
                for (int ja = 0; ja < 32; ja++) printf(" %d", kmapA[16 * ga][ja]);
                printf("\n");
            }
        }
    }
    // ---- exactness with the discovered pairing: two MFMAs (plane P1 with digits d1, plane P2 with digits d2)
    std::vector<int> f(16 * 128), d1(128 * 16), d2(128 * 16);
    srand(7);
    for (auto& x : f) x = rand() & 3;
    for (int i = 0; i < 128 * 16; i++) { d1[i] = rand() % 17 - 8; d2[i] = rand() % 17 - 8; }
    std::vector<uint32_t> A(2 * 64 * 4, 0), B(2 * 64 * 8, 0);
    for (int l = 0; l < 64; l++) {
        const int r = l & 15, g = l >> 4;
        for (int ja = 0; ja < 32; ja++) {
            const int kb = kmapA[16 * g][ja];                      // B-side k this nibble pairs with
            const int code = f[r * 128 + kb];
            A[(0 * 64 + l) * 4 + ja / 8] |= (uint32_t)(code << 1) << (4 * (ja % 8));      // P1: nibble f << 1 -> {0, 1, 2, 4}
            A[(1 * 64 + l) * 4 + ja / 8] |= (uint32_t)code << (4 * (ja % 8));             // P2: nibble f      -> {0, .5, 1, 1.5}
        }
        for (int jb = 0; jb < 32; jb++) {
            const int k = 32 * g + jb;
            B[(0 * 64 + l) * 8 + jb / 4] |= (uint32_t)e4m3_of_int(d1[k * 16 + r]) << (8 * (jb % 4));
            B[(1 * 64 + l) * 8 + jb / 4] |= (uint32_t)e4m3_of_int(d2[k * 16 + r]) << (8 * (jb % 4));
        }
    }
    auto C = run_raw(A, B, 2);
    const double P1[4] = {0, 1, 2, 4}, P2[4] = {0, 0.5, 1, 1.5};
    int bad = 0;
    double maxabs = 0;
    for (int r = 0; r < 16; r++)
        for (int c = 0; c < 16; c++) {
            double ref = 0;
            for (int k = 0; k < 128; k++) ref += P1[f[r * 128 + k]] * d1[k * 16 + c] + P2[f[r * 128 + k]] * d2[k * 16 + c];
            const float got = C[(16 * (r >> 2) + c) * 4 + (r & 3)];      // C/D: col = lane & 15, row = 4 (lane >> 4) + reg
            if (ref != (double)got) { if (bad < 5) printf("mismatch C[%d][%d] = %g, reference %g\n", r, c, got, ref); bad++; }
            if (fabs(ref) > maxabs) maxabs = fabs(ref);
        }
    printf("fp4 (A) x fp8 e4m3 digits in [-8, 8] (B), 16x16x128, fp32 accumulation, against the integer reference: %d of 256 entries differ "
           "(largest |entry| %.1f): %s\n", bad, maxabs, bad ? "NOT exact" : "EXACT");
    // keep k_expand_only alive for the disassembly
    uint32_t* dout; hipMalloc(&dout, 64 * 4);
    hipLaunchKernelGGL(k_expand_only, dim3(1), dim3(64), 0, 0, dA, dout, 128);
    hipDeviceSynchronize();
    return bad ? 2 : 0;
}
