// mall_reread.hip -- development probe (not part of the product): can a second read of a block that has just been streamed
// from HBM be served by the 256 MiB Infinity Cache at a rate that ADDS to the HBM stream?
//
// Motivation (docs/history/rounds1-3.md section 10): A^T (A p) = sum over blocks b of individuals of A_b^T (A_b p).  On the tile layout the
// two products read the SAME bytes, so a blocked operator would stream the matrix from HBM once per lmmse_mult and take the
// second read of every block from the last-level cache -- if that cache delivers.
//
// A persistent grid of 768 workgroups walks a buffer of T bytes in blocks of B bytes.  Per block every workgroup reads its own
// 1/768 slice (first touch: HBM) and re-reads the slice that a workgroup on ANOTHER XCD read `lag` blocks earlier (its own
// L2 cannot hold it; the Infinity Cache can, if 2-3 blocks fit).  No grid synchronisation: the probe only needs every byte read
// twice within a short window, not an order.  Reported: GB/s counting both reads, against the single-read stream of the same
// grid.  Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/mall_reread scripts/probes/mall_reread.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) { return NT ? __builtin_nontemporal_load(p) : *p; }

// slice = pieces of 4 KiB (256 lanes x 16 B); a workgroup reads `pps` pieces of a block, eight loads in flight per lane
template <bool NT1, bool NT2>
__global__ __launch_bounds__(256) void k_reread(const u32x4* __restrict__ buf, long nblocks, long pps, int lag, int second,
                                                unsigned int* __restrict__ sink) {
    const int w = blockIdx.x, nw = gridDim.x, tid = threadIdx.x;
    const int wo = (w + nw / 2 + 4) % nw;              // a workgroup whose index differs by 4 mod 8: another XCD
    u32x4 acc = {0, 0, 0, 0};
    const long bpieces = pps * nw;                     // pieces per block
    for (long b = 0; b < nblocks + lag; b++) {
        if (b < nblocks) {
            const u32x4* p = buf + ((b * bpieces + (long)w * pps) << 8) + tid;
            long i = 0;
            for (; i + 8 <= pps; i += 8) {
                u32x4 a[8];
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = ld<NT1>(p + ((i + k) << 8));
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= a[k];
            }
            for (; i < pps; i++) acc ^= ld<NT1>(p + (i << 8));
        }
        if (second && b >= lag) {
            const u32x4* p = buf + (((b - lag) * bpieces + (long)wo * pps) << 8) + tid;
            long i = 0;
            for (; i + 8 <= pps; i += 8) {
                u32x4 a[8];
#pragma unroll
                for (int k = 0; k < 8; k++) a[k] = ld<NT2>(p + ((i + k) << 8));
#pragma unroll
                for (int k = 0; k < 8; k++) acc ^= a[k];
            }
            for (; i < pps; i++) acc ^= ld<NT2>(p + (i << 8));
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[w * 256 + tid] = acc.x;   // keeps the loads alive
}

int main(int argc, char** argv) {
    const size_t T = (argc > 1 ? (size_t)atol(argv[1]) : 16) << 30;      // GiB walked
    const int nw = 768;
    u32x4* buf = nullptr;
    unsigned int* sink = nullptr;
    CHK(hipMalloc(&buf, T));
    CHK(hipMalloc(&sink, sizeof(unsigned int) * nw * 256));
    CHK(hipMemset(buf, 0x5a, T));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto run = [&](int variant, long nblocks, long pps, int lag, int second) -> double {
        auto launch = [&]() {
            switch (variant) {
                case 0: hipLaunchKernelGGL((k_reread<false, false>), dim3(nw), dim3(256), 0, 0, buf, nblocks, pps, lag, second, sink); break;
                case 1: hipLaunchKernelGGL((k_reread<true, false>), dim3(nw), dim3(256), 0, 0, buf, nblocks, pps, lag, second, sink); break;
                case 2: hipLaunchKernelGGL((k_reread<false, true>), dim3(nw), dim3(256), 0, 0, buf, nblocks, pps, lag, second, sink); break;
                default: hipLaunchKernelGGL((k_reread<true, true>), dim3(nw), dim3(256), 0, 0, buf, nblocks, pps, lag, second, sink); break;
            }
        };
        launch();
        (void)hipEventRecord(e0, 0);
        launch();
        launch();
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) return -1.0;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        return ms / 2;
    };
    const char* vname[4] = {"first default / second default", "first nt      / second default", "first default / second nt     ",
                            "first nt      / second nt     "};
    printf("buffer %.1f GiB, %d workgroups x 256 threads\n", T / 1073741824.0, nw);
    for (int mb : {12, 24, 48, 96, 192, 384}) {         // block size in MiB (multiples of 768 x 4 KiB = 3 MiB)
        const long pps = (long)mb * 1024 * 1024 / 4096 / nw;
        const size_t B = (size_t)pps * nw * 4096;
        const long nblocks = (long)(T / B);
        const double bytes = (double)nblocks * B;
        const double t1 = run(1, nblocks, pps, 1, 0);
        printf("block %4d MiB  single read (nt): %7.3f ms  %7.1f GB/s\n", mb, t1, bytes / t1 / 1e6);
        for (int lag : {1, 2}) {
            for (int v = 0; v < 4; v++) {
                const double t = run(v, nblocks, pps, lag, 1);
                printf("   lag %d  %s : %7.3f ms  %7.1f GB/s over both reads  (%.2f x the single-read time)\n", lag, vname[v], t,
                       2 * bytes / t / 1e6, t / t1);
            }
        }
        fflush(stdout);
    }
    (void)hipFree(buf);
    (void)hipFree(sink);
    return 0;
}
