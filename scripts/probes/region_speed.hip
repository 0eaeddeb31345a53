// Development probe: how fast does a plain read stream run over different 8 GiB regions of ONE large allocation, and of several
// smaller ones?  (docs/history/rounds1-3.md section 4.2: the placement of a 100 GB allocation moves the kernel that streams it by 1.5-3.5 %.)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/region_speed scripts/probes/region_speed.hip && scripts/probes/region_speed [GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_read(const u4* __restrict__ p, long n16, unsigned* sink) {
    // each wave walks its own contiguous 4 KiB blocks, four loads in flight
    const long waves = (long)gridDim.x * 4, wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const long blocks = n16 / 256, per = blocks / waves;
    const u4* q = p + wave * per * 256 + lane;
    u4 acc = {0, 0, 0, 0};
    for (long b = 0; b < per; b++) {
        u4 a0 = __builtin_nontemporal_load(q + b * 256), a1 = __builtin_nontemporal_load(q + b * 256 + 64),
           a2 = __builtin_nontemporal_load(q + b * 256 + 128), a3 = __builtin_nontemporal_load(q + b * 256 + 192);
        acc ^= a0 ^ a1 ^ a2 ^ a3;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u) sink[0] = 1;
}
static double gbps(const void* p, size_t bytes, unsigned* sink) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    k_read<<<768 * 4, 256>>>((const u4*)p, (long)(bytes / 16), sink);
    hipEventRecord(a);
    for (int i = 0; i < 3; i++) k_read<<<768 * 4, 256>>>((const u4*)p, (long)(bytes / 16), sink);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    hipEventDestroy(a); hipEventDestroy(b);
    return 3.0 * bytes / (ms * 1e-3) / 1e9;
}
int main(int argc, char** argv) {
    const size_t GiB = (size_t)1 << 30, total = (size_t)(argc > 1 ? atoi(argv[1]) : 200) * GiB, reg = 8 * GiB;
    unsigned* sink;
    (void)hipMalloc(&sink, 64);
    for (int round = 0; round < 2; round++) {
        char* big = nullptr;
        if (hipMalloc(&big, total) != hipSuccess) { printf("hipMalloc(%zu GiB) failed\n", total / GiB); return 1; }
        hipMemset(big, 1, total);
        hipDeviceSynchronize();
        printf("round %d: one allocation of %zu GiB at %p, read stream per 8 GiB region (GB/s):\n ", round, total / GiB, (void*)big);
        for (size_t off = 0; off + reg <= total; off += reg) printf(" %.0f", gbps(big + off, reg, sink));
        printf("\n  whole first 100 GiB: %.0f   whole second 100 GiB: %.0f\n", gbps(big, 100 * GiB, sink),
               total >= 200 * GiB ? gbps(big + 100 * GiB, 100 * GiB, sink) : 0.0);
        hipFree(big);
        // the same memory as separate allocations
        std::vector<char*> parts;
        for (size_t got = 0; got + 25 * GiB <= total; got += 25 * GiB) {
            char* q = nullptr;
            if (hipMalloc(&q, 25 * GiB) != hipSuccess) break;
            hipMemset(q, 1, 25 * GiB);
            parts.push_back(q);
        }
        hipDeviceSynchronize();
        printf("  %zu separate allocations of 25 GiB (GB/s each):", parts.size());
        for (char* q : parts) printf(" %.0f", gbps(q, 25 * GiB, sink));
        printf("\n");
        for (char* q : parts) hipFree(q);
    }
    return 0;
}
