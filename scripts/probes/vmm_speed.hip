// Development probe: does memory mapped through the HIP virtual-memory API (physical chunks of a chosen size behind an aligned
// virtual range) stream faster / more predictably than hipMalloc?  (docs/history/rounds1-3.md section 4.2.)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/probes/vmm_speed scripts/probes/vmm_speed.hip && scripts/probes/vmm_speed [chunk GiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_read(const u4* __restrict__ p, long n16, unsigned* sink) {
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
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    k_read<<<768 * 4, 256>>>((const u4*)p, (long)(bytes / 16), sink);
    (void)hipEventRecord(a);
    for (int i = 0; i < 3; i++) k_read<<<768 * 4, 256>>>((const u4*)p, (long)(bytes / 16), sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    return 3.0 * bytes / (ms * 1e-3) / 1e9;
}
int main(int argc, char** argv) {
    const size_t GiB = (size_t)1 << 30, total = 200 * GiB, chunk = (size_t)(argc > 1 ? atoi(argv[1]) : 1) * GiB;
    unsigned* sink;
    CK(hipMalloc(&sink, 64));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu KiB, recommended %zu KiB; chunk %zu GiB\n", gmin >> 10, grec >> 10, chunk / GiB);
    for (int round = 0; round < 2; round++) {
        // hipMalloc for comparison
        char* big = nullptr;
        CK(hipMalloc(&big, total));
        CK(hipMemset(big, 1, total));
        CK(hipDeviceSynchronize());
        printf("round %d  hipMalloc 200 GiB at %p:  first 100 GiB %.0f GB/s, second 100 GiB %.0f GB/s\n", round, (void*)big,
               gbps(big, 100 * GiB, sink), gbps(big + 100 * GiB, 100 * GiB, sink));
        CK(hipFree(big));
        // the same through the virtual-memory API
        void* va0 = nullptr;                      // (the alignment argument is not honoured: reserve more and align by hand)
        const size_t al = argc > 2 ? (size_t)atoi(argv[2]) * GiB : chunk;
        CK(hipMemAddressReserve(&va0, total + al, al, nullptr, 0));
        void* va = (void*)(((uintptr_t)va0 + al - 1) / al * al);
        std::vector<hipMemGenericAllocationHandle_t> hs;
        for (size_t off = 0; off < total; off += chunk) {
            hipMemGenericAllocationHandle_t h;
            CK(hipMemCreate(&h, chunk, &prop, 0));
            CK(hipMemMap((char*)va + off, chunk, 0, h, 0));
            hs.push_back(h);
        }
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        CK(hipMemSetAccess(va, total, &acc, 1));
        CK(hipMemset(va, 1, total));
        CK(hipDeviceSynchronize());
        printf("round %d  VMM 200 GiB at %p (%zu chunks):  first 100 GiB %.0f GB/s, second 100 GiB %.0f GB/s\n", round, va, hs.size(),
               gbps(va, 100 * GiB, sink), gbps((char*)va + 100 * GiB, 100 * GiB, sink));
        CK(hipMemUnmap(va, total));
        for (auto h : hs) CK(hipMemRelease(h));
        CK(hipMemAddressFree(va0, total + al));
    }
    return 0;
}
