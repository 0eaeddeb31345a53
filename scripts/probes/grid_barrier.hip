// grid_barrier.hip -- development probe (not part of the product): what does a device-wide barrier INSIDE a kernel cost on
// MI355X, against the launch boundary it would replace?
//
// Motivation (launch diet): between two streaming passes a CG step runs 4-5 dependent O(M) kernels; every launch
// boundary behind a dependent kernel costs ~5-7 us of idle chip plus the ramp of a tiny kernel.  A co-resident "glue" kernel
// that separates its phases by a barrier of its own only pays if that barrier is well below a launch boundary.
//
// Variants: 0 flat -- one counter, every block's thread 0 adds 1 and spins on it;  1 two-level -- a counter per XCD (the last
// arriver of an XCD adds 1 to the top counter), everyone spins on the top counter;  2 cooperative groups grid.sync().
// Each iteration: every thread stores `it` into its slot, barrier, reads the slot of a thread of a block 37 blocks away (on
// another XCD) and counts a mismatch -- so the number printed is for a barrier that makes data visible across XCDs.
// Reference: the same store / load split over 2 kernels per iteration, back to back on one stream (events around the chain).
// Every spin is bounded: a block that waits longer than ~1 s sets `bail` and every barrier falls through (no hang).
// Build: hipcc --offload-arch=gfx950 -O3 -o scripts/probes/grid_barrier scripts/probes/grid_barrier.hip
#include <hip/hip_cooperative_groups.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace cg = cooperative_groups;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Bar {
    unsigned int* top;        // [0] arrivals (flat) or XCD arrivals (two-level)
    unsigned int* xcd;        // 8 counters, 64 words apart
    unsigned int* bail;
    unsigned int nblocks;
    unsigned int per_xcd[8];  // blocks that report to each XCD counter
    unsigned int nxcd;        // XCD counters in use
};

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}

// epoch: number of barriers passed so far (same in every block)
template <int VARIANT>
__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned epoch, unsigned my_xcd) {
    __syncthreads();            // (every wave has waited for its own stores: vmcnt(0) before s_barrier)
    if (threadIdx.x == 0) {
        __threadfence();        // release at agent scope: L2 write-back so that other XCDs can see this XCD's stores
        unsigned target;
        if (VARIANT == 0) {
            __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            target = (epoch + 1) * b.nblocks;
        } else {
            const unsigned old = __hip_atomic_fetch_add(b.xcd + 64 * my_xcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == (epoch + 1) * b.per_xcd[my_xcd]) __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            target = (epoch + 1) * b.nxcd;
        }
        unsigned spins = 0;
        while (__hip_atomic_load(b.top, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1u << 24)) { *b.bail = 1; }
            if (__hip_atomic_load(b.bail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        }
    }
    __syncthreads();
    __threadfence();            // acquire: every wave drops stale lines of its L1 / this XCD's L2
}

template <int VARIANT>
__global__ __launch_bounds__(256) void k_bar(Bar b, int iters, unsigned int* __restrict__ slots, unsigned long long* clocks,
                                             unsigned int* mism, unsigned int* xcd_of_block) {
    const unsigned gid = blockIdx.x * 256 + threadIdx.x, total = gridDim.x * 256;
    const unsigned peer = ((blockIdx.x + 37) % gridDim.x) * 256 + threadIdx.x;
    unsigned my_xcd = 0;
    if (VARIANT == 1) my_xcd = blockIdx.x & 7;      // (a split of the contention; where the block physically runs does not matter)
    (void)xcd_of_block;
    unsigned bad = 0;
    cg::grid_group grid = cg::this_grid();
    unsigned long long t0 = 0;
    for (int it = 0; it < iters; it++) {
        if (it == 8 && gid == 0) t0 = wall_clock64();        // the first rounds warm the code and the counters' lines
        slots[gid] = (unsigned)it + 1;
        if (VARIANT == 2) grid.sync(); else grid_barrier<VARIANT>(b, 2 * it, my_xcd);
        if (__builtin_nontemporal_load(slots + peer) != (unsigned)it + 1) bad++;
        if (VARIANT == 2) grid.sync(); else grid_barrier<VARIANT>(b, 2 * it + 1, my_xcd);   // (nobody overwrites before all have read)
    }
    if (gid == 0) { clocks[0] = wall_clock64() - t0; }
    if (bad) atomicAdd(mism, bad);
    (void)total;
}

// which XCD does block i of a G-block launch land on?  (the dispatcher deals workgroups round-robin: i mod 8 -- verified here)
__global__ void k_where(unsigned int* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

__global__ __launch_bounds__(256) void k_store(unsigned int* __restrict__ slots, unsigned v) { slots[blockIdx.x * 256 + threadIdx.x] = v; }
__global__ __launch_bounds__(256) void k_check(const unsigned int* __restrict__ slots, unsigned v, unsigned int* mism) {
    const unsigned peer = ((blockIdx.x + 37) % gridDim.x) * 256 + threadIdx.x;
    if (slots[peer] != v) atomicAdd(mism, 1u);
}

template <int VARIANT>
static int run(int G, int iters, unsigned int* d_ctr, unsigned int* slots, unsigned long long* d_clk, unsigned int* d_mism,
               unsigned int* d_where, const std::vector<unsigned>& where) {
    CHK(hipMemset(d_ctr, 0, 4096 * sizeof(unsigned)));
    CHK(hipMemset(d_mism, 0, sizeof(unsigned)));
    Bar b{};
    b.top = d_ctr; b.xcd = d_ctr + 256; b.bail = d_ctr + 2048; b.nblocks = (unsigned)G;
    for (int k = 0; k < 8; k++) b.per_xcd[k] = 0;
    for (int i = 0; i < G; i++) b.per_xcd[i & 7]++;
    (void)where;
    b.nxcd = 0;
    for (int k = 0; k < 8; k++) if (b.per_xcd[k]) b.nxcd++;
    // (two-level needs contiguous XCD ids among those in use; counters of unused XCDs never fire)
    void* args[] = {&b, &iters, &slots, &d_clk, &d_mism, &d_where};
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    CHK(hipEventRecord(e0, 0));
    CHK(hipLaunchCooperativeKernel((const void*)k_bar<VARIANT>, dim3(G), dim3(256), args, 0, 0));
    CHK(hipEventRecord(e1, 0));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long clk = 0;
    unsigned mism = 0, bail = 0;
    CHK(hipMemcpy(&clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&mism, d_mism, sizeof(mism), hipMemcpyDeviceToHost));
    CHK(hipMemcpy(&bail, d_ctr + 2048, sizeof(bail), hipMemcpyDeviceToHost));
    printf("  variant %d  G %4d : %.2f us per barrier (device clock, %d barriers), kernel %.3f ms, mismatches %u%s\n", VARIANT, G,
           (double)clk / 100.0 / (2.0 * (iters - 8)), 2 * (iters - 8), ms, mism, bail ? "  BAILED (a spin timed out)" : "");
    CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    unsigned int *d_ctr, *slots, *d_mism, *d_where;
    unsigned long long* d_clk;
    CHK(hipMalloc(&d_ctr, 4096 * sizeof(unsigned)));
    CHK(hipMalloc(&slots, 1024 * 256 * sizeof(unsigned)));
    CHK(hipMalloc(&d_clk, 64));
    CHK(hipMalloc(&d_mism, 64));
    CHK(hipMalloc(&d_where, 1024 * sizeof(unsigned)));
    hipDeviceProp_t pr;
    CHK(hipGetDeviceProperties(&pr, 0));
    printf("%s, %d CUs, cooperative launch %d\n", pr.gcnArchName, pr.multiProcessorCount, pr.cooperativeLaunch);
    for (int G : {32, 64, 128, 256, 512}) {
        std::vector<unsigned> where(G);
        hipLaunchKernelGGL(k_where, dim3(G), dim3(64), 0, 0, d_where);
        CHK(hipMemcpy(where.data(), d_where, G * sizeof(unsigned), hipMemcpyDeviceToHost));
        int rr = 0;
        for (int i = 0; i < G; i++) rr += (where[i] == (unsigned)(i & 7));
        printf("G %d: blocks on XCD (i mod 8): %d of %d\n", G, rr, G);
        if (run<0>(G, iters, d_ctr, slots, d_clk, d_mism, d_where, where)) return 1;
        if (run<1>(G, iters, d_ctr, slots, d_clk, d_mism, d_where, where)) return 1;
        if (run<2>(G, iters, d_ctr, slots, d_clk, d_mism, d_where, where)) return 1;
        // reference: launch boundaries (2 kernels per iteration, dependent, one stream)
        hipEvent_t e0, e1;
        CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
        CHK(hipMemset(d_mism, 0, sizeof(unsigned)));
        for (int it = 0; it < 8; it++) {
            hipLaunchKernelGGL(k_store, dim3(G), dim3(256), 0, 0, slots, (unsigned)it);
            hipLaunchKernelGGL(k_check, dim3(G), dim3(256), 0, 0, slots, (unsigned)it, d_mism);
        }
        CHK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; it++) {
            hipLaunchKernelGGL(k_store, dim3(G), dim3(256), 0, 0, slots, (unsigned)it + 100);
            hipLaunchKernelGGL(k_check, dim3(G), dim3(256), 0, 0, slots, (unsigned)it + 100, d_mism);
        }
        CHK(hipEventRecord(e1, 0));
        CHK(hipEventSynchronize(e1));
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        unsigned mism = 0;
        CHK(hipMemcpy(&mism, d_mism, sizeof(mism), hipMemcpyDeviceToHost));
        printf("  launches   G %4d : %.2f us per dependent launch (%d launches), mismatches %u\n", G, ms * 1000.0 / (2.0 * iters), 2 * iters, mism);
        CHK(hipEventDestroy(e0)); CHK(hipEventDestroy(e1));
    }
    return 0;
}
