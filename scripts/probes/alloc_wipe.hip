// alloc_wipe.hip -- development probe: hipMalloc of a large buffer right after a large buffer was freed takes seconds on this
// pool (the driver wipes / remaps the pages).  Does allocating the two resident layouts from two threads, or in pieces, shorten it?
// Build: hipcc --offload-arch=gfx950 -O2 -o scripts/probes/alloc_wipe scripts/probes/alloc_wipe.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void dirty(size_t bytes) {          // touch and free: what a previous process (or a previous shard) leaves behind
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("dirty: hipMalloc failed\n"); return; }
    (void)hipMemset(p, 1, bytes);
    (void)hipDeviceSynchronize();
    (void)hipFree(p);
}

int main(int argc, char** argv) {
    const size_t G = (size_t)1 << 30;
    const size_t each = (size_t)(argc > 1 ? atol(argv[1]) : 90) * G;       // bytes per layout
    (void)hipSetDevice(0);
    (void)hipFree(nullptr);
    {   // untouched memory first
        void *a = nullptr, *b = nullptr;
        double t = now();
        (void)hipMalloc(&a, each);
        (void)hipMalloc(&b, each);
        printf("fresh:      two sequential hipMalloc of %zu GiB: %.3f s\n", each / G, now() - t);
        (void)hipMemset(a, 1, each); (void)hipMemset(b, 1, each); (void)hipDeviceSynchronize();
        (void)hipFree(a); (void)hipFree(b);
    }
    for (int rep = 0; rep < 2; rep++) {
        {
            void *a = nullptr, *b = nullptr;
            double t = now();
            (void)hipMalloc(&a, each);
            double t1 = now();
            (void)hipMalloc(&b, each);
            printf("after free: two sequential hipMalloc: %.3f s (%.3f + %.3f)\n", now() - t, t1 - t, now() - t1);
            (void)hipMemset(a, 1, each); (void)hipMemset(b, 1, each); (void)hipDeviceSynchronize();
            (void)hipFree(a); (void)hipFree(b);
        }
        {
            void *a = nullptr, *b = nullptr;
            double t = now();
            std::thread th([&] { (void)hipSetDevice(0); (void)hipMalloc(&b, each); });
            (void)hipMalloc(&a, each);
            th.join();
            printf("after free: two hipMalloc from two threads: %.3f s\n", now() - t);
            (void)hipMemset(a, 1, each); (void)hipMemset(b, 1, each); (void)hipDeviceSynchronize();
            (void)hipFree(a); (void)hipFree(b);
        }
        {
            std::vector<void*> p(16, nullptr);
            double t = now();
            for (auto& q : p) (void)hipMalloc(&q, each / 8);
            printf("after free: sixteen sequential hipMalloc of 1/8 each: %.3f s\n", now() - t);
            for (auto& q : p) (void)hipMemset(q, 1, each / 8);
            (void)hipDeviceSynchronize();
            for (auto& q : p) (void)hipFree(q);
        }
    }
    // first-touch cost after a slow / fast allocation: is the wipe inside hipMalloc or deferred?
    {
        void* a = nullptr;
        double t = now();
        (void)hipMalloc(&a, each);
        double t1 = now();
        (void)hipMemset(a, 2, each);
        (void)hipDeviceSynchronize();
        double t2 = now();
        (void)hipMemset(a, 3, each);
        (void)hipDeviceSynchronize();
        printf("after free: hipMalloc %.3f s, first memset %.3f s, second memset %.3f s\n", t1 - t, t2 - t1, now() - t2);
        (void)hipFree(a);
    }
    return 0;
}
