// gv_capi.hip -- the C ABI of include/gvamp.h over the gfx950 kernels.  No CPU fallback anywhere: every
// compute entry point launches HIP kernels on the context's stream or fails.
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <condition_variable>
#include <cctype>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>

#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include "gv_internal.h"
#include "gv_tune_builtin.h"

namespace gvi {

thread_local std::string g_create_err;

int fail(gv_ctx* c, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (c) {
        c->err = buf;
        c->pub_armed = false;          // an armed read-back whose reduction never ran must not leave read_scalars spinning
        gvk::disarm_publish();
    } else
        g_create_err = buf;
    return 1;
}

inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

int vec_new(gv_ctx* c, int space, gv_vec** out) {
    NEED(c, c->N > 0, "gv_set_dims must be called first");
    gv_vec* v = new gv_vec();
    v->ctx = c;
    v->space = space;
    v->len = (space == GV_SPACE_M) ? c->M : 4 * c->mbytes;
    v->cap = (space == GV_SPACE_M) ? (c->M > 0 ? c->M : 1) : c->npad;
    v->d = nullptr;
    hipError_t e = hipMalloc(&v->d, sizeof(double) * v->cap);
    if (e != hipSuccess) {
        delete v;
        return fail(c, "hipMalloc(%lld doubles) failed: %s", (long long)v->cap, hipGetErrorString(e));
    }
    e = hipMemsetAsync(v->d, 0, sizeof(double) * v->cap, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(v->d);
        delete v;
        return fail(c, "hipMemsetAsync failed: %s", hipGetErrorString(e));
    }
    c->live_vecs.insert(v);
    *out = v;
    return 0;
}

// every gv_vec of a context goes through vec_new / vec_del, so gv_destroy can release what a caller never freed
void vec_del(gv_ctx* c, gv_vec* v) {
    if (!v) return;
    c->live_vecs.erase(v);
    if (v->owns) (void)hipFree(v->d);
    delete v;
}

// w_n and w_n2 (the N-space results of a two-vector Ax) share one allocation, w_n2 right behind w_n, so that a sharded
// job all-reduces both in ONE call (ax2_device)
int ensure_w2(gv_ctx* c) {
    if (c->w_n2) return 0;
    gv_vec* v = new gv_vec();
    *v = *c->w_n;
    v->d = c->w_n->d + c->npad;
    v->owns = false;
    c->live_vecs.insert(v);
    c->w_n2 = v;
    return 0;
}
int ensure_work(gv_ctx* c) {
    if (c->w_n && c->cg_d) return 0;
    // all or nothing: a half-built set must never be published (the entry points test w_n / cg_d and then use all five)
    gv_vec* made[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto undo = [&]() {
        for (gv_vec* v : made) vec_del(c, v);
        c->w_n = c->cg_r = c->cg_z = c->cg_p = c->cg_d = nullptr;
        return 1;
    };
    if (c->w_n) { vec_del(c, c->w_n); c->w_n = nullptr; }
    if (vec_new(c, GV_SPACE_N, &made[0])) return undo();
    double* both = nullptr;                                    // twice the room: w_n2 lives in the second half
    if (hipMalloc(&both, sizeof(double) * 2 * c->npad) != hipSuccess ||
        hipMemsetAsync(both, 0, sizeof(double) * 2 * c->npad, c->stream) != hipSuccess) {
        (void)hipGetLastError();
        if (both) (void)hipFree(both);
        undo();
        return fail(c, "ensure_work: no room for the N-space scratch vectors (%lld doubles)", (long long)(2 * c->npad));
    }
    (void)hipFree(made[0]->d);
    made[0]->d = both;
    for (int k = 1; k < 5; k++)
        if (vec_new(c, GV_SPACE_M, &made[k])) return undo();
    c->w_n = made[0]; c->cg_r = made[1]; c->cg_z = made[2]; c->cg_p = made[3]; c->cg_d = made[4];
    return 0;
}

// read K scalars produced by a reduction launcher back to the host.  Mailbox form (default): a one-block kernel writes
// them into mapped coherent host memory and then a sequence number; the host spins on that number -- no copy engine,
// no interrupt-driven stream synchronisation (measured: ~55 us of GPU idle per read-back with hipMemcpyAsync +
// hipStreamSynchronize, a CG step has three).  When the flag arrives every earlier kernel of the stream has finished.
void arm_scalars(gv_ctx* c) {
    if (!c->use_mbox || !c->pub_counter) return;
    c->pub_seq = ++c->mbox_seq;
    c->pub_armed = true;
    gvk::arm_publish(c->mbox_dev, reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK), c->pub_seq, c->pub_counter);
}
int read_scalars(gv_ctx* c, int K, double* out) {
    if (c->use_mbox && K <= RED_MAXK) {
        unsigned long long* flag = reinterpret_cast<unsigned long long*>(c->mbox + RED_MAXK);
        unsigned long long* flag_dev = reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK);
        unsigned long long seq;
        if (c->pub_armed) {            // the reduction's own finalisation publishes (arm_scalars)
            c->pub_armed = false;
            seq = c->pub_seq;
        } else {
            seq = ++c->mbox_seq;
            gvk::publish(c->stream, c->red_out, K, c->mbox_dev, flag_dev, seq);
        }
        KCHK(c);
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long spins = 0;
        while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != seq) {
            if ((++spins & 0xFFFF) == 0 &&
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 20.0) {
                // the kernel before it may have faulted: let the runtime say so instead of spinning for ever
                HIPCHK(c, hipStreamSynchronize(c->stream));
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) break;
                return fail(c, "read_scalars: the device never published the scalars");
            }
        }
        memcpy(out, c->mbox, sizeof(double) * K);
        return 0;
    }
    c->pub_armed = false;
    gvk::disarm_publish();
    HIPCHK(c, hipMemcpyAsync(c->host_pin, c->red_out, sizeof(double) * K, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    memcpy(out, c->host_pin, sizeof(double) * K);
    return 0;
}

bool is_multi(const gv_ctx* c);
int comm_allreduce(gv_ctx* c, double* dev, size_t n);
int comm_allreduce_on(gv_ctx* c, double* dev, size_t n, hipStream_t stream);

// K scalars a reduction launcher left in red_out, summed over the ranks (utilities.cpp:203) and read back: the all-reduce
// runs on the device buffer itself, so a CG scalar costs one stream synchronisation whether or not the job is sharded
int read_scalars_global(gv_ctx* c, int K, double* out, bool multi) {
    if (multi && comm_allreduce(c, c->red_out, K)) return 1;
    return read_scalars(c, K, out);
}

// MPI_Allreduce(SUM, MPI_DOUBLE) of K host scalars (utilities.cpp:203): device round trip through RCCL
int allreduce_scalars(gv_ctx* c, double* buf, int K) {
    if (!is_multi(c)) return 0;
    NEED(c, K <= RED_MAXK, "allreduce_scalars: too many scalars");
    memcpy(c->host_pin, buf, sizeof(double) * K);
    HIPCHK(c, hipMemcpyAsync(c->red_out, c->host_pin, sizeof(double) * K, hipMemcpyHostToDevice, c->stream));
    if (comm_allreduce(c, c->red_out, K)) return 1;
    return read_scalars(c, K, buf);
}

// ---- in-process communicator: nranks contexts of ONE process (threads) behave like nranks MPI ranks.  Sums in rank
// order on the host (deterministic).  For tests of the sharded algorithm on a single GPU; production uses RCCL.
struct LocalGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    std::vector<const double*> slots;
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        const long g = gen;
        if (++arrived == n) {
            arrived = 0;
            gen++;
            cv.notify_all();
        } else
            cv.wait(lk, [&] { return gen != g; });
    }
};
std::mutex g_groups_mu;
std::map<int, std::shared_ptr<LocalGroup>> g_groups;

// (force_multi: a one-rank context made to take the sharded branches -- gv_debug_force_multi)
bool is_multi(const gv_ctx* c) { return c->force_multi != 0 || (c->nranks > 1 && (c->comm || c->local || c->cb)); }

// The exchange of a forced one-rank job: asynchronous and in-stream like RCCL's, with nothing for the host to wait on.  The
// loop-back moves the message through scratch and poisons it in between, so a consumer that is not ordered behind the exchange
// (a missing event edge between the side stream and the context's stream, a kernel enqueued ahead of its all-reduce) reads NaNs.
static int forced_allreduce(gv_ctx* c, double* dev, size_t n, hipStream_t stream) {
    if ((c->force_multi & 2) && c->comm)
        NCCLCHK(c, ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, c->comm, stream));
    if (c->force_multi & 1) {
        const int q = (stream == c->stream) ? 0 : 1;
        if (c->loop_cap[q] < n) {
            // (grown once per stream to the largest message of a job, w_n | w_n2; the wait is the test hook's, not the product's)
            HIPCHK(c, hipStreamSynchronize(stream));
            if (c->loop_buf[q]) (void)hipFree(c->loop_buf[q]);
            c->loop_buf[q] = nullptr;
            c->loop_cap[q] = 0;
            const size_t cap = n > (size_t)(2 * c->npad + 64) ? n : (size_t)(2 * c->npad + 64);
            HIPCHK(c, hipMalloc(&c->loop_buf[q], sizeof(double) * cap));
            c->loop_cap[q] = cap;
        }
        gvk::loopback(stream, dev, c->loop_buf[q], (int64_t)n, c->loop_delay_us);
        KCHK(c);
    }
    return 0;
}

// SUM all-reduce of n doubles living on the device, on the context's stream
int comm_allreduce(gv_ctx* c, double* dev, size_t n) { return comm_allreduce_on(c, dev, n, c->stream); }
int comm_allreduce_on(gv_ctx* c, double* dev, size_t n, hipStream_t stream) {
    if (!is_multi(c)) return 0;
    if (c->force_multi) return forced_allreduce(c, dev, n, stream);
    if (c->comm) {
        NCCLCHK(c, ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, c->comm, stream));
        return 0;
    }
    if (c->cb) {   // caller's transport (gv_comm_init_callback): host round trip
        c->local_buf.resize(n);
        HIPCHK(c, hipMemcpyAsync(c->local_buf.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        if (c->cb(c->cb_user, c->local_buf.data(), n) != 0) return fail(c, "comm_allreduce: the all-reduce callback failed");
        HIPCHK(c, hipMemcpyAsync(dev, c->local_buf.data(), sizeof(double) * n, hipMemcpyHostToDevice, stream));
        HIPCHK(c, hipStreamSynchronize(stream));
        return 0;
    }
    LocalGroup* g = static_cast<LocalGroup*>(c->local);
    c->local_buf.resize(n);
    HIPCHK(c, hipMemcpyAsync(c->local_buf.data(), dev, sizeof(double) * n, hipMemcpyDeviceToHost, stream));
    HIPCHK(c, hipStreamSynchronize(stream));
    g->slots[c->rank] = c->local_buf.data();
    g->barrier();
    std::vector<double> sum(n, 0.0);
    for (int r = 0; r < g->n; r++) {
        const double* s = g->slots[r];
        for (size_t i = 0; i < n; i++) sum[i] += s[i];
    }
    g->barrier();
    c->local_buf.swap(sum);
    HIPCHK(c, hipMemcpyAsync(dev, c->local_buf.data(), sizeof(double) * n, hipMemcpyHostToDevice, stream));
    HIPCHK(c, hipStreamSynchronize(stream));
    return 0;
}

// Host <-> device transfers of whole vectors (the std::vector<double> arguments and results of data::Ax / data::ATx,
// data.hpp:117-121) through a pinned staging buffer of XFER_BYTES: a pageable user buffer costs an 8 MB copy ~5 ms on this
// runtime, a pinned one ~0.15 ms.  What is left is the host's own memcpy between the caller's buffer and the staging buffer --
// 4 MB take ~0.33 ms on one core, 15 % of a 2 ms matvec at config-2 size -- so that copy is shared among a few helper threads
// (CopyPool: GV_XFER_THREADS helpers, default 3, 0 = none; they sleep between calls and spin briefly after one).
// Both functions return with the data in place.
constexpr size_t XFER_BYTES = (size_t)8 << 20;
namespace {
class CopyPool {
    struct Job { char* dst; const char* src; size_t n; };
    std::vector<std::thread> th;
    std::vector<Job> jobs;
    std::unique_ptr<std::atomic<int>[]> taken;      // 1: somebody (the helper it was meant for, or the caller) has claimed job i
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<unsigned long> gen{0};
    std::atomic<int> pending{0};
    bool stop = false;
    pid_t owner = 0;
    void run(int i) {
        const Job j = jobs[i];
        if (j.n) memcpy(j.dst, j.src, j.n);
        pending.fetch_sub(1, std::memory_order_acq_rel);
    }
    void work(int id) {
        unsigned long seen = 0;
        for (;;) {
            // spin a little for the next job (back-to-back matvecs), then sleep
            bool got = false;
            for (int i = 0; i < 2000 && !got; i++) {
                got = gen.load(std::memory_order_acquire) != seen;
#if defined(__x86_64__)
                if (!got) __builtin_ia32_pause();
#endif
            }
            if (!got) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return stop || gen.load(std::memory_order_acquire) != seen; });
                if (stop) return;
            }
            seen = gen.load(std::memory_order_acquire);
            if (stop) return;
            int expect = 0;
            if (taken[id].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) run(id);   // else the caller took it
        }
    }
public:
    static CopyPool& get() {
        static CopyPool* p = new CopyPool();      // leaked on purpose: no destructor order games at process exit
        return *p;
    }
    CopyPool() {
        // helpers = min(3, CPUs this process may run on - 1); none when it has two CPUs or fewer (a cgroup of one core, several
        // ranks pinned to few cores: the caller would spin on the core its helpers need).  GV_XFER_THREADS overrides (0 = none).
        int n = 3;
        cpu_set_t cs;
        if (sched_getaffinity(0, sizeof(cs), &cs) == 0) {
            const int ncpu = CPU_COUNT(&cs);
            n = ncpu <= 2 ? 0 : (ncpu - 1 < 3 ? ncpu - 1 : 3);
        }
        if (const char* e = getenv("GV_XFER_THREADS")) n = atoi(e) < 0 ? 0 : (atoi(e) > 15 ? 15 : atoi(e));
        owner = getpid();
        jobs.assign(n, Job{nullptr, nullptr, 0});
        taken.reset(new std::atomic<int>[n > 0 ? n : 1]);
        for (int i = 0; i < n; i++) taken[i].store(1);
        try {
            for (int i = 0; i < n; i++) th.emplace_back(&CopyPool::work, this, i);
        } catch (...) {      // (no more threads to be had: the pool works with the helpers it got, copy() reads th.size())
        }
    }
    // dst <- src, n bytes, shared among the caller and the helpers (below 256 KiB, or in a forked child whose helpers did not
    // survive the fork, the caller copies alone).  Calls are serialised by copy_mu: contexts of several threads share the pool.
    // The caller never just waits: after its own part it takes over whatever a helper has not claimed yet (a helper that is
    // descheduled, or gone, cannot stall the call), and yields the core while claimed parts finish.
    std::mutex copy_mu;
    void copy(void* dst, const void* src, size_t n) {
        const int nh = (int)th.size();
        if (nh == 0 || n < ((size_t)256 << 10) || getpid() != owner) { memcpy(dst, src, n); return; }
        std::lock_guard<std::mutex> one(copy_mu);
        const size_t parts = (size_t)nh + 1, per = ((n / parts) + 63) & ~(size_t)63;
        {
            std::lock_guard<std::mutex> lk(mu);
            // Order matters: a helper still finishing its loop iteration of the PREVIOUS call may claim a job of this one the moment
            // its `taken` flag reads 0 -- so the count it will decrement is set first, and the flag is released only after the job
            // it guards has been written (the helper's claim acquires it).
            pending.store(nh, std::memory_order_relaxed);
            for (int i = 0; i < nh; i++) {
                const size_t off = per * (size_t)(i + 1);
                const size_t len = off >= n ? 0 : (i == nh - 1 ? n - off : (off + per > n ? n - off : per));
                jobs[i] = Job{(char*)dst + off, (const char*)src + off, len};
                taken[i].store(0, std::memory_order_release);
            }
            gen.fetch_add(1, std::memory_order_release);
        }
        cv.notify_all();
        memcpy(dst, src, per < n ? per : n);
        for (int i = nh - 1; i >= 0; i--) {      // work stealing, from the far end (the helpers start from their own slots)
            int expect = 0;
            if (taken[i].compare_exchange_strong(expect, 1, std::memory_order_acq_rel)) run(i);
        }
        for (unsigned long spins = 0; pending.load(std::memory_order_acquire) != 0; spins++) {
            if (spins > 4000) sched_yield();
#if defined(__x86_64__)
            else __builtin_ia32_pause();
#endif
        }
    }
};
}  // namespace
int xfer_stage(gv_ctx* c) {
    if (!c->xfer_pin) HIPCHK(c, hipHostMalloc(&c->xfer_pin, XFER_BYTES));
    return 0;
}
int to_host(gv_ctx* c, void* dst, const void* src_dev, size_t nbytes) {
    if (xfer_stage(c)) return 1;
    // pieces of 2 MiB: the host's copy of piece k into the caller's buffer runs while pieces k + 1 ... cross PCIe (one event per
    // piece; a whole staging buffer of device-to-host copy followed by a whole buffer of memcpy cost 1.8 ms per 8 MB, of which
    // 0.3 ms were the link)
    constexpr size_t PIECE = (size_t)2 << 20;
    constexpr int NP = (int)(XFER_BYTES / PIECE);
    for (int k = 0; k < NP; k++)
        if (!c->xfer_ev[k]) HIPCHK(c, hipEventCreateWithFlags(&c->xfer_ev[k], hipEventDisableTiming));
    for (size_t off = 0; off < nbytes; off += XFER_BYTES) {
        const size_t n = nbytes - off < XFER_BYTES ? nbytes - off : XFER_BYTES;
        int np = 0;
        for (size_t q = 0; q < n; q += PIECE, np++) {
            const size_t len = n - q < PIECE ? n - q : PIECE;
            HIPCHK(c, hipMemcpyAsync((char*)c->xfer_pin + q, (const char*)src_dev + off + q, len, hipMemcpyDeviceToHost, c->stream));
            HIPCHK(c, hipEventRecord(c->xfer_ev[np], c->stream));
        }
        np = 0;
        for (size_t q = 0; q < n; q += PIECE, np++) {
            const size_t len = n - q < PIECE ? n - q : PIECE;
            HIPCHK(c, hipEventSynchronize(c->xfer_ev[np]));
            CopyPool::get().copy((char*)dst + off + q, (char*)c->xfer_pin + q, len);
        }
    }
    if (nbytes == 0) HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}
// sync = false: returns once the caller's buffer has been read (its bytes are in the staging buffer or on their way); the
// copy to the device is ordered on the context's stream like any kernel.  The next to_host / to_device synchronises the stream
// before it touches the staging buffer again.
int to_device(gv_ctx* c, void* dst_dev, const void* src, size_t nbytes, bool sync) {
    if (xfer_stage(c)) return 1;
    for (size_t off = 0; off < nbytes; off += XFER_BYTES) {
        const size_t n = nbytes - off < XFER_BYTES ? nbytes - off : XFER_BYTES;
        HIPCHK(c, hipStreamSynchronize(c->stream));       // whatever used the staging buffer last has left it
        CopyPool::get().copy(c->xfer_pin, (const char*)src + off, n);
        HIPCHK(c, hipMemcpyAsync((char*)dst_dev + off, c->xfer_pin, n, hipMemcpyHostToDevice, c->stream));
    }
    if (sync) HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}


// timing == 2: resolve the pending event pairs into the kernel counters
void ev_resolve(gv_ctx* c) {
    for (size_t i = 0; i < c->ev_used; i++) {
        gv_ctx::EvRec& r = c->ev_pool[i];
        float ms = 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            if (r.kind == 0) { c->cnt.ms_ax_kernel += ms; c->cnt.n_ax_kernel++; }
            else if (r.kind == 1) { c->cnt.ms_atx_kernel += ms; c->cnt.n_atx_kernel++; }
            else { c->cnt.ms_allreduce += ms; c->cnt.n_allreduce++; }
        }
    }
    c->ev_used = 0;
}
gv_ctx::EvRec* ev_next(gv_ctx* c, int kind) {
    if (c->timing != 2) return nullptr;
    if (c->ev_used == 4096) ev_resolve(c);
    if (c->ev_used == c->ev_pool.size()) {
        gv_ctx::EvRec r{nullptr, nullptr, kind};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return nullptr;
        c->ev_pool.push_back(r);
    }
    gv_ctx::EvRec* r = &c->ev_pool[c->ev_used++];
    r->kind = kind;
    return r;
}

// ---- persisted picks ---------------------------------------------------------------------------------------------
// The decomposition picked for a (device ISA, CU count, N, M, layout) is appended to a small text file under $GV_TUNE_CACHE_DIR, else
// $XDG_CACHE_HOME/gvamp_amd, else ~/.cache/gvamp_amd (GV_TUNE_CACHE=0: neither read nor written), so that only the first
// run on a shape pays for the measurement.  One line per key, the last matching line wins; a line is written with one
// O_APPEND write (ranks of a sharded job may share the file).  Results never depend on the pick (exact integer
// accumulation), so a stale or foreign line can cost time, never correctness; every loaded pick is range-checked.
constexpr int GV_TUNE_VERSION = 8;   // bump when the candidate set or the line format changes shape
#ifndef GV_KERNEL_SRC_HASH
#error "build with -DGV_KERNEL_SRC_HASH=\"...\" (gvamp_amd/build.py computes it from the streaming-kernel sources)"
#endif
static std::string tune_cache_file() {
    const char* on = getenv("GV_TUNE_CACHE");
    if (on && atoi(on) == 0) return std::string();
    std::string dir;
    if (const char* d = getenv("GV_TUNE_CACHE_DIR")) dir = d;
    else if (const char* x = getenv("XDG_CACHE_HOME")) dir = std::string(x) + "/gvamp_amd";
    else if (const char* h = getenv("HOME")) dir = std::string(h) + "/.cache/gvamp_amd";
    else return std::string();
    return dir + "/decomp.txt";
}
static std::string tune_key(gv_ctx* c) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, c->device) != hipSuccess) { (void)hipGetLastError(); return std::string(); }
    // the device is identified by ISA target and CU count (the marketing name is not stable: it reads empty under rocprofv3)
    char buf[256];
    // GV_KERNEL_SRC_HASH (gvamp_amd/build.py: sha256 of gv_mfma.hip + gv_mfma.h) ties a line to the kernels it was measured on: a
    // pick made for other kernel sources is never read back
    snprintf(buf, sizeof(buf), "v%d|%s|%s|%d|%lld|%lld|L%d|", GV_TUNE_VERSION, GV_KERNEL_SRC_HASH, pr.gcnArchName,
             pr.multiProcessorCount, (long long)c->N, (long long)c->M, c->plan.layout);
    return buf;
}
// is decomposition d admissible for side (0: ATx / stripes_m, 1: Ax / stripes_n) of this context?
static bool decomp_ok(const gv_ctx* c, const gvm::Decomp& d, int side) {
    const gvm::Plan& pl = c->plan;
    const int64_t nkb = side ? pl.nkb_n : pl.nkb_m, nrg = side ? pl.nrg_n : pl.nrg_m;
    const int64_t min_ks = side ? (c->M * 512 + 2147483646LL) / 2147483647LL : 1;
    int64_t pieces;
    if (d.skL > 0) {
        if (d.skL < 8 || (side && min_ks > 1) || nkb < 2 || d.piv < 0 || d.piv > (nrg + 3) / 4) return false;
        pieces = (nkb + d.skL - 1) / d.skL + 1;
    } else {
        if (d.ks < 1 || d.ks > 64 || d.ks > nkb || d.ks < min_ks || d.piv != 0) return false;
        pieces = d.ks;
    }
    if (!(d.taper >= 0.f && d.taper < 1.f) || !(d.geo >= 0.f && d.geo < 1.f) || (d.geo > 0.f && d.skL > 0) || (d.prio != 0 && d.prio != 1)) return false;
    if (d.occ != 0 && d.occ != 2 && d.occ != 3) return false;
    if (!(d.xskew >= -0.2f && d.xskew <= 0.2f) || (d.xskew != 0.f && d.skL > 0)) return false;
    return (size_t)pieces * 4 * nrg * (side ? pl.rows_n : 64) * 8 * 4 <= pl.partial_bytes;
}
static bool tune_cache_load(gv_ctx* c) {
    const std::string path = tune_cache_file(), key = tune_key(c);
    if (path.empty() || key.empty()) return false;
    FILE* f = fopen(path.c_str(), "r");
    if (!f) return false;
    char line[1024];
    gvm::Decomp got[4];
    bool have = false;
    while (fgets(line, sizeof(line), f)) {
        if (strncmp(line, key.c_str(), key.size()) != 0) continue;
        gvm::Decomp d[4];
        long long sk[4], pv[4];
        {
            const char* q = line + key.size();
            int nread = 0, ok = 1;
            for (int k = 0; k < 4 && ok; k++) {
                if (sscanf(q, "%d %lld %lld %d %f %f %d %f%n", &d[k].ks, &sk[k], &pv[k], &d[k].prio, &d[k].taper, &d[k].geo, &d[k].occ, &d[k].xskew,
                           &nread) != 8)
                    ok = 0;
                q += nread;
            }
            if (!ok) continue;
        }
        for (int k = 0; k < 4; k++) { d[k].skL = sk[k]; d[k].piv = pv[k]; got[k] = d[k]; }
        have = true;
    }
    fclose(f);
    if (!have) return false;
    for (int k = 0; k < 4; k++)
        if (!decomp_ok(c, got[k], k >> 1)) return false;
    if (!c->ks_fixed_m) { c->plan.dm[0] = got[0]; c->plan.dm[1] = got[1]; }
    if (!c->ks_fixed_n) { c->plan.dn[0] = got[2]; c->plan.dn[1] = got[3]; }
    return true;
}
// picks shipped in-tree for this very build of the kernels (gv_tune_builtin.h); GV_TUNE_BUILTIN=0 ignores them
static bool tune_builtin_load(gv_ctx* c) {
    const char* on = getenv("GV_TUNE_BUILTIN");
    if ((on && atoi(on) == 0) || strcmp(GV_BUILTIN_FOR_HASH, GV_KERNEL_SRC_HASH) != 0) return false;
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, c->device) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (strncmp(pr.gcnArchName, "gfx950", 6) != 0 || pr.multiProcessorCount != 256) return false;
    for (const BuiltinPick& b : GV_BUILTIN_PICKS) {
        if (b.N != c->N || b.M != c->M || b.layout != c->plan.layout || b.N == 0) continue;
        for (int k = 0; k < 4; k++)
            if (!decomp_ok(c, b.d[k], k >> 1)) return false;
        if (!c->ks_fixed_m) { c->plan.dm[0] = b.d[0]; c->plan.dm[1] = b.d[1]; }
        if (!c->ks_fixed_n) { c->plan.dn[0] = b.d[2]; c->plan.dn[1] = b.d[3]; }
        return true;
    }
    return false;
}
static void tune_cache_store(gv_ctx* c) {
    const std::string path = tune_cache_file(), key = tune_key(c);
    if (path.empty() || key.empty() || c->ks_fixed_m || c->ks_fixed_n) return;   // overrides are not picks
    const size_t slash = path.rfind('/');
    std::string dir = path.substr(0, slash);
    for (size_t i = 1; i <= dir.size(); i++)                                      // mkdir -p
        if (i == dir.size() || dir[i] == '/') (void)mkdir(dir.substr(0, i).c_str(), 0755);
    const gvm::Decomp* d[4] = {&c->plan.dm[0], &c->plan.dm[1], &c->plan.dn[0], &c->plan.dn[1]};
    char buf[1024];
    int n = snprintf(buf, sizeof(buf), "%s", key.c_str());
    for (int k = 0; k < 4; k++)
        n += snprintf(buf + n, sizeof(buf) - n, "%d %lld %lld %d %.2f %.2f %d %.3f ", d[k]->ks, (long long)d[k]->skL, (long long)d[k]->piv, d[k]->prio,
                      d[k]->taper, d[k]->geo, d[k]->occ, d[k]->xskew);
    n += snprintf(buf + n, sizeof(buf) - n, "\n");
    const int fd = open(path.c_str(), O_WRONLY | O_APPEND | O_CREAT, 0644);
    if (fd < 0) return;
    ssize_t w = write(fd, buf, (size_t)n);
    (void)w;
    close(fd);
}

// The work decomposition of each streaming-kernel class (ATx, two-vector ATx, Ax, two-vector Ax) is picked by measurement
// among the candidates gv_set_dims lists, once per shard, before its first matvec in kernel mode 1, on the resident stripes
// with throw-away vectors (no counters, no collectives) -- unless an earlier run on the same (device, N, M) left its picks in
// the cache above.  Protocol, sized so that the cold cost stays a fraction of a second at 100 GB:
//   stage A  the uniform splits short-listed by the cost model and the balanced grids, without / with their natural priority
//            setting;  stage B  on the winner only: progress-based wave priority (uniform splits), then tapered segment
//            lengths 0.5 / 0.9 (uniform splits with more than one segment), then longer segments for one set of four XCDs (both
//            signs), then two workgroups per CU instead of three.  At most ~20 timed candidates per class.
//   long kernels (>= 4 ms): ONE run of the product being tuned per candidate -- at that length neither the clocks nor what
//            ran before move the result; short kernels: one untimed pair, then two batches of products of the side being tuned,
//            each timed on its own inside the alternating Ax -> ATx sequence the solvers issue (the other side on its current
//            pick, untimed) -- a decomposition that won by 2 % back to back with itself was measured 10 % behind inside the
//            alternating sequence (N = 50k x M = 200k, two-vector Ax); the faster batch counts (one launch in 20-30 of some
//            decompositions lands 15-35 % above the rest).  Operands are pseudo-random: a constant vector populates one digit
//            plane and ranks the candidates differently.
// Results do not depend on the decomposition (exact integer accumulation), so tuning never changes a bit of output.
int autotune_ks(gv_ctx* c) {
    c->ks_tuned = true;
    c->tune_seconds = 0.0;
    c->tune_source = 0;
    if ((c->ks_fixed_m && c->ks_fixed_n) || !c->have_stripes || c->M <= 0 || !c->have_stats) { c->tune_source = 3; return 0; }
    if (tune_cache_load(c)) { c->tune_source = 2; return 0; }
    if (tune_builtin_load(c)) { c->tune_source = 4; return 0; }
    const auto wall0 = std::chrono::steady_clock::now();
    gvm::Plan& pl = c->plan;
    double *xm = nullptr, *wm = nullptr, *wm2 = nullptr, *pn = nullptr, *zn = nullptr, *zn2 = nullptr;
    auto done = [&](int rc) {
        for (double* q : {xm, wm, wm2, pn, zn, zn2}) if (q) (void)hipFree(q);
        pl.ev0 = pl.ev1 = nullptr;
        c->tune_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
        return rc;
    };
    if (hipMalloc(&xm, sizeof(double) * c->M) != hipSuccess || hipMalloc(&wm, sizeof(double) * c->M) != hipSuccess ||
        hipMalloc(&wm2, sizeof(double) * c->M) != hipSuccess || hipMalloc(&pn, sizeof(double) * c->npad) != hipSuccess ||
        hipMalloc(&zn, sizeof(double) * c->npad) != hipSuccess || hipMalloc(&zn2, sizeof(double) * c->npad) != hipSuccess) {
        (void)hipGetLastError();
        return done(0);                      // no room for the scratch vectors: keep the model's pick
    }
    gvk::fill_hash(c->stream, xm, c->M, 1);           // representative operands: every digit plane populated
    gvk::fill_hash(c->stream, pn, c->npad, 2);
    pl.ev0 = pl.ev1 = nullptr;
    constexpr int TUNE_MAXREPS = 12;
    hipEvent_t tev[2 * TUNE_MAXREPS] = {};
    for (hipEvent_t& e : tev)
        if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); e = nullptr; }
    struct EvGuard { hipEvent_t* e; int n; ~EvGuard() { for (int i = 0; i < n; i++) if (e[i]) (void)hipEventDestroy(e[i]); } } ev_guard{tev, 2 * TUNE_MAXREPS};
    // side: 0 = both products (the pair the solvers issue), 1 = Ax side only, 2 = ATx side only
    auto run = [&](int dual, int side) {
        if (side != 2) {
            if (dual) gvm::ax2(c->stream, pl, xm, xm, c->mave, c->msig, c->mask2, c->npad, 1.0, c->red_partial, zn, zn2);
            else gvm::ax(c->stream, pl, xm, c->mave, c->msig, c->mask2, c->npad, 1.0, c->red_partial, zn);
        }
        if (side != 1) {
            if (dual) gvm::atx2(c->stream, pl, pn, pn, c->npad, c->mave, c->msig, 1.0, c->red_partial, wm, wm2, nullptr, nullptr, 1.0, 0.0);
            else gvm::atx(c->stream, pl, pn, c->npad, c->mave, c->msig, 1.0, c->red_partial, wm);
        }
    };
    auto timed = [&](int dual, int side, int reps) -> double {   // ms per repetition
        (void)hipEventRecord(c->ev0, c->stream);
        for (int r = 0; r < reps; r++) run(dual, side);
        (void)hipEventRecord(c->ev1, c->stream);
        if (hipEventSynchronize(c->ev1) != hipSuccess) return -1.0;
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        return ms / reps;
    };
    // short kernels: the product being tuned alone, timed launch by launch INSIDE the alternating sequence the solvers issue (the
    // other side runs, untimed, before every timed product): ms per product, prepare / quantise / finalise included.  Timing the
    // pair diluted a 5 % difference on one side to 2.5 % of a number that moves by 1-2 % from batch to batch.
    auto timed_side = [&](int dual, bool ax_side, int reps) -> double {
        if (reps > TUNE_MAXREPS) reps = TUNE_MAXREPS;
        for (int r = 0; r < reps; r++) {
            if (!tev[2 * r] || !tev[2 * r + 1]) return -1.0;
            run(dual, ax_side ? 2 : 1);
            (void)hipEventRecord(tev[2 * r], c->stream);
            run(dual, ax_side ? 1 : 2);
            (void)hipEventRecord(tev[2 * r + 1], c->stream);
        }
        if (hipEventSynchronize(tev[2 * reps - 1]) != hipSuccess) return -1.0;
        double tot = 0.0;
        for (int r = 0; r < reps; r++) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, tev[2 * r], tev[2 * r + 1]);
            tot += ms;
        }
        return tot / reps;
    };
    const bool verbose = getenv("GV_AUTOTUNE_VERBOSE") != nullptr;
    if (timed(0, 0, 1) < 0) { KCHK(c); return done(1); }        // clocks and caches up before anything is compared
    for (int step = 0; step < 4; step++) {
        const int dual = step >> 1;
        const bool is_ax = (step & 1) == 0;             // Ax side first, then the ATx side against the tuned Ax
        const int cls = is_ax ? 2 + dual : dual;        // 0 ATx, 1 two-vector ATx, 2 Ax, 3 two-vector Ax
        if (is_ax ? c->ks_fixed_n : c->ks_fixed_m) continue;
        const std::vector<gvm::Decomp>& cand = is_ax ? c->dec_cand_n : c->dec_cand_m;
        gvm::Decomp& d = is_ax ? pl.dn[dual] : pl.dm[dual];
        if (cand.empty()) continue;
        d = cand[0];
        const double t_pair = timed(dual, 0, 1);        // sizes the protocol of this class (and warms it)
        if (t_pair < 0) { KCHK(c); return done(1); }
        const bool solo = t_pair >= 8.0;                // both products >= ~4 ms
        const int side = solo ? (is_ax ? 1 : 2) : 0;
        int reps = solo ? 1 : (int)(8.0 / (t_pair > 1e-3 ? t_pair : 1e-3)) + 1;     // (timed products per batch; the other side runs beside each)
        if (reps > 8) reps = 8;
        auto measure = [&](const gvm::Decomp& cd) -> double {
            d = cd;
            if (!solo && timed(dual, 0, 1) < 0) return -1.0;   // untimed: the first launch of a new grid shape
            // short kernels: two batches, the faster one counts.  One launch in 20-30 of some decompositions lands 15-35 % above
            // the rest (profiles/r6_launch_dist_shard.txt); a single batch of 2-8 pairs that catches one ranks its candidate by
            // the accident -- round 5's table held a pick 2.5 % behind its own priority variant that way.
            double t = solo ? timed(dual, side, reps) : timed_side(dual, is_ax, reps);
            if (!solo && t >= 0) {
                const double t2 = timed_side(dual, is_ax, reps);
                if (t2 < 0) return -1.0;
                if (t2 < t) t = t2;
            }
            if (verbose)
                fprintf(stderr, "[gvamp autotune] class %d ks %d skL %lld whole quads %lld prio %d taper %.1f geo %.2f occ %d xskew %.3f : %.4f ms / %s\n", cls, cd.ks,
                        (long long)cd.skL, (long long)cd.piv, cd.prio, cd.taper, cd.geo, cd.occ, cd.xskew, t, "product");
            return t;
        };
        gvm::Decomp best = cand[0];
        double best_t = -1;
        auto consider = [&](const gvm::Decomp& cd) -> int {
            const double t = measure(cd);
            if (t < 0) return 1;
            if (best_t < 0 || t < best_t * 0.997) { best_t = t; best = cd; }   // the list order breaks near-ties
            return 0;
        };
        // stage A: uniform splits without priority, balanced grids (which carry their priority setting)
        bool any_plain = false;
        for (const gvm::Decomp& cd : cand) any_plain |= cd.skL <= 0 && cd.prio == 0;
        for (const gvm::Decomp& cd : cand) {
            if (cd.skL <= 0 && cd.prio != 0 && cd.geo == 0.f && any_plain) continue;
            if (consider(cd)) { d = cand[0]; KCHK(c); return done(1); }
        }
        // stage B on the winner: priority, then taper
        if (best.skL <= 0) {
            bool prio_listed = false;
            for (const gvm::Decomp& cd : cand) prio_listed |= cd.skL <= 0 && cd.prio == 1 && cd.ks == best.ks;
            if (best.prio == 0 && prio_listed) {
                gvm::Decomp t = best; t.prio = 1;
                if (consider(t)) { d = cand[0]; KCHK(c); return done(1); }
            }
            if (best.ks > 1 && best.taper == 0.f && best.geo == 0.f) {      // (a geometric split has its own segment lengths)
                const gvm::Decomp base = best;
                for (float tp : {0.5f, 0.9f}) {
                    gvm::Decomp t = base; t.taper = tp;
                    if (consider(t)) { d = cand[0]; KCHK(c); return done(1); }
                }
            }
        }
        // ... then more work for four of the eight XCDs (Decomp::xskew), on a winner whose quads have at least two segments.  WHICH four
        // finish equal shares first belongs to the box and to where the allocation landed (profiles/r6_xcd_skew.txt): both signs are
        // measured on the resident data, and the better one is pushed once more if it beat the equal shares
        if (best.skL <= 0 && best.ks >= 2) {
            const gvm::Decomp base = best;
            const double t_base = best_t;
            for (float sk : {0.02f, -0.02f}) {
                gvm::Decomp t = base; t.xskew = sk;
                if (consider(t)) { d = cand[0]; KCHK(c); return done(1); }
            }
            if (best.xskew != 0.f && best_t < t_base) {
                gvm::Decomp t = best; t.xskew = best.xskew > 0.f ? 0.035f : -0.035f;
                if (consider(t)) { d = cand[0]; KCHK(c); return done(1); }
            }
        }
        // ... then two workgroups per CU instead of three, on the winner and on the best geometric split (which is what gains from it
        // where anything does: many short workgroups late in the launch)
        {
            const gvm::Decomp base = best;
            gvm::Decomp t = base; t.occ = 2;
            if (consider(t)) { d = cand[0]; KCHK(c); return done(1); }
            for (const gvm::Decomp& cd : cand)
                if (cd.geo > 0.f && cd.ks >= 6 && !(cd.ks == base.ks && cd.geo == base.geo)) {
                    gvm::Decomp g = cd; g.occ = 2;
                    if (consider(g)) { d = cand[0]; KCHK(c); return done(1); }
                    break;
                }
        }
        d = best;
        if (verbose)
            fprintf(stderr, "[gvamp autotune] class %d -> ks %d skL %lld whole quads %lld prio %d taper %.1f geo %.2f occ %d xskew %.3f\n", cls, d.ks,
                    (long long)d.skL, (long long)d.piv, d.prio, d.taper, d.geo, d.occ, d.xskew);
    }
    KCHK(c);
    c->tune_source = 1;
    tune_cache_store(c);
    return done(0);
}

// ---- the exchange step of data::Ax (data.cpp:928/:995) overlapped with the decode (north_star; GV_OVERLAP=T or gv_set_overlap)
// The product is cut into T chunks of individuals (whole quads of row groups).  Chunk t is decoded on the context's stream;
// its slice of the N-vector is all-reduced and scaled on a side stream while chunk t + 1 decodes; the context's stream joins
// the side stream at the end.  Every chunk is the same exact integer arithmetic as the undivided pass and the all-reduce
// of a slice adds the same numbers in the same rank order: results are bit-identical to the one-message form.  What it
// buys is (T - 1)/T of the exchange time at the price of T - 1 more kernel tails (docs/history/rounds1-3.md section 6): a knob to measure
// on an 8-GPU node, off by default.
int ax_overlapped(gv_ctx* c, int nv, const double* xa, const double* xb, double* outa, double* outb,
                  const gvm::CgHook* cg) {
    const double scale = 1.0 / sqrt((double)c->N);
    gvm::Plan& pl = c->plan;
    if (!c->comm_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_chunk, hipEventDisableTiming));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_comm, hipEventDisableTiming));
    }
    // The slices are cut from N alone, in units of 1024 individuals -- a whole number of row-group quads in either resident
    // layout (4 x 4 x 64 rows on two stripe sets, 4 x 256 rows on the tile layout): the ranks of one job may hold different
    // layouts (gv_set_layout(.., 3) decides per rank from its free HBM) and must still exchange the same ranges.
    const int64_t nu = (c->N + 1023) / 1024;
    int T = c->overlap_tiles;
    if (T > nu) T = (int)nu;
    const int64_t gpu_ = 1024 / pl.rows_n;      // row groups per unit: 16 or 4
    const bool empty = c->M == 0;      // an empty shard sends zeros through the same sequence of slice messages
    if (empty) {
        gvk::fill(c->stream, outa, c->npad, 0.0);
        if (nv == 2) gvk::fill(c->stream, outb, c->npad, 0.0);
    } else
        gvm::ax_prep(c->stream, pl, xa, nv == 2 ? xb : nullptr, c->mave, c->msig, c->red_partial, cg);
    for (int t = 0; t < T; t++) {
        const int64_t u0 = nu * t / T, u1 = nu * (t + 1) / T;
        int64_t rg0 = u0 * gpu_, rg1 = t == T - 1 ? pl.nrg_n : u1 * gpu_;
        if (rg0 > pl.nrg_n) rg0 = pl.nrg_n;
        if (rg1 > pl.nrg_n) rg1 = pl.nrg_n;
        if (!empty && rg1 > rg0) gvm::ax_rows(c->stream, pl, nv, rg0, rg1, c->mask2, c->npad, 1.0, outa, nv == 2 ? outb : nullptr, cg);
        KCHK(c);
        const int64_t n0 = u0 * 1024;
        const int64_t cnt = (t == T - 1 ? c->npad : u1 * 1024) - n0;    // the last slice takes the pad tail (zeros) along
        if (cnt <= 0) continue;
        HIPCHK(c, hipEventRecord(c->ev_chunk, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->comm_stream, c->ev_chunk, 0));
        if (comm_allreduce_on(c, outa + n0, (size_t)cnt, c->comm_stream)) return 1;
        gvk::scale_vec(c->comm_stream, outa + n0, cnt, scale);
        if (nv == 2) {
            if (comm_allreduce_on(c, outb + n0, (size_t)cnt, c->comm_stream)) return 1;
            gvk::scale_vec(c->comm_stream, outb + n0, cnt, scale);
        }
        KCHK(c);
    }
    HIPCHK(c, hipEventRecord(c->ev_comm, c->comm_stream));
    if (!(c->force_multi & 4))     // (bit 4 of gv_debug_force_multi: fault injection for tests/test_gpu_forced_multi.py -- the join is dropped)
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_comm, 0));
    return 0;
}
bool use_overlap(const gv_ctx* c) {   // nothing rank-local in here (have_stripes: every rank holds SOME re-encoded layout, whichever)
    return c->overlap_tiles > 1 && is_multi(c) && c->kernel_mode == 1 && c->have_stripes;
}

// data::Ax on device pointers.  x: M doubles, out: npad doubles.
int ax_device(gv_ctx* c, const double* x, double* out, const gvm::CgHook* cg) {
    NEED(c, c->have_stats && c->mask2, "Ax: bed, mask and marker statistics must be set first");
    const double scale = 1.0 / sqrt((double)c->N);
    const bool multi = is_multi(c);
    if (c->kernel_mode == 1 && use_overlap(c)) {
        if (c->M > 0 && !c->ks_tuned && autotune_ks(c)) return 1;
        c->plan.ev0 = c->plan.ev1 = nullptr;
        if (ax_overlapped(c, 1, x, nullptr, out, nullptr, cg)) return 1;
        c->cnt.n_ax++;
        c->cnt.n_ax_pass++;
        return 0;
    } else if (c->M == 0) {   // an empty shard (Mt < ranks) contributes zeros through the same collective as its peers
        gvk::fill(c->stream, out, c->npad, 0.0);
        KCHK(c);
    } else if (c->kernel_mode != 0) {
        NEED(c, c->have_stripes, "Ax: kernel modes 1 and 2 need a re-encoded layout (gv_set_layout before ingest)");
        if (!c->ks_tuned && autotune_ks(c)) return 1;
        Timer t(c, &c->cnt.ms_ax);
        gv_ctx::EvRec* er = ev_next(c, 0);
        c->plan.ev0 = er ? er->a : nullptr;
        c->plan.ev1 = er ? er->b : nullptr;
        if (c->kernel_mode == 2)      // two-level fixed point: head and residual of x in the two slots of one pass (no CG hooks in this mode)
            gvm::ax_wide(c->stream, c->plan, x, c->mave, c->msig, c->mask2, c->npad, multi ? 1.0 : scale, c->red_partial, out);
        else
            gvm::ax(c->stream, c->plan, x, c->mave, c->msig, c->mask2, c->npad, multi ? 1.0 : scale, c->red_partial, out, cg);
        KCHK(c);
        t.stop();
    } else {
        NEED(c, c->have_raw, "Ax: kernel mode 0 needs the raw row layout (not the default: gv_set_layout(ctx, 1, ..) before ingest)");
        Timer t(c, &c->cnt.ms_ax);
        gvk::ax_table(c->stream, x, c->mave, c->msig, c->M, c->t3);
        gv_ctx::EvRec* er = ev_next(c, 0);
        if (er) (void)hipEventRecord(er->a, c->stream);
        gvk::ax_f64(c->stream, c->bed, c->M, c->pitch, c->t3, c->ax_chunks, c->ax_partial, c->npad);
        if (er) (void)hipEventRecord(er->b, c->stream);
        gvk::ax_reduce(c->stream, c->ax_partial, c->ax_chunks, c->npad, c->mask2, multi ? 1.0 : scale, out);
        KCHK(c);
        t.stop();
    }
    c->cnt.n_ax++;
    c->cnt.n_ax_pass++;
    if (multi) {   // data.cpp:995 MPI_Allreduce, then the 1/sqrt(N) of :998-1005
        Timer t(c, &c->cnt.ms_allreduce);
        gv_ctx::EvRec* er = ev_next(c, 2);                 // timing == 2: the exchange step of the pass, un-synchronised
        if (er) (void)hipEventRecord(er->a, c->stream);
        if (comm_allreduce(c, out, c->npad)) return 1;
        gvk::scale_vec(c->stream, out, c->npad, scale);
        if (er) (void)hipEventRecord(er->b, c->stream);
        KCHK(c);
        t.stop();
        if (c->timing == 1) c->cnt.n_allreduce++;
    }
    return 0;
}

// data::ATx on device pointers.  p: npad doubles (zero at NA / pad slots), out: M doubles.
int atx_device(gv_ctx* c, const double* p, double* out, const double* addx, double tau, double gam2, const gvm::CgHook* cg) {
    NEED(c, c->have_stats, "ATx: bed and marker statistics must be set first");
    if (c->M == 0) {   // empty shard: no markers; the <d,p> a CG hook asks for is 0 from this rank (it is all-reduced next)
        if (cg && cg->dot_out[0]) { gvk::fill(c->stream, cg->dot_out[0], 8, 0.0); KCHK(c); }
        // (the N-space search direction a pass would have advanced on its way in -- CgHook::pn -- is replicated on every rank)
        if (cg && cg->pn[0]) { gvk::p_update_st(c->stream, cg->pn[0], cg->zn[0], cg->state[0], c->npad); KCHK(c); }
        c->cnt.n_atx++;
        c->cnt.n_atx_pass++;
        return 0;
    }
    if (c->kernel_mode != 0 && c->have_stripes && !c->ks_tuned && autotune_ks(c)) return 1;
    Timer t(c, &c->cnt.ms_atx);
    if (c->kernel_mode != 0 && c->M > 0) {
        NEED(c, c->have_stripes, "ATx: kernel modes 1 and 2 need a re-encoded layout (gv_set_layout before ingest)");
        gv_ctx::EvRec* er = ev_next(c, 1);
        c->plan.ev0 = er ? er->a : nullptr;
        c->plan.ev1 = er ? er->b : nullptr;
        if (c->kernel_mode == 2)
            gvm::atx_wide(c->stream, c->plan, p, c->npad, c->mave, c->msig, 1.0 / sqrt((double)c->N), c->red_partial, out, addx, tau, gam2);
        else
            gvm::atx(c->stream, c->plan, p, c->npad, c->mave, c->msig, 1.0 / sqrt((double)c->N), c->red_partial, out, addx, tau, gam2, cg);
    } else {
        NEED(c, c->have_raw, "ATx: kernel mode 0 needs the raw row layout (not the default: gv_set_layout(ctx, 1, ..) before ingest)");
        gv_ctx::EvRec* er = ev_next(c, 1);
        if (er) (void)hipEventRecord(er->a, c->stream);
        gvk::atx_f64(c->stream, c->bed, c->M, c->pitch, p, c->mave, c->msig, 1.0 / sqrt((double)c->N), out);
        if (er) (void)hipEventRecord(er->b, c->stream);
        if (addx) gvk::axpby(c->stream, out, tau, out, gam2, addx, c->M);
    }
    KCHK(c);
    t.stop();
    c->cnt.n_atx++;
    c->cnt.n_atx_pass++;
    return 0;
}

// two-vector forms: ONE pass over the shard in kernel mode 1, two single passes otherwise
int ax2_device(gv_ctx* c, const double* xa, const double* xb, double* outa, double* outb, const gvm::CgHook* cg) {
    // The collective sequence below must not depend on rank-local state (an empty shard, M == 0, enters the same calls
    // with zeros): it is chosen by the kernel mode -- the same on every rank of a job -- and the output pointers only.
    if (c->kernel_mode != 1) {
        if (ax_device(c, xa, outa)) return 1;
        return ax_device(c, xb, outb);
    }
    NEED(c, c->have_stats && c->mask2, "Ax: bed, mask and marker statistics must be set first");
    const double scale = 1.0 / sqrt((double)c->N);
    const bool multi = is_multi(c);
    if (use_overlap(c)) {
        if (c->M > 0 && !c->ks_tuned && autotune_ks(c)) return 1;
        c->plan.ev0 = c->plan.ev1 = nullptr;
        if (ax_overlapped(c, 2, xa, xb, outa, outb, cg)) return 1;
        c->cnt.n_ax += 2;
        c->cnt.n_ax_pass += 1;
        return 0;
    } else if (c->M == 0) {
        gvk::fill(c->stream, outa, c->npad, 0.0);
        gvk::fill(c->stream, outb, c->npad, 0.0);
        KCHK(c);
    } else {
        NEED(c, c->have_stripes, "Ax: kernel mode 1 needs the stripe layouts (gv_set_layout before ingest)");
        if (!c->ks_tuned && autotune_ks(c)) return 1;
        Timer t(c, &c->cnt.ms_ax);
        gv_ctx::EvRec* er = ev_next(c, 0);
        c->plan.ev0 = er ? er->a : nullptr;
        c->plan.ev1 = er ? er->b : nullptr;
        gvm::ax2(c->stream, c->plan, xa, xb, c->mave, c->msig, c->mask2, c->npad, multi ? 1.0 : scale, c->red_partial, outa, outb, cg);
        KCHK(c);
        t.stop();
    }
    c->cnt.n_ax += 2;
    c->cnt.n_ax_pass += 1;
    if (multi) {
        Timer t(c, &c->cnt.ms_allreduce);
        gv_ctx::EvRec* er = ev_next(c, 2);
        if (er) (void)hipEventRecord(er->a, c->stream);
        if (c->w_n && c->w_n2 && outa == c->w_n->d && outb == c->w_n2->d) {   // w_n | w_n2 (one allocation): one message, one scaling pass
            if (comm_allreduce(c, outa, 2 * c->npad)) return 1;
            gvk::scale_vec(c->stream, outa, 2 * c->npad, scale);
        } else {
            if (comm_allreduce(c, outa, c->npad)) return 1;
            if (comm_allreduce(c, outb, c->npad)) return 1;
            gvk::scale_vec(c->stream, outa, c->npad, scale);
            gvk::scale_vec(c->stream, outb, c->npad, scale);
        }
        if (er) (void)hipEventRecord(er->b, c->stream);
        KCHK(c);
        t.stop();
        if (c->timing == 1) c->cnt.n_allreduce++;
    }
    return 0;
}
int atx2_device(gv_ctx* c, const double* pa, const double* pb, double* outa, double* outb, const double* addxa,
                const double* addxb, double tau, double gam2, const gvm::CgHook* cg) {
    if (c->M == 0) {   // empty shard: no local markers, no collective in ATx
        for (int k = 0; k < 2 && cg; k++) {
            if (cg->dot_out[k]) gvk::fill(c->stream, cg->dot_out[k], 8, 0.0);
            if (cg->pn[k]) gvk::p_update_st(c->stream, cg->pn[k], cg->zn[k], cg->state[k], c->npad);
        }
        KCHK(c);
        c->cnt.n_atx += 2;
        c->cnt.n_atx_pass += 1;
        return 0;
    }
    if (!(c->kernel_mode == 1 && c->have_stripes)) {
        if (atx_device(c, pa, outa, addxa, tau, gam2)) return 1;
        return atx_device(c, pb, outb, addxb, tau, gam2);
    }
    NEED(c, c->have_stats, "ATx: bed and marker statistics must be set first");
    if (!c->ks_tuned && autotune_ks(c)) return 1;
    Timer t(c, &c->cnt.ms_atx);
    gv_ctx::EvRec* er = ev_next(c, 1);
    c->plan.ev0 = er ? er->a : nullptr;
    c->plan.ev1 = er ? er->b : nullptr;
    gvm::atx2(c->stream, c->plan, pa, pb, c->npad, c->mave, c->msig, 1.0 / sqrt((double)c->N), c->red_partial, outa, outb, addxa,
              addxb, tau, gam2, cg);
    KCHK(c);
    t.stop();
    c->cnt.n_atx += 2;
    c->cnt.n_atx_pass += 1;
    return 0;
}

int lmmse_device(gv_ctx* c, const double* v, double tau, double gam2, double* out) {
    if (ensure_work(c)) return 1;
    if (ax_device(c, v, c->w_n->d)) return 1;
    return atx_device(c, c->w_n->d, out, v, tau, gam2);   // res = tau * A^T A v + gam2 v (vamp.cpp:1112-1115), in the ATx epilogue
}

void free_dataset(gv_ctx* c) {
    auto F = [](auto*& p) {
        if (p) (void)hipFree(p);
        p = nullptr;
    };
    F(c->bed); F(c->mask2); F(c->mave); F(c->msig); F(c->t3); F(c->ax_partial); F(c->counts);
    if (c->stripes_slab) {
        (void)hipFree(c->stripes_slab);
        c->stripes_slab = nullptr;
        c->plan.stripes_m = c->plan.stripes_n = nullptr;
    }
    F(c->plan.stripes_m); F(c->plan.stripes_n); F(c->plan.tiles); F(c->plan.dig0); F(c->plan.dig1); F(c->plan.cv); F(c->plan.ev);
    F(c->plan.cv2); F(c->plan.ev2);
    F(c->plan.scal); F(c->plan.partial);
    F(c->cgx_state); F(c->cgx_go);
    if (c->cgx_rel_h) (void)hipHostFree(c->cgx_rel_h);
    c->cgx_rel = c->cgx_rel_h = nullptr;
    F(c->aat_slab);
    c->aat_slab_cap = 0;
    c->cgx_relcap = 0;
    c->spec_hint_steps[0] = c->spec_hint_steps[1] = c->spec_hint_steps[2] = 0;
    c->spec_hint_passes = 0;
    c->plan = gvm::Plan();
    c->have_raw = c->have_stripes = false;
    for (gv_vec** v : {&c->w_n, &c->cg_r, &c->cg_z, &c->cg_p, &c->cg_d, &c->mave_p, &c->msig_p, &c->numb_p, &c->w_n2,
                      &c->cg2_r, &c->cg2_z, &c->cg2_p, &c->cg2_d})
        if (*v) {
            vec_del(c, *v);
            *v = nullptr;
        }
    c->have_stats = false;
}

}  // namespace gvi

using namespace gvi;

// Geometry of the streaming kernels for the layout that will be built (c->want_tile) and the candidate work decompositions
// of each side.  Called by gv_set_dims and again by ingest when gv_set_layout changed the layout in between.
static int plan_decomps(gv_ctx* c) {
    const int64_t N = c->N, M = c->M;
    gvm::Plan& pl = c->plan;
    pl.M = M; pl.N = N;
    pl.nrg_m = (M + 63) / 64;  pl.nkb_m = (N + 255) / 256;
    pl.nrg_n = (N + 63) / 64;  pl.nkb_n = (M + 255) / 256;
    pl.layout = c->want_tile ? 1 : 0;
    pl.rows_n = 64;
    if (c->want_tile) {   // one layout: the Ax side walks the marker-group-major super-blocks transposed
        pl.nrg_n = pl.nkb_m;       // row groups of 256 individuals
        pl.nkb_n = pl.nrg_m;       // K-steps of 64 markers
        pl.rows_n = 256;
    }
    // K-splits.  A launch is W = ceil(nrg / 4) * ks workgroups, each walking nkb / ks K-blocks; 768 are resident at a time
    // (256 CUs x 3).  Sweeps on MI355X (GV_KS_M / GV_KS_N overrides; N = 50k ... 400k, M = 125k ... 1M) show 2-7 % between
    // neighbouring splits, from three effects no closed form ranks reliably: a short last round of workgroups running at
    // their own ceiling (a workgroup keeps 32 KiB in flight, ~450 of them saturate HBM), stragglers of the last round
    // against an emptying chip, and per-workgroup prologue / epilogue / partial sums (~40 K-blocks' worth).  So a small
    // cost model only SHORT-LISTS three candidates here
    //   per = nkb/ks + 40 ;  W <= 768: T = per * max(W, 448) / 768 ;  W > 768: T = per * W / 768 + straggle * per
    // and the pick among them is MEASURED once per shard on the resident data (autotune_ks, before the first matvec).
    // Results do not depend on the split (exact integer accumulation), so tuning never changes a bit of output.
    auto rank_ks = [](int64_t nrg, int64_t nkb, int64_t min_ks, double straggle, int* out3) {
        const int64_t nq = (nrg + 3) / 4;
        out3[0] = out3[1] = out3[2] = (int)(min_ks > 1 ? min_ks : 1);
        if (nq <= 0 || nkb <= 0) return;
        int64_t hi = nkb / 32 > 1 ? nkb / 32 : 1;                  // never fewer than 32 K-blocks per workgroup
        if (hi > 64) hi = 64;
        if (min_ks > hi) hi = min_ks;
        double cost[3] = {0, 0, 0};
        int n = 0;
        for (int64_t ks = min_ks > 1 ? min_ks : 1; ks <= hi && ks <= nkb; ks++) {
            const double per = (double)nkb / (double)ks + 40.0;
            const int64_t W = nq * ks;
            const double t = W <= 768 ? per * (double)(W > 448 ? W : 448) / 768.0 : per * (double)W / 768.0 + straggle * per;
            int pos = n < 3 ? n : 3;                                // insertion into the three cheapest
            while (pos > 0 && t < cost[pos - 1]) pos--;
            if (pos >= 3) continue;
            for (int j = (n < 3 ? n : 2); j > pos; j--) { cost[j] = cost[j - 1]; out3[j] = out3[j - 1]; }
            cost[pos] = t;
            out3[pos] = (int)ks;
            if (n < 3) n++;
        }
        for (int j = n; j < 3; j++) out3[j] = out3[n > 0 ? n - 1 : 0];
    };
    const int64_t min_ks_n = (M * 512 + 2147483646LL) / 2147483647LL;
    int ks3_m[3], ks3_n[3];
    rank_ks(pl.nrg_m, pl.nkb_m, 1, 0.4, ks3_m);
    rank_ks(pl.nrg_n, pl.nkb_n, min_ks_n, 0.8, ks3_n);
    // balanced decomposition (k_mfma_matvec<., true>): cells per workgroup for a grid of G workgroups.  A segment is at most
    // min(skL, nkb) K-blocks long; on the Ax side it must respect the int32 bound that min_ks_n expresses.
    auto skL_of = [](int64_t nrg, int64_t nkb, int64_t G) -> int64_t {
        const int64_t U = ((nrg + 3) / 4) * nkb;
        if (U <= 0 || G <= 0) return 0;
        const int64_t L = (U + G - 1) / G;
        return L < 8 ? 8 : L;
    };
    // Candidate list per side, default first: the uniform splits in the model's order without priority, the same with
    // priority, then balanced grids of one and two workgroups per slot (always with priority: without it the staggered
    // workgroups of a balanced launch lose ~10 % to the arbiter's oldest-first tail).
    int prio_only = -1;                                    // GV_PRIO=0/1 (development): restrict to one setting
    if (const char* e = getenv("GV_PRIO")) prio_only = atoi(e) ? 1 : 0;
    // piv quads whole (0: as many whole rounds of 768 as the quads allow), the rest balanced over G workgroups; the pieces a row
    // of the remainder is cut into are bounded so that the int32 partial sums stay below 1 GB (4 planes x 32 B per row and piece)
    auto hybrid_of = [](int64_t nrg, int64_t nkb, int64_t rows, int64_t piv, int64_t G) -> gvm::Decomp {
        gvm::Decomp h;
        const int64_t nq = (nrg + 3) / 4;
        if (piv <= 0) piv = nq / 768 * 768;
        if (piv <= 0 || piv >= nq || nkb < 2 || G <= 0) return h;
        const int64_t cells = (nq - piv) * nkb;
        int64_t maxp = (int64_t)(1.0e9 / (128.0 * (double)nrg * (double)rows));
        if (maxp > 60) maxp = 60;
        if (maxp < 3) return h;
        int64_t L = (cells + G - 1) / G;
        const int64_t Lmin = (nkb + maxp - 2) / (maxp - 1);
        if (L < Lmin) L = Lmin;
        if (L < 8) L = 8;
        h.ks = 1; h.skL = L; h.piv = piv; h.prio = 1;
        return h;
    };
    auto build = [&](const int* ks3, int64_t nrg, int64_t nkb, int64_t rows, bool balanced_ok, int64_t min_ks_u, bool geo_side, std::vector<gvm::Decomp>& out) {
        out.clear();
        for (int prio = 0; prio < 2; prio++) {
            if (prio_only >= 0 && prio != prio_only) continue;
            for (int j = 0; j < 3; j++) {
                if (j > 0 && (ks3[j] == ks3[0] || (j == 2 && ks3[2] == ks3[1]))) continue;
                gvm::Decomp d; d.ks = ks3[j]; d.skL = 0; d.prio = prio;
                out.push_back(d);
            }
        }
        // geometric splits (big first): ks segments per quad, segment j = geo^j of segment 0, every one at least 8 K-blocks long.
        // Many short segments (6-8): Ax side only -- measured in-process against the tuner's picks
        // (profiles/r4_decomp_ab_inprocess.txt) they gain 1-2.4 % on the Ax classes of 12.5 GB and 2.5 GB shards and lose 1-10 % on
        // every ATx class (GV_TUNE_GEO=1 lists them there too).  Two to four segments: both sides.  A launch of nq < 768 whole-K
        // workgroups (one round that does not fill the chip) streams with nq of the 768 slots for its
        // whole length and ends on the spread of their speeds; a short second segment fills the idle slots for the first part of
        // the launch instead: two-vector ATx of the 8-GPU shard (N = 400k x M = 125k, 489 quads, tile layout), per-launch
        // distributions of 80 launches each in one process (profiles/r6_launch_dist_shard.txt): ks 1 p50 2.016 ms with 3 launches
        // of 80 at 2.2-2.7 ms, ks 2 geo 0.5 p50 1.904 ms, max 1.941.  Just above a round (config 5's two-vector ATx, 782 quads): ks 4
        // geo 0.5 0.424 ms with no launch above 0.431 against the hybrid's 0.441 with 4 of 30 at 0.49-0.56.
        if (prio_only != 0) {
            std::vector<std::pair<int, float>> gks;
            for (const auto& gk : {std::pair<int, float>{2, 0.5f}, {2, 0.35f}, {3, 0.5f}, {4, 0.5f}}) gks.push_back(gk);
            if (geo_side || getenv("GV_TUNE_GEO")) for (const auto& gk : {std::pair<int, float>{6, 0.6f}, {8, 0.65f}, {8, 0.8f}}) gks.push_back(gk);
            for (const auto& gk : gks) {
                double tot = 0.0, wlast = 1.0;
                for (int j = 0; j < gk.first; j++) { tot += wlast; if (j + 1 < gk.first) wlast *= gk.second; }
                if ((double)nkb * wlast / tot < 8.0 || gk.first < min_ks_u) continue;
                bool dup = false;
                for (const gvm::Decomp& o : out) dup |= o.skL <= 0 && o.ks == gk.first && o.geo == gk.second;
                if (dup) continue;
                gvm::Decomp d; d.ks = gk.first; d.skL = 0; d.prio = 1; d.geo = gk.second;
                out.push_back(d);
            }
        }
        if (balanced_ok && prio_only != 0 && nkb >= 2) {
            // hybrid: whole rounds of the 768 resident workgroups go one quad per workgroup (in step over K), the quads that are
            // left over are cut into 768 balanced ranges -- for quad counts just above a multiple of 768 (gv_mfma.hip).  Listed
            // before the fully balanced grids: on a tie it is the one that fetches every digit block once per XCD
            gvm::Decomp h = hybrid_of(nrg, nkb, rows, 0, 768);
            if (h.skL > 0) out.push_back(h);
            for (int r = 1; r <= 2; r++) {
                gvm::Decomp d; d.ks = 1; d.skL = skL_of(nrg, nkb, 768 * r); d.prio = 1;
                if (d.skL > 0) out.push_back(d);
            }
        }
    };
    build(ks3_m, pl.nrg_m, pl.nkb_m, 64, true, 1, false, c->dec_cand_m);
    build(ks3_n, pl.nrg_n, pl.nkb_n, pl.rows_n, min_ks_n <= 1, min_ks_n, true, c->dec_cand_n);
    c->ks_tuned = c->ks_fixed_m = c->ks_fixed_n = false;
    // overrides (development): GV_KS_M / GV_KS_N fix a uniform K-split of the ATx / Ax kernels, GV_SK_M / GV_SK_N a balanced
    // grid of that many workgroups (both with the priority setting of GV_PRIO, default off / on), GV_AUTOTUNE=0 keeps the
    // first candidate
    auto fix = [&](std::vector<gvm::Decomp>& cand, bool& fixed, gvm::Decomp d) { cand.assign(1, d); fixed = true; };
    const float taper_env = getenv("GV_TAPER") ? (float)atof(getenv("GV_TAPER")) : 0.f;
    const float geo_env = getenv("GV_GEO") ? (float)atof(getenv("GV_GEO")) : 0.f;
    if (const char* e = getenv("GV_KS_M")) {
        int v = atoi(e);
        if (v >= 1 && v <= pl.nkb_m && v <= 64) { gvm::Decomp d; d.ks = v; d.prio = prio_only == 1; d.taper = taper_env; d.geo = geo_env; fix(c->dec_cand_m, c->ks_fixed_m, d); }
    }
    if (const char* e = getenv("GV_KS_N")) {
        int v = atoi(e);
        if (v >= min_ks_n && v >= 1 && v <= pl.nkb_n && v <= 64) { gvm::Decomp d; d.ks = v; d.prio = prio_only == 1; d.taper = taper_env; d.geo = geo_env; fix(c->dec_cand_n, c->ks_fixed_n, d); }
    }
    if (const char* e = getenv("GV_SK_M")) {
        gvm::Decomp d; d.skL = skL_of(pl.nrg_m, pl.nkb_m, atoi(e)); d.prio = prio_only != 0;
        if (d.skL > 0) fix(c->dec_cand_m, c->ks_fixed_m, d);
    }
    if (const char* e = getenv("GV_SK_N")) {
        gvm::Decomp d; d.skL = min_ks_n > 1 ? 0 : skL_of(pl.nrg_n, pl.nkb_n, atoi(e)); d.prio = prio_only != 0;
        if (d.skL > 0) fix(c->dec_cand_n, c->ks_fixed_n, d);
    }
    // GV_HY_M / GV_HY_N = "<whole quads>:<workgroups of the remainder>": a hybrid decomposition (0 whole quads: whole rounds of 768)
    auto hy = [&](const char* e, int64_t nrg, int64_t nkb, int64_t rows) {
        long long piv = 0, G = 768;
        sscanf(e, "%lld:%lld", &piv, &G);
        gvm::Decomp d = hybrid_of(nrg, nkb, rows, piv, G);
        d.prio = prio_only != 0;
        return d;
    };
    if (const char* e = getenv("GV_HY_M")) {
        gvm::Decomp d = hy(e, pl.nrg_m, pl.nkb_m, 64);
        if (d.skL > 0) fix(c->dec_cand_m, c->ks_fixed_m, d);
    }
    if (const char* e = getenv("GV_HY_N")) {
        gvm::Decomp d = hy(e, pl.nrg_n, pl.nkb_n, pl.rows_n);
        if (d.skL > 0 && min_ks_n <= 1) fix(c->dec_cand_n, c->ks_fixed_n, d);
    }
    if (const char* e = getenv("GV_AUTOTUNE"))
        if (atoi(e) == 0) c->ks_fixed_m = c->ks_fixed_n = true;
    if (c->dec_cand_m.empty()) c->dec_cand_m.assign(1, gvm::Decomp());
    if (c->dec_cand_n.empty()) { gvm::Decomp d; d.ks = (int)(min_ks_n > 1 ? min_ks_n : 1); c->dec_cand_n.assign(1, d); }
    pl.dm[0] = pl.dm[1] = c->dec_cand_m[0];
    pl.dn[0] = pl.dn[1] = c->dec_cand_n[0];
    return 0;
}



extern "C" {

int gv_abi_version(void) { return GV_ABI_VERSION; }

// Contexts of one process (the in-process rank groups of the tests: one thread per rank) are created and torn down under one
// lock: stream / event creation and destruction racing across threads is where a runtime is least exercised, and neither
// call is on any hot path.
static std::mutex g_lifecycle_mu;
static void gv_destroy_locked(gv_ctx* c);      // the caller holds g_lifecycle_mu and has drained the streams

int gv_create(int device, gv_ctx** out) {
    if (!out) return fail(nullptr, "gv_create: out is NULL");
    *out = nullptr;
    std::lock_guard<std::mutex> life(g_lifecycle_mu);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(nullptr, "gv_create: no HIP device available (%s); libgvamp has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(nullptr, "gv_create: device %d out of range (0..%d)", device, ndev - 1);
    gv_ctx* c = new gv_ctx();
    c->device = device;
    auto bail = [&](const char* what, hipError_t err) {
        fail(nullptr, "gv_create: %s failed: %s", what, hipGetErrorString(err));
        delete c;
        return 1;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreate(&c->stream)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipMalloc(&c->red_partial, sizeof(double) * RED_BLOCKS * RED_MAXK)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc(&c->red_out, sizeof(double) * RED_MAXK)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipHostMalloc(&c->host_pin, sizeof(double) * RED_MAXK)) != hipSuccess) return bail("hipHostMalloc", e);
    // scalar mailbox (read_scalars); a runtime that cannot map coherent host memory keeps the copy + synchronise read-back and the
    // host-driven CG loop
    {
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, sizeof(double) * RED_MAXK + 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
            hipHostGetDevicePointer(&dp, hp, 0) == hipSuccess) {
            memset(hp, 0, sizeof(double) * RED_MAXK + 64);
            c->mbox = static_cast<double*>(hp);
            c->mbox_dev = static_cast<double*>(dp);
            c->use_mbox = true;
            if (hipMalloc(&c->pub_counter, sizeof(unsigned int)) != hipSuccess ||
                hipMemset(c->pub_counter, 0, sizeof(unsigned int)) != hipSuccess) {
                (void)hipGetLastError();
                c->pub_counter = nullptr;          // read_scalars then keeps its own publish launch
            }
        } else {
            (void)hipGetLastError();
            if (hp) (void)hipHostFree(hp);
        }
    }
    if (const char* ov = getenv("GV_OVERLAP")) c->overlap_tiles = atoi(ov) > 64 ? 64 : (atoi(ov) < 0 ? 0 : atoi(ov));
    *out = c;
    // GVAMP_FORCE_MULTI=<transport>[:<delay_us>] -- gv_debug_force_multi for every context of the process (drivers, bench.py)
    // The fault-injection bit (4) is reachable through the explicit call only, and a forced context says so once per process:
    // a job that inherits the variable must not run the loop-back transport silently.
    if (const char* fm = getenv("GVAMP_FORCE_MULTI")) {
        const int tr = atoi(fm);
        const char* colon = strchr(fm, ':');
        if (tr < 0 || tr > 3) {
            g_create_err = "gv_create: GVAMP_FORCE_MULTI=" + std::string(fm) + ": transport must be 0..3 (the fault-injection bit is gv_debug_force_multi only)";
            *out = nullptr;
            gv_destroy_locked(c);
            return 1;
        }
        if (tr > 0) {
            if (gv_debug_force_multi(c, tr, colon ? atoi(colon + 1) : 0)) {
                g_create_err = "gv_create: GVAMP_FORCE_MULTI: " + c->err;
                *out = nullptr;
                gv_destroy_locked(c);
                return 1;
            }
            static std::atomic<bool> said{false};
            if (!said.exchange(true))
                fprintf(stderr, "[gvamp] GVAMP_FORCE_MULTI=%s: one-rank contexts take the multi-rank branches over a loop-back exchange "
                                "(test hook; results are unchanged, every pass pays the exchange)\n", fm);
        }
    }
    return 0;
}

void gv_destroy(gv_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
    std::lock_guard<std::mutex> life(g_lifecycle_mu);
    gv_destroy_locked(c);
}
static void gv_destroy_locked(gv_ctx* c) {
    c->comm = nullptr;
    c->comm_keep.reset();          // ncclCommDestroy if this was the last context sharing the communicator
    free_dataset(c);
    while (!c->live_vecs.empty()) vec_del(c, *c->live_vecs.begin());   // vectors the caller never gave back
    if (c->red_partial) (void)hipFree(c->red_partial);
    if (c->red_out) (void)hipFree(c->red_out);
    if (c->host_pin) (void)hipHostFree(c->host_pin);
    if (c->mbox) (void)hipHostFree(c->mbox);
    if (c->pub_counter) (void)hipFree(c->pub_counter);
    for (double* q : c->loop_buf) if (q) (void)hipFree(q);
    if (c->xfer_pin) (void)hipHostFree(c->xfer_pin);
    for (hipEvent_t e : c->xfer_ev) if (e) (void)hipEventDestroy(e);
    for (auto& r : c->ev_pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_chunk) (void)hipEventDestroy(c->ev_chunk);
    if (c->ev_comm) (void)hipEventDestroy(c->ev_comm);
    if (c->comm_stream) { (void)hipStreamSynchronize(c->comm_stream); (void)hipStreamDestroy(c->comm_stream); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* gv_last_error(const gv_ctx* c) { return c ? c->err.c_str() : g_create_err.c_str(); }

int gv_synchronize(gv_ctx* c) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the overlapped exchange's side stream normally joins c->stream through ev_comm; a caller that synchronises wants the
    // context quiescent whatever edges were (or, under fault injection, were not) recorded
    if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));
    return 0;
}

int gv_set_dims(gv_ctx* c, int64_t N, int64_t M, int64_t Mt, int64_t S) {
    NEED(c, N > 0 && M >= 0 && Mt >= M && S >= 0 && S + M <= Mt, "gv_set_dims: need N > 0, 0 <= M, S + M <= Mt");
    HIPCHK(c, hipSetDevice(c->device));
    free_dataset(c);
    c->N = N; c->M = M; c->Mt = Mt; c->S = S;
    c->mbytes = (N + 3) / 4;
    c->pitch = align_up(c->mbytes, 64);
    c->npad = 4 * c->pitch;
    c->nonas = N;
    const int64_t Mal = M > 0 ? M : 1;
    HIPCHK(c, hipMalloc(&c->mave, sizeof(double) * Mal));
    HIPCHK(c, hipMalloc(&c->msig, sizeof(double) * Mal));
    HIPCHK(c, hipMalloc(&c->t3, sizeof(double) * 3 * Mal));
    HIPCHK(c, hipMalloc(&c->counts, sizeof(uint32_t) * 3 * Mal));
    int64_t col_tiles = (c->pitch / 4 + 255) / 256;
    int64_t chunks = (2048 + col_tiles - 1) / col_tiles;
    if (chunks > 256) chunks = 256;
    if (chunks > M) chunks = M > 0 ? M : 1;
    if (chunks < 1) chunks = 1;
    c->ax_chunks = (int)chunks;
    HIPCHK(c, hipMalloc(&c->ax_partial, sizeof(double) * c->ax_chunks * c->npad));
    // i8 MFMA family: int32 digit sums must not overflow (|r'| <= 3, |digit| <= 128; Ax adds the miss plane)
    NEED(c, N * 384 < 2147483647LL, "gv_set_dims: N too large for the int32 accumulators of kernel mode 1");
    // the streaming kernel counts (quad, K-block) cells in 32 bits: M N / 65536 of them (35 TB of genotypes at the limit)
    NEED(c, ((M + 255) / 256 + 1) * ((N + 255) / 256 + 1) < 2147483647LL, "gv_set_dims: shard too large for the 32-bit cell index");
    if (plan_decomps(c)) return 1;
    return gv_set_mask(c, nullptr, N);
}

int64_t gv_mbytes(const gv_ctx* c) { return c->mbytes; }

// Ingest: fills the resident layouts chunk by chunk (markers [m0, m0+mc), m0 % 256 == 0) so that the raw rows never
// have to be resident as a whole when only the stripes are wanted (N=400k x M=1M: 100 GB raw + 2 x 100 GB stripes).
// nbytes of the file at `off` into the pinned staging buffer, by GV_IO_THREADS (default 8) concurrent pread streams: one
// thread copying out of the page cache moves ~9 GB/s, a fraction of what the PCIe link takes
// returns 0 ok, -1 end of file before nbytes were read, else the errno of the failing pread (EINTR is retried)
static int read_slab(int fd, int64_t off, uint8_t* dst, size_t nbytes) {
    // 8 concurrent pread streams (measured at config-2 size out of the page cache: 4 -> 27, 8 -> 33-34, 12 -> 32-37 GB/s including
    // the allocation of the layout), never more than the CPUs this process may use
    int nt = 8;
    {
        cpu_set_t cs;
        if (sched_getaffinity(0, sizeof(cs), &cs) == 0 && CPU_COUNT(&cs) < nt) nt = CPU_COUNT(&cs) < 1 ? 1 : CPU_COUNT(&cs);
    }
    if (const char* e = getenv("GV_IO_THREADS")) nt = atoi(e) < 1 ? 1 : (atoi(e) > 32 ? 32 : atoi(e));
    if (nbytes < ((size_t)8 << 20)) nt = 1;
    std::vector<int> st(nt, 0);
    auto work = [&](int t) {
        const size_t lo = nbytes * (size_t)t / (size_t)nt, hi = nbytes * (size_t)(t + 1) / (size_t)nt;
        size_t done = lo;
        while (done < hi) {
            const ssize_t r = pread(fd, dst + done, hi - done, (off_t)(off + (int64_t)done));
            if (r < 0 && errno == EINTR) continue;
            if (r < 0) { st[t] = errno ? errno : EIO; return; }
            if (r == 0) { st[t] = -1; return; }
            done += (size_t)r;
        }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)nt);
    int started = 1;                            // ranges [1, started) have a thread; the rest are read by this one
    try {
        for (; started < nt; started++) th.emplace_back(work, started);
    } catch (...) {
    }
    work(0);
    for (int t = started; t < nt; t++) work(t);
    for (std::thread& x : th) x.join();
    for (int v : st) if (v > 0) return v;      // a real I/O error wins over a short file
    for (int v : st) if (v) return v;
    return 0;
}

static int ingest(gv_ctx* c, const uint8_t* host_bed, bool synth, uint64_t seed, uint32_t miss_thr, FILE* file = nullptr,
                  uint32_t ld_block = 0, uint32_t ld_thr = 0, int64_t file_off = 0) {
    NEED(c, c->N > 0, "ingest: gv_set_dims must be called first");
    NEED(c, c->want_raw || c->want_stripes, "ingest: gv_set_layout disabled both layouts");
    HIPCHK(c, hipSetDevice(c->device));
    const int64_t M = c->M, P = c->pitch;
    gvm::Plan& pl = c->plan;
    c->have_raw = c->have_stripes = c->have_stats = false;
    if (c->want_raw && !c->bed) HIPCHK(c, hipMalloc(&c->bed, (size_t)(M > 0 ? M : 1) * P));
    if (!c->want_raw && c->bed) { (void)hipFree(c->bed); c->bed = nullptr; }
    const auto t_in0 = std::chrono::steady_clock::now();
    if (c->want_auto && c->want_stripes && (pl.tiles || pl.stripes_m)) {
        // a re-ingest on a context whose layout auto already picked keeps that pick: tearing a tile layout down to try two
        // stripe sets again would fail at exactly the sizes auto exists for
        c->want_tile = pl.layout == 1;
    } else if (c->want_auto && c->want_stripes) {
        // gv_set_layout(.., 3), the default: ONE tile layout -- half the bytes to allocate and fill -- unless the caller has announced a
        // long run (gv_set_expected_passes >= 1000) AND two stripe sets (the faster ATx, by 2-5 %) fit the free HBM with room for the
        // vectors and scratch.  The second set costs its bytes once more at ingest -- allocated at 25-400 GB/s depending on whether
        // the driver is still wiping freed memory, filled at ~400 GB/s: 0.5-4.3 s per 100 GB measured -- and returns ~3 % of one ATx
        // pass (bytes / 6.5 TB/s) per pass: break-even between 500 and 9 000 passes whatever the shard size (both sides scale with the
        // bytes).  A run that says nothing about its length (a bare binding, bench.py's five iterations: 5.3 of 6.4 s to solution were
        // that allocation in round 4) is not assumed to be long.
        size_t free_b = 0, total_b = 0;
        HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
        const double one = (double)((M + 63) / 64) * (double)((c->N + 255) / 256) * 4096.0;
        const double other = (c->want_raw ? (double)M * (double)P : 32768.0 * (double)P) + 64.0 * (double)(M + c->npad) + 2.0e9;
        const bool two_fit = 2.0 * one + other <= 0.92 * (double)free_b;
        c->want_tile = !(two_fit && c->expected_passes >= 1000);
    }
    const int want_layout = c->want_tile ? 1 : 0;
    const bool rebuild = c->want_stripes && (pl.layout != want_layout || !(want_layout ? pl.tiles : pl.stripes_m));
    if (rebuild) {
        // (re)build the geometry and the buffers of the MFMA family for the layout asked for
        if (c->stripes_slab) {       // the two stripe sets are views into one allocation
            (void)hipFree(c->stripes_slab);
            c->stripes_slab = nullptr;
            pl.stripes_m = pl.stripes_n = nullptr;
        }
        for (void** q : {&pl.stripes_m, &pl.stripes_n, &pl.tiles, &pl.dig0, &pl.dig1, (void**)&pl.cv, (void**)&pl.ev,
                         (void**)&pl.cv2, (void**)&pl.ev2, (void**)&pl.scal, (void**)&pl.partial})
            if (*q) { (void)hipFree(*q); *q = nullptr; }
        if (plan_decomps(c)) return 1;
    }
    // The allocation of the resident layout -- seconds when the driver is still wiping what an earlier process freed -- runs on a
    // helper thread while this one gets the source ready: pinned staging buffers, the chunk buffer, and for a file source the
    // first two chunks read from the file system.  Wall = max(allocate, prepare) instead of their sum (gv_ingest_info2: overlap_s).
    std::string alloc_err;
    double alloc_secs = 0.0;
    auto alloc_layout = [&]() {
        const auto ta = std::chrono::steady_clock::now();
        auto A = [&](hipError_t e, const char* what) {
            if (e != hipSuccess && alloc_err.empty()) alloc_err = std::string(what) + ": " + hipGetErrorString(e);
            return e == hipSuccess;
        };
        if (!A(hipSetDevice(c->device), "hipSetDevice")) return;
        if (rebuild) {
            const int64_t nkbmax = pl.nkb_m > pl.nkb_n ? pl.nkb_m : pl.nkb_n;
            if (want_layout) {
                if (!A(hipMalloc(&pl.tiles, (size_t)(pl.nrg_m > 0 ? pl.nrg_m : 1) * pl.nkb_m * 4096), "hipMalloc(tile layout)")) return;
            } else {
                // ONE allocation for the two stripe sets, stripes_n (the Ax side) first.  Where the driver places a 100 GB allocation
                // moves the kernel that streams it by 1.5-3.5 % (docs/history/rounds1-3.md section 4.2: nine ingests on one box, Ax 14.9-15.6 ms and
                // ATx 14.8-16.0 ms from one ingest to the next); of two sets carved out of one allocation the first was in its fast
                // mode in nearly every ingest measured (Ax 14.80-14.99 ms in 13 of 14) and the second near it (ATx 14.95-15.4), whichever set
                // came first.  An allocation that large failing falls back to one allocation per set.
                const size_t sz_m = (size_t)(pl.nrg_m > 0 ? pl.nrg_m : 1) * pl.nkb_m * 4096,
                             sz_n = (size_t)pl.nrg_n * (pl.nkb_n > 0 ? pl.nkb_n : 1) * 4096;
                void* slab = nullptr;
                const size_t al = (size_t)1 << 30, off_m = (sz_n + al - 1) / al * al;
                if (hipMalloc(&slab, off_m + sz_m) == hipSuccess) {
                    c->stripes_slab = slab;
                    pl.stripes_n = slab;
                    pl.stripes_m = (char*)slab + off_m;
                } else {
                    (void)hipGetLastError();
                    if (!A(hipMalloc(&pl.stripes_m, sz_m), "hipMalloc(stripes_m)")) return;
                    if (!A(hipMalloc(&pl.stripes_n, sz_n), "hipMalloc(stripes_n)")) return;
                }
            }
            const size_t Mn = (size_t)(M > 0 ? M : 1);
            if (!A(hipMalloc(&pl.dig0, (size_t)(nkbmax > 0 ? nkbmax : 1) * 4096), "hipMalloc(dig0)")) return;
            if (!A(hipMalloc(&pl.dig1, (size_t)(nkbmax > 0 ? nkbmax : 1) * 4096), "hipMalloc(dig1)")) return;
            if (!A(hipMalloc(&pl.cv, sizeof(double) * Mn), "hipMalloc(cv)")) return;
            if (!A(hipMalloc(&pl.ev, sizeof(double) * Mn), "hipMalloc(ev)")) return;
            if (!A(hipMalloc(&pl.cv2, sizeof(double) * Mn), "hipMalloc(cv2)")) return;
            if (!A(hipMalloc(&pl.ev2, sizeof(double) * Mn), "hipMalloc(ev2)")) return;
            if (!A(hipMalloc(&pl.scal, sizeof(double) * 8), "hipMalloc(scal)")) return;
            auto pieces = [](const std::vector<gvm::Decomp>& cand, int64_t nkb) {   // room for every candidate of autotune_ks
                int k = 1;
                for (const gvm::Decomp& d : cand) {
                    const int p = (int)gvm::pieces_max(d, nkb);
                    if (p > k) k = p;
                }
                return k;
            };
            const int km = pieces(c->dec_cand_m, pl.nkb_m), kn = pieces(c->dec_cand_n, pl.nkb_n);
            size_t pa = (size_t)km * 4 * pl.nrg_m * 64 * 8 * 4, pb = (size_t)kn * 4 * pl.nrg_n * pl.rows_n * 8 * 4;
            pl.partial_bytes = pa > pb ? pa : pb;
            if (!A(hipMalloc(&pl.partial, pl.partial_bytes > 0 ? pl.partial_bytes : 4), "hipMalloc(partial sums)")) return;
        }
        alloc_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - ta).count();
    };
    std::thread alloc_thr;
    try {
        alloc_thr = std::thread(alloc_layout);
    } catch (...) {      // (no thread to be had: allocate here, nothing overlaps)
        alloc_layout();
    }
    const auto t_prep0 = std::chrono::steady_clock::now();
    const int64_t CH = file ? 8192 : 32768;   // file source: each pinned staging buffer is CH * mbytes bytes
    uint8_t* tmp = nullptr;
    uint8_t* stage[2] = {nullptr, nullptr};
    hipEvent_t stage_free[2] = {nullptr, nullptr};
    // File source: two pinned staging buffers, so that reading chunk k + 1 from the file system overlaps the PCIe copy and the
    // re-encoding kernels of chunk k (the stream serialises the device side; an event per buffer says when its copy has left)
    int rc = 0;
    for (int b = 0; b < 2 && file && !rc; b++) {
        hipError_t e = hipHostMalloc(&stage[b], (size_t)(M < CH ? (M > 0 ? M : 1) : CH) * c->mbytes);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&stage_free[b], hipEventDisableTiming);
        if (e != hipSuccess) rc = fail(c, "ingest: no pinned staging buffer: %s", hipGetErrorString(e));
    }
    if (!c->want_raw && !rc) {
        const hipError_t e = hipMalloc(&tmp, (size_t)(M < CH ? (M > 0 ? M : 1) : CH) * P);
        if (e != hipSuccess) rc = fail(c, "ingest: no room for the chunk buffer: %s", hipGetErrorString(e));
    }
    // chunks 0 and 1 of a file come off the file system while the layout is still being allocated
    int pre_read = 0, pre_io[2] = {0, 0};
    try {
        for (int b = 0; b < 2 && file && !rc && (int64_t)b * CH < M; b++) {
            const int64_t m0 = (int64_t)b * CH, mc = M - m0 < CH ? M - m0 : CH;
            pre_io[b] = read_slab(fileno(file), file_off + m0 * c->mbytes, stage[b], (size_t)mc * c->mbytes);
            pre_read = b + 1;
        }
    } catch (const std::exception& e) {      // (no reader thread to be had: the allocation thread is still joined below)
        rc = fail(c, "ingest: reading the .bed file failed: %s", e.what());
    }
    const double prep_secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count();
    if (alloc_thr.joinable()) alloc_thr.join();
    if (!alloc_err.empty()) {
        // nothing half-built stays behind: a retry on this context must allocate everything again (rebuild is decided from these
        // pointers), and the streaming kernels must never meet a layout whose digit / partial-sum buffers are missing
        if (c->stripes_slab) { (void)hipFree(c->stripes_slab); c->stripes_slab = nullptr; pl.stripes_m = pl.stripes_n = nullptr; }
        for (void** q : {&pl.stripes_m, &pl.stripes_n, &pl.tiles, &pl.dig0, &pl.dig1, (void**)&pl.cv, (void**)&pl.ev,
                         (void**)&pl.cv2, (void**)&pl.ev2, (void**)&pl.scal, (void**)&pl.partial})
            if (*q) { (void)hipFree(*q); *q = nullptr; }
        (void)hipGetLastError();
        if (!rc) rc = fail(c, "ingest: allocating the resident layout failed: %s", alloc_err.c_str());
    }
    if (!rc && hipDeviceSynchronize() != hipSuccess) rc = fail(c, "ingest: hipDeviceSynchronize failed");
    const auto t_in1 = std::chrono::steady_clock::now();      // the layouts are allocated (the driver maps / wipes 100+ GB)
    c->ingest_overlap_s = alloc_secs < prep_secs ? alloc_secs : prep_secs;
    int64_t chunk = 0;
    for (int64_t m0 = 0; m0 < M && !rc; m0 += CH, chunk++) {
        const int64_t mc = M - m0 < CH ? M - m0 : CH;
        uint8_t* rawp = c->want_raw ? c->bed + m0 * P : tmp;
        hipError_t e = hipSuccess;
        const int sb = (int)(chunk & 1);
        if (synth) {
            gvk::synth_bed(c->stream, rawp, mc, c->S + m0, c->N, P, seed, miss_thr, ld_block, ld_thr);
        } else {
            const uint8_t* src = host_bed ? host_bed + (size_t)m0 * c->mbytes : stage[sb];
            if (file) {
                if (chunk >= 2) e = hipEventSynchronize(stage_free[sb]);      // the copy of chunk - 2 has left this buffer
                const int io = chunk < pre_read ? pre_io[chunk]               // (read while the layout was being allocated)
                               : (e == hipSuccess ? read_slab(fileno(file), file_off + (int64_t)m0 * c->mbytes, stage[sb], (size_t)mc * c->mbytes) : 0);
                if (io) {
                    rc = io < 0 ? fail(c, "ingest: the .bed file ends before marker %lld is complete (short file)", (long long)(c->S + m0 + mc - 1))
                                : fail(c, "ingest: reading the .bed file at marker %lld failed: %s", (long long)(c->S + m0), strerror(io));
                    break;
                }
            }
            if (e == hipSuccess) e = hipMemsetAsync(rawp, 0, (size_t)mc * P, c->stream);
            if (e == hipSuccess)
                e = hipMemcpy2DAsync(rawp, P, src, c->mbytes, c->mbytes, mc, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess && file) e = hipEventRecord(stage_free[sb], c->stream);
        }
        if (e == hipSuccess && c->want_stripes && pl.layout == 1)
            gvm::tile_chunk(c->stream, rawp, P, mc, c->N, pl.tiles, m0 / 64, pl.nkb_m);
        else if (e == hipSuccess && c->want_stripes) {
            gvm::stripes_m_chunk(c->stream, rawp, P, mc, c->N, pl.stripes_m, m0 / 64, pl.nkb_m);
            gvm::stripes_n_chunk(c->stream, rawp, P, mc, c->N, pl.stripes_n, m0 / 256, pl.nkb_n, pl.nrg_n);
        }
        if (e == hipSuccess) e = hipGetLastError();
        // a caller-owned pageable host buffer (gv_upload_bed) and the synthetic source have nothing to overlap: keep the
        // launch queue short; the file source runs ahead by one chunk
        if (e == hipSuccess && !file) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, "ingest chunk at marker %lld failed: %s", (long long)m0, hipGetErrorString(e));
    }
    {
        const hipError_t e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess && !rc) rc = fail(c, "ingest failed: %s", hipGetErrorString(e));
    }
    for (int b = 0; b < 2; b++) {
        if (stage[b]) (void)hipHostFree(stage[b]);
        if (stage_free[b]) (void)hipEventDestroy(stage_free[b]);
    }
    if (tmp) (void)hipFree(tmp);
    c->ingest_alloc_s = std::chrono::duration<double>(t_in1 - t_in0).count();
    c->ingest_bytes = c->want_stripes ? (size_t)(pl.layout == 1 ? 1 : 2) * (size_t)(pl.nrg_m > 0 ? pl.nrg_m : 1) * pl.nkb_m * 4096 : 0;
    if (c->want_raw) c->ingest_bytes += (size_t)(M > 0 ? M : 1) * P;
    c->ingest_fill_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_in1).count();
    if (rc) return rc;
    c->have_raw = c->want_raw;
    c->have_stripes = c->want_stripes;
    return 0;
}

int gv_set_layout(gv_ctx* c, int raw_rows, int stripes) {
    NEED(c, raw_rows || stripes, "gv_set_layout: at least one layout is required");
    NEED(c, stripes >= 0 && stripes <= 3, "gv_set_layout: stripes is 0 (none), 1 (two stripe sets), 2 (one tile layout) or 3 (auto)");
    c->want_raw = raw_rows != 0;
    c->want_stripes = stripes != 0;
    c->want_tile = stripes == 2;
    c->want_auto = stripes == 3;
    return 0;
}

int gv_upload_bed(gv_ctx* c, const uint8_t* bed, size_t nbytes) {
    NEED(c, c->N > 0, "gv_upload_bed: gv_set_dims must be called first");
    NEED(c, nbytes == (size_t)c->M * (size_t)c->mbytes, "gv_upload_bed: nbytes != M * ceil(N/4)");
    return ingest(c, bed, false, 0, 0);
}

int gv_upload_bed_file(gv_ctx* c, const char* path, int64_t offset) {
    NEED(c, c->N > 0, "gv_upload_bed_file: gv_set_dims must be called first");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(c, "gv_upload_bed_file: could not open bed file: %s", path);
    if (fseeko(f, (off_t)offset, SEEK_SET) != 0) {
        fclose(f);
        return fail(c, "gv_upload_bed_file: cannot seek to %lld in %s", (long long)offset, path);
    }
    int rc = ingest(c, nullptr, false, 0, 0, f, 0, 0, offset);
    fclose(f);
    return rc;
}

int gv_synth_bed(gv_ctx* c, uint64_t seed, uint32_t miss_ppm) {
    NEED(c, c->N > 0, "gv_synth_bed: gv_set_dims must be called first");
    uint32_t thr = (uint32_t)(((uint64_t)miss_ppm << 32) / 1000000ull);
    return ingest(c, nullptr, true, seed, thr);
}

int gv_synth_bed_ld(gv_ctx* c, uint64_t seed, uint32_t miss_ppm, uint32_t ld_block, uint32_t ld_ppm) {
    NEED(c, c->N > 0, "gv_synth_bed_ld: gv_set_dims must be called first");
    NEED(c, ld_ppm <= 1000000, "gv_synth_bed_ld: ld_ppm is a probability in 1e-6");
    const uint32_t thr = (uint32_t)(((uint64_t)miss_ppm << 32) / 1000000ull);
    const uint64_t lt = ((uint64_t)ld_ppm << 32) / 1000000ull;
    return ingest(c, nullptr, true, seed, thr, nullptr, ld_block, (uint32_t)(lt > 0xFFFFFFFFull ? 0xFFFFFFFFull : lt));
}

int gv_download_bed(gv_ctx* c, uint8_t* bed, size_t nbytes) {
    NEED(c, c->have_raw, "gv_download_bed: the raw row layout is not resident (not the default: call gv_set_layout(ctx, 1, stripes) before the ingest)");
    NEED(c, nbytes == (size_t)c->M * (size_t)c->mbytes, "gv_download_bed: nbytes != M * ceil(N/4)");
    if (c->M > 0)
        HIPCHK(c, hipMemcpy2DAsync(bed, c->mbytes, c->bed, c->pitch, c->mbytes, c->M, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int gv_set_mask(gv_ctx* c, const uint8_t* mask4, int64_t nonas) {
    NEED(c, c->N > 0, "gv_set_mask: gv_set_dims must be called first");
    const int64_t P4 = c->pitch / 4;
    std::vector<uint32_t> m2(P4, 0u);
    for (int64_t n = 0; n < c->N; n++) {
        bool present = mask4 ? ((mask4[n >> 2] >> (n & 3)) & 1u) : true;
        if (present) m2[n >> 4] |= 3u << (2 * (n & 15));
    }
    if (!c->mask2) HIPCHK(c, hipMalloc(&c->mask2, sizeof(uint32_t) * P4));
    HIPCHK(c, hipMemcpyAsync(c->mask2, m2.data(), sizeof(uint32_t) * P4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->nonas = nonas;
    c->have_stats = false;
    return 0;
}

int gv_marker_stats(gv_ctx* c, double alpha_scale) {
    NEED(c, (c->have_raw || c->have_stripes) && c->mask2, "gv_marker_stats: bed and mask must be set first");
    if (c->have_stripes && c->plan.layout == 1 && (c->kernel_mode != 0 || !c->have_raw))
        gvm::stats_from_tiles(c->stream, c->plan.tiles, c->mask2, c->M, c->plan.nrg_m, c->plan.nkb_m, c->pitch / 4,
                              (double)c->nonas, alpha_scale, c->mave, c->msig, c->counts);
    else if (c->have_stripes && (c->kernel_mode != 0 || !c->have_raw))
        gvm::stats_from_stripes(c->stream, c->plan.stripes_m, c->mask2, c->M, c->plan.nkb_m, c->pitch / 4,
                                (double)c->nonas, alpha_scale, c->mave, c->msig, c->counts);
    else
        gvk::marker_stats(c->stream, c->bed, c->mask2, c->M, c->pitch, (double)c->nonas, alpha_scale, c->mave, c->msig,
                          c->counts);
    c->alpha_scale = alpha_scale;
    KCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_stats = true;
    return 0;
}

int gv_get_marker_stats(gv_ctx* c, double* mave, double* msig) {
    NEED(c, c->have_stats, "gv_get_marker_stats: gv_marker_stats has not run");
    HIPCHK(c, hipMemcpyAsync(mave, c->mave, sizeof(double) * c->M, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(msig, c->msig, sizeof(double) * c->M, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int gv_set_kernel_mode(gv_ctx* c, int mode) {
    NEED(c, mode == 0 || mode == 1 || mode == 2, "gv_set_kernel_mode: mode must be 0 (fp64 VALU), 1 (i8 MFMA fixed point) or 2 (two-level fixed point)");
    c->kernel_mode = mode;
    return 0;
}
int gv_get_kernel_mode(const gv_ctx* c) { return c->kernel_mode; }

// ---- vectors --------------------------------------------------------------------------------------------
int gv_vec_alloc(gv_ctx* c, int space, gv_vec** out) {
    NEED(c, space == GV_SPACE_M || space == GV_SPACE_N, "gv_vec_alloc: bad space");
    HIPCHK(c, hipSetDevice(c->device));
    return vec_new(c, space, out);
}
void gv_vec_free(gv_ctx* c, gv_vec* v) {
    if (!v) return;
    (void)hipStreamSynchronize(c->stream);
    vec_del(c, v);
}
int64_t gv_vec_len(const gv_vec* v) { return v->len; }
int gv_vec_upload(gv_ctx* c, gv_vec* v, const double* src) {
    return to_device(c, v->d, src, sizeof(double) * (v->len > 0 ? v->len : 0));
}
int gv_vec_download(gv_ctx* c, const gv_vec* v, double* dst) {
    return to_host(c, dst, v->d, sizeof(double) * (v->len > 0 ? v->len : 0));
}
int gv_vec_fill(gv_ctx* c, gv_vec* v, double value) {
    gvk::fill(c->stream, v->d, v->len, value);
    KCHK(c);
    return 0;
}
int gv_vec_copy(gv_ctx* c, gv_vec* dst, const gv_vec* src) {
    NEED(c, dst->space == src->space, "gv_vec_copy: space mismatch");
    gvk::copy(c->stream, dst->d, src->d, src->cap);      // (a kernel: enqueued in ~3 us where hipMemcpyAsync takes the host 10-15)
    KCHK(c);
    return 0;
}
int gv_vec_axpby(gv_ctx* c, gv_vec* out, double a, const gv_vec* x, double b, const gv_vec* y) {
    NEED(c, out->space == x->space && (!y || y->space == x->space), "gv_vec_axpby: space mismatch");
    NEED(c, y || b == 0.0, "gv_vec_axpby: y is NULL but b != 0");
    gvk::axpby(c->stream, out->d, a, x->d, b, y ? y->d : nullptr, x->len);
    KCHK(c);
    return 0;
}
int gv_vec_mul(gv_ctx* c, gv_vec* out, const gv_vec* x, const gv_vec* y) {
    NEED(c, out->space == x->space && y->space == x->space, "gv_vec_mul: space mismatch");
    gvk::mul(c->stream, out->d, x->d, y->d, x->len);
    KCHK(c);
    return 0;
}
int gv_vec_dots(gv_ctx* c, int n, const gv_vec* const* x, const gv_vec* const* y, int sync, double* out) {
    NEED(c, n >= 1 && n <= 8, "gv_vec_dots: 1 <= n <= 8");
    const double *xs[8], *ys[8];
    for (int k = 0; k < n; k++) {
        NEED(c, x[k]->space == x[0]->space && y[k]->space == x[0]->space, "gv_vec_dots: space mismatch");
        xs[k] = x[k]->d;
        ys[k] = y[k]->d;
    }
    if (!(sync && is_multi(c))) arm_scalars(c);
    gvk::dots(c->stream, n, xs, ys, x[0]->len, c->red_partial, c->red_out);
    KCHK(c);
    if (sync && comm_allreduce(c, c->red_out, n)) return 1;
    return read_scalars(c, n, out);
}
int gv_vec_dots_ex(gv_ctx* c, int n, const gv_dot_spec* spec, double* out) {
    NEED(c, n >= 1 && n <= 8, "gv_vec_dots_ex: 1 <= n <= 8");
    const double *xa[8], *xb[8], *ya[8], *yb[8];
    int64_t len[8];
    bool any_sync = false;
    for (int k = 0; k < n; k++) {
        const gv_dot_spec& q = spec[k];
        NEED(c, q.xa && q.ya, "gv_vec_dots_ex: xa and ya are required");
        NEED(c, q.ya->space == q.xa->space && (!q.xb || q.xb->space == q.xa->space) && (!q.yb || q.yb->space == q.xa->space),
             "gv_vec_dots_ex: the vectors of one pair live in one space");
        xa[k] = q.xa->d; xb[k] = q.xb ? q.xb->d : nullptr;
        ya[k] = q.ya->d; yb[k] = q.yb ? q.yb->d : nullptr;
        len[k] = q.xa->len;
        any_sync = any_sync || q.sync != 0;
    }
    const bool multi = any_sync && is_multi(c);
    if (!multi) arm_scalars(c);
    gvk::dots_ex(c->stream, n, xa, xb, ya, yb, len, c->red_partial, c->red_out);
    KCHK(c);
    if (multi)      // the scalars to be summed over the ranks, one all-reduce per run of neighbours
        for (int k = 0; k < n;) {
            if (!spec[k].sync) { k++; continue; }
            int e = k;
            while (e < n && spec[e].sync) e++;
            if (comm_allreduce(c, c->red_out + k, (size_t)(e - k))) return 1;
            k = e;
        }
    return read_scalars(c, n, out);
}
int gv_vec_dot(gv_ctx* c, const gv_vec* x, const gv_vec* y, int sync, double* out) {
    return gv_vec_dots(c, 1, &x, &y, sync, out);
}

int gv_ax_dev(gv_ctx* c, const gv_vec* x, gv_vec* out) {
    NEED(c, x->space == GV_SPACE_M && out->space == GV_SPACE_N, "gv_ax_dev: x must be M-space, out N-space");
    return ax_device(c, x->d, out->d);
}
int gv_atx_dev(gv_ctx* c, const gv_vec* p, gv_vec* out) {
    NEED(c, p->space == GV_SPACE_N && out->space == GV_SPACE_M, "gv_atx_dev: p must be N-space, out M-space");
    return atx_device(c, p->d, out->d);
}

int gv_ax2_dev(gv_ctx* c, const gv_vec* xa, const gv_vec* xb, gv_vec* outa, gv_vec* outb) {
    NEED(c, xa->space == GV_SPACE_M && xb->space == GV_SPACE_M && outa->space == GV_SPACE_N && outb->space == GV_SPACE_N &&
                outa != outb, "gv_ax2_dev: x M-space, out N-space, distinct outputs");
    if (ensure_work(c)) return 1;
    return ax2_device(c, xa->d, xb->d, outa->d, outb->d);
}
int gv_atx2_dev(gv_ctx* c, const gv_vec* pa, const gv_vec* pb, gv_vec* outa, gv_vec* outb) {
    NEED(c, pa->space == GV_SPACE_N && pb->space == GV_SPACE_N && outa->space == GV_SPACE_M && outb->space == GV_SPACE_M &&
                outa != outb, "gv_atx2_dev: p N-space, out M-space, distinct outputs");
    if (ensure_work(c)) return 1;
    return atx2_device(c, pa->d, pb->d, outa->d, outb->d);
}

int gv_ax(gv_ctx* c, const double* x, double* out) {
    if (ensure_work(c)) return 1;
    if (to_device(c, c->cg_d->d, x, sizeof(double) * c->M, false)) return 1;   // the kernels queue up behind the copy
    if (ax_device(c, c->cg_d->d, c->w_n->d)) return 1;
    return to_host(c, out, c->w_n->d, sizeof(double) * 4 * c->mbytes);
}
int gv_atx(gv_ctx* c, const double* p, double* out) {
    if (ensure_work(c)) return 1;
    NEED(c, c->mask2, "gv_atx: the phenotype mask must be set first");
    if (to_device(c, c->w_n->d, p, sizeof(double) * 4 * c->mbytes, false)) return 1;
    // The kernels (like data::dot_product, data.cpp:728-801, which applies no mask) need p = 0 at NA-phenotype and pad
    // slots; the reference's callers hand in filter_pheno()'d vectors.  A caller-owned host vector is not trusted to be
    // filtered -- data::get_phen() carries DBL_MAX at NA individuals (data.cpp:147) -- so the staged copy is masked here:
    // a no-op for filtered input, a defined result (the NA individuals dropped) otherwise.
    gvk::mask_copy(c->stream, c->w_n->d, c->w_n->d, c->mask2, c->npad);
    KCHK(c);
    if (atx_device(c, c->w_n->d, c->cg_d->d)) return 1;
    return to_host(c, out, c->cg_d->d, sizeof(double) * (c->M > 0 ? c->M : 0));
}

int gv_set_phen(gv_ctx* c, gv_vec* y_out, const double* y_host) {
    NEED(c, y_out->space == GV_SPACE_N, "gv_set_phen: y_out must be N-space");
    if (ensure_work(c)) return 1;
    std::vector<double> tmp(c->npad, 0.0);
    memcpy(tmp.data(), y_host, sizeof(double) * c->N);
    HIPCHK(c, hipMemcpyAsync(c->w_n->d, tmp.data(), sizeof(double) * c->npad, hipMemcpyHostToDevice, c->stream));
    gvk::mask_copy(c->stream, y_out->d, c->w_n->d, c->mask2, c->npad);
    KCHK(c);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- denoiser side ------------------------------------------------------------------------------------------
static int fill_prior(gv_ctx* c, gv_prior& pr, const double* probs, const double* vars, int L) {
    NEED(c, L >= 1 && L <= GV_LMAX, "prior: 1 <= L <= 32");
    pr.L = L;
    for (int i = 0; i < GV_LMAX; i++) {
        pr.probs[i] = i < L ? probs[i] : 0.0;
        pr.vars[i] = i < L ? vars[i] : 0.0;
    }
    return 0;
}

static int denoise_impl(gv_ctx* c, const gv_vec* r1, double gam1, const double* probs, const double* vars, int L,
                        gv_vec* x1_out, gv_vec* d_out, double* sums2, bool global) {
    NEED(c, r1->space == GV_SPACE_M && x1_out->space == GV_SPACE_M, "gv_denoise: M-space vectors required");
    gv_prior pr;
    if (fill_prior(c, pr, probs, vars, L)) return 1;
    const bool multi = global && is_multi(c);
    if (!multi) arm_scalars(c);
    gvk::denoise(c->stream, r1->d, c->M, gam1, pr, x1_out->d, d_out ? d_out->d : nullptr, c->red_partial, c->red_out);
    KCHK(c);
    return read_scalars_global(c, 2, sums2, multi);       // (sharded: all-reduced on the device buffer, in stream, before the read-back)
}
int gv_denoise(gv_ctx* c, const gv_vec* r1, double gam1, const double* probs, const double* vars, int L,
               gv_vec* x1_out, gv_vec* d_out, double* sums2) {
    return denoise_impl(c, r1, gam1, probs, vars, L, x1_out, d_out, sums2, false);
}
int gv_denoise_global(gv_ctx* c, const gv_vec* r1, double gam1, const double* probs, const double* vars, int L,
                      gv_vec* x1_out, gv_vec* d_out, double* sums2) {
    return denoise_impl(c, r1, gam1, probs, vars, L, x1_out, d_out, sums2, true);
}

static int prior_estep_impl(gv_ctx* c, const gv_vec* r1, double gam1, double lambda, const double* omegas,
                            const double* vars, int L, double* sums, bool global) {
    NEED(c, r1->space == GV_SPACE_M, "gv_prior_estep: M-space vector required");
    NEED(c, L >= 2, "gv_prior_estep: L >= 2");
    gv_prior pr;
    if (fill_prior(c, pr, omegas, vars, L)) return 1;
    const bool multi = global && is_multi(c);
    if (!multi) arm_scalars(c);
    gvk::prior_estep(c->stream, r1->d, c->M, gam1, lambda, pr, c->red_partial, c->red_out);
    KCHK(c);
    return read_scalars_global(c, 1 + 2 * (L - 1), sums, multi);
}
int gv_prior_estep(gv_ctx* c, const gv_vec* r1, double gam1, double lambda, const double* omegas,
                   const double* vars, int L, double* sums) {
    return prior_estep_impl(c, r1, gam1, lambda, omegas, vars, L, sums, false);
}
int gv_prior_estep_global(gv_ctx* c, const gv_vec* r1, double gam1, double lambda, const double* omegas,
                          const double* vars, int L, double* sums) {
    return prior_estep_impl(c, r1, gam1, lambda, omegas, vars, L, sums, true);
}

// ---- probit: z-side denoiser of vamp::infere_bin_class (vamp_probit.cpp:335-352) -------------------------------------
int gv_probit_denoise_cov(gv_ctx* c, const gv_vec* p1, const gv_vec* y, const gv_vec* m_cov, double tau1, double probit_var,
                          gv_vec* z1_out, double* sums2) {
    NEED(c, p1->space == GV_SPACE_N && y->space == GV_SPACE_N && z1_out->space == GV_SPACE_N &&
                (!m_cov || m_cov->space == GV_SPACE_N),
         "gv_probit_denoise: N-space vectors required");
    arm_scalars(c);
    gvk::probit_denoise(c->stream, p1->d, y->d, m_cov ? m_cov->d : nullptr, c->N, c->npad, tau1, probit_var, z1_out->d,
                        c->red_partial, c->red_out);
    KCHK(c);
    return read_scalars(c, 2, sums2);
}
int gv_probit_denoise(gv_ctx* c, const gv_vec* p1, const gv_vec* y, double tau1, double probit_var, gv_vec* z1_out,
                      double* sums2) {
    return gv_probit_denoise_cov(c, p1, y, nullptr, tau1, probit_var, z1_out, sums2);
}

// ---- --use-XXT-denoiser 1: LMMSE through CG in N-space (denoiserXXT.cpp), matrix-free ------------------------------
// data::compute_people_statistics (data.cpp:558-716): three table passes of the fp64 Ax kernel over the raw rows.
int gv_people_stats(gv_ctx* c, double* mave_people, double* msig_people, double* numb_people) {
    NEED(c, c->have_stats && c->mask2, "gv_people_stats: marker statistics must be computed first");
    const bool from_stripes = c->have_stripes && (c->kernel_mode != 0 || !c->have_raw);
    NEED(c, c->have_raw || from_stripes, "gv_people_stats: no genotype layout resident");
    if (ensure_work(c)) return 1;
    for (gv_vec** v : {&c->mave_p, &c->msig_p, &c->numb_p})
        if (!*v && vec_new(c, GV_SPACE_N, v)) return 1;
    gv_vec* dst[3] = {c->mave_p, c->numb_p, c->msig_p};   // kinds 0 (sum value), 1 (count), 2 (sum value^2)
    if (from_stripes) {
        // four passes over stripes_n in exact fixed point: the sum is an Ax of the all-ones vector, the count and the
        // two halves of the sum of squares have their own operand tables (k_prep_people); the quadratic half reads the
        // a^2 plane of the codes (MODE 4 of the streaming kernel)
        hipStream_t s = c->stream;
        if (c->M == 0) {      // an empty shard adds zeros to the three sums, through the same collectives as its peers
            for (int kind = 0; kind < 3; kind++) gvk::fill(s, dst[kind]->d, c->npad, 0.0);
        } else {
            double* ones = c->cg_d->d;
            gvk::fill(s, ones, c->M, 1.0);
            gvm::ax(s, c->plan, ones, c->mave, c->msig, c->mask2, c->npad, 1.0, c->red_partial, c->mave_p->d);
            gvm::ax_people(s, c->plan, 0, c->mave, c->msig, c->mask2, c->npad, c->red_partial, c->numb_p->d);
            gvm::ax_people(s, c->plan, 1, c->mave, c->msig, c->mask2, c->npad, c->red_partial, c->msig_p->d);
            gvm::ax_people(s, c->plan, 2, c->mave, c->msig, c->mask2, c->npad, c->red_partial, c->w_n->d);
            gvk::axpby(s, c->msig_p->d, 1.0, c->msig_p->d, 1.0, c->w_n->d, c->npad);
        }
        KCHK(c);
        for (int kind = 0; kind < 3; kind++)
            if (comm_allreduce(c, dst[kind]->d, c->npad)) return 1;     // data.cpp:604-606
    }
    for (int kind = 0; kind < 3 && !from_stripes; kind++) {
        gvk::people_table(c->stream, c->mave, c->msig, c->M, kind, c->t3);
        gvk::ax_f64(c->stream, c->bed, c->M, c->pitch, c->t3, c->ax_chunks, c->ax_partial, c->npad);
        gvk::ax_reduce(c->stream, c->ax_partial, c->ax_chunks, c->npad, c->mask2, 1.0, dst[kind]->d);
        KCHK(c);
        if (comm_allreduce(c, dst[kind]->d, c->npad)) return 1;     // data.cpp:604-606
    }
    gvk::people_finish(c->stream, c->mave_p->d, c->msig_p->d, c->numb_p->d, c->mask2, c->N, c->npad);
    KCHK(c);
    const size_t n4 = sizeof(double) * 4 * c->mbytes;
    if (mave_people) HIPCHK(c, hipMemcpyAsync(mave_people, c->mave_p->d, n4, hipMemcpyDeviceToHost, c->stream));
    if (msig_people) HIPCHK(c, hipMemcpyAsync(msig_people, c->msig_p->d, n4, hipMemcpyDeviceToHost, c->stream));
    if (numb_people) HIPCHK(c, hipMemcpyAsync(numb_people, c->numb_p->d, n4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- p-values: data::pvals_calc (data.cpp:1108-1226) and pvals_calc_LOCO (:1235-1353), one estimator ----------------
// Kernel mode 1: per call ONE pass over the shard whose epilogue runs the per-marker regression test (gvm::marker_pvals: operands,
// digits, stream, test -- four launches, no allocation: the operands live in the context's N-space scratch, the p-values in an
// M-space work vector until they are copied out).  Kernel mode 0 (fp64 family, parity anchor): the sums of k_marker_sums2_f64, then
// the stand-alone test kernel.
// out4[4m..] = {sum a p, sum b p, sum a p^2, sum b p^2} for the N-space device vector p, fp64 family
static int marker_sums_p_p2_f64(gv_ctx* c, const double* p, double* p2_scratch, double* out4_dev) {
    NEED(c, c->have_raw, "p-values: kernel mode 0 needs the raw row layout");
    gvk::mul(c->stream, p2_scratch, p, p, c->npad);
    gvk::marker_sums2_f64(c->stream, c->bed, c->M, c->pitch, p, p2_scratch, out4_dev);
    KCHK(c);
    return 0;
}

// chrom == NULL: leave-one-out (the marker's own effect is added back analytically); else leave-one-chromosome-out.
static int pvals_impl(gv_ctx* c, const gv_vec* z1, const gv_vec* y, const gv_vec* x1_hat, const int* chrom,
                      double* pvals, double* chrom_pred = nullptr) {
    NEED(c, z1->space == GV_SPACE_N && y->space == GV_SPACE_N && x1_hat->space == GV_SPACE_M, "gv_pvals: bad vector spaces");
    NEED(c, c->have_stats, "gv_pvals: marker statistics must be computed first");
    if (ensure_work(c) || ensure_w2(c)) return 1;
    const int64_t M = c->M;
    const double sqrtN = sqrt((double)c->N);
    const bool fused = c->kernel_mode != 0;      // (kernel mode 2: the p-value pass is mode 1's -- its sums run over exact planes already)
    if (fused && M > 0) {
        NEED(c, c->have_stripes, "p-values: kernel modes 1 and 2 need a re-encoded layout");
        if (!c->ks_tuned && autotune_ks(c)) return 1;      // (a p-value call may be the first streaming pass of a context)
    }
    gv_vec *ymod = nullptr, *ych = nullptr, *sq = nullptr, *xch = nullptr;
    double* sums_dev = nullptr;
    int* chrom_dev = nullptr;
    int rc = 0;
    const size_t Mn = (size_t)(M > 0 ? M : 1);
    auto cleanup = [&]() {
        for (gv_vec* v : {ymod, ych, sq, xch}) vec_del(c, v);
        if (sums_dev) (void)hipFree(sums_dev);
        if (chrom_dev) (void)hipFree(chrom_dev);
    };
#define PV_TRY(expr) do { if ((rc = (expr)) != 0) { cleanup(); return rc; } } while (0)
#define PV_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); return fail(c, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    // an M-space work vector of the CG: INVARIANT -- ax_device / ax_overlapped (run per chromosome between the marker passes of the LOCO
    // loop below) never touch the CG work vectors cg_r / cg_z / cg_p / cg_d; gv_ax / gv_atx (which stage through cg_d) are host entry
    // points and cannot run inside this call
    double* pv_dev = c->cg_d->d;
    double *pa = c->w_n->d, *pb = c->w_n2->d;   // operands of the fused pass: p and p^2
    gvm::PvArgs pva{c->counts, nullptr, 0.0, nullptr, 0};
    if (!fused) {
        PV_TRY(vec_new(c, GV_SPACE_N, &ymod));
        PV_TRY(vec_new(c, GV_SPACE_N, &sq));
        PV_HIP(hipMalloc(&sums_dev, sizeof(double) * 4 * Mn));
        gvk::axpby(c->stream, ymod->d, 1.0, y->d, -1.0, z1->d, c->npad);            // y_mod = y - z1 (data.cpp:1117-1119)
        // the reference masks every term of the regression sums with na_lut[mask4] (data.cpp:1155-1175); the sums here are
        // matvec-shaped and need zeros at NA / pad slots instead, whatever the caller left there (an unfiltered y holds DBL_MAX)
        gvk::mask_copy(c->stream, ymod->d, ymod->d, c->mask2, c->npad);
    }
    if (!chrom) {
        // y_mark = y_mod + gen_part * x1_hat[k] (data.cpp:1145-1148): the marker's own column, c = x1_hat[k] / sqrt(N)
        if (fused) {
            pva.xself = x1_hat->d;
            pva.self_scale = 1.0 / sqrtN;
            if (M > 0)      // (an empty shard has no marker to test)
                gvm::marker_pvals(c->stream, c->plan, y->d, z1->d, nullptr, c->mask2, c->npad, c->mave, c->msig, pa, pb, c->red_partial, pva, pv_dev);
        } else if (M > 0) {
            PV_TRY(marker_sums_p_p2_f64(c, ymod->d, sq->d, sums_dev));
            gvk::pvals_test(c->stream, c->counts, c->mave, c->msig, sums_dev, x1_hat->d, 1.0 / sqrtN, nullptr, 0, M, pv_dev);
        }
    } else {
        PV_HIP(hipMemsetAsync(pv_dev, 0, sizeof(double) * Mn, c->stream));      // markers of chromosomes outside 1..23 keep 0
        PV_TRY(vec_new(c, GV_SPACE_N, &ych));
        PV_TRY(vec_new(c, GV_SPACE_M, &xch));
        PV_HIP(hipMalloc(&chrom_dev, sizeof(int) * Mn));
        PV_HIP(hipMemcpyAsync(chrom_dev, chrom, sizeof(int) * M, hipMemcpyHostToDevice, c->stream));
        double present[24];
        for (int ch = 0; ch < 24; ch++) present[ch] = 0;
        for (int64_t k = 0; k < M; k++) if (chrom[k] >= 1 && chrom[k] <= 23) present[chrom[k]] += 1;
        PV_TRY(allreduce_scalars(c, present, 24));
        if (chrom_pred) memset(chrom_pred, 0, sizeof(double) * 23 * 4 * (size_t)c->mbytes);   // chromosomes nobody holds: zeros
        for (int ch = 1; ch <= 23; ch++) {
            if (present[ch] == 0) continue;      // no rank holds a marker of this chromosome
            gvk::select_eq(c->stream, xch->d, x1_hat->d, chrom_dev, ch, M);
            PV_TRY(ax_device(c, xch->d, ych->d));                                // chromosome predictor, all ranks (:1268-1272)
            if (chrom_pred)                                                      // the vector the reference dumps (:1276-1281)
                PV_TRY(to_host(c, chrom_pred + (size_t)(ch - 1) * 4 * c->mbytes, ych->d, sizeof(double) * 4 * c->mbytes));
            if (fused) {                                                         // p = chromosome predictor + y_mod (:1284)
                pva.chrom = chrom_dev;
                pva.ch = ch;
                if (M > 0)
                    gvm::marker_pvals(c->stream, c->plan, y->d, z1->d, ych->d, c->mask2, c->npad, c->mave, c->msig, pa, pb, c->red_partial, pva, pv_dev);
            } else if (M > 0) {
                gvk::axpby(c->stream, ych->d, 1.0, ych->d, 1.0, ymod->d, c->npad);
                PV_TRY(marker_sums_p_p2_f64(c, ych->d, sq->d, sums_dev));
                gvk::pvals_test(c->stream, c->counts, c->mave, c->msig, sums_dev, nullptr, 0.0, chrom_dev, ch, M, pv_dev);
            }
        }
    }
    KCHK(c);
    if (M > 0) PV_TRY(to_host(c, pvals, pv_dev, sizeof(double) * M));
#undef PV_TRY
#undef PV_HIP
    if (ymod || ych || sq || xch || sums_dev || chrom_dev) HIPCHK(c, hipStreamSynchronize(c->stream));
    cleanup();
    return 0;
}

int gv_pvals_loo(gv_ctx* c, const gv_vec* z1, const gv_vec* y, const gv_vec* x1_hat, double* pvals) {
    return pvals_impl(c, z1, y, x1_hat, nullptr, pvals);
}
int gv_pvals_loco(gv_ctx* c, const gv_vec* z1, const gv_vec* y, const gv_vec* x1_hat, const int* chrom, double* pvals) {
    NEED(c, chrom != nullptr, "gv_pvals_loco: chrom is NULL");
    return pvals_impl(c, z1, y, x1_hat, chrom, pvals);
}
int gv_pvals_loco_pred(gv_ctx* c, const gv_vec* z1, const gv_vec* y, const gv_vec* x1_hat, const int* chrom, double* pvals,
                       double* chrom_pred) {
    NEED(c, chrom != nullptr, "gv_pvals_loco_pred: chrom is NULL");
    return pvals_impl(c, z1, y, x1_hat, chrom, pvals, chrom_pred);
}

int gv_allreduce_host(gv_ctx* c, double* buf, int n) {
    for (int off = 0; off < n; off += RED_MAXK) {
        int k = n - off < RED_MAXK ? n - off : RED_MAXK;
        if (allreduce_scalars(c, buf + off, k)) return 1;
    }
    return 0;
}

// ---- communicator ---------------------------------------------------------------------------------------------
// the context lets go of whatever communicator it holds (the RCCL one is destroyed when its last sharer does)
static void comm_drop(gv_ctx* c) {
    c->comm = nullptr;
    c->comm_keep.reset();
    c->local = nullptr;
    c->local_keep.reset();
    c->cb = nullptr;
    c->cb_user = nullptr;
    c->rank = 0;
    c->nranks = 1;
    c->force_multi = 0;
}
int gv_comm_share(gv_ctx* c, const gv_ctx* owner) {
    NEED(c, owner != nullptr && owner != c, "gv_comm_share: owner is NULL or the context itself");
    NEED(c, owner->device == c->device || !owner->comm, "gv_comm_share: an RCCL communicator belongs to its device");
    comm_drop(c);
    c->comm = owner->comm;
    c->comm_keep = owner->comm_keep;
    c->local = owner->local;
    c->local_keep = owner->local_keep;
    c->cb = owner->cb;
    c->cb_user = owner->cb_user;
    c->rank = owner->rank;
    c->nranks = owner->nranks;
    return 0;
}
int gv_comm_unique_id(void* id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, "ncclGetUniqueId failed: %s", ncclGetErrorString(r));
    memcpy(id128, &id, 128);
    return 0;
}
int gv_comm_init(gv_ctx* c, int nranks, int rank, const void* id128) {
    NEED(c, nranks >= 1 && rank >= 0 && rank < nranks, "gv_comm_init: bad rank / nranks");
    HIPCHK(c, hipSetDevice(c->device));
    comm_drop(c);
    c->rank = rank;
    c->nranks = nranks;
    if (nranks == 1 && !id128) return 0;
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    NCCLCHK(c, ncclCommInitRank(&comm, nranks, id, rank));
    c->comm = comm;
    c->comm_keep = std::shared_ptr<void>(comm, [](void* p) { (void)ncclCommDestroy(static_cast<ncclComm_t>(p)); });
    // self-test: a 4-double SUM all-reduce on the context's stream must give nranks * (rank-independent value)
    double probe[4] = {1.0, 2.0, 3.0, 4.0};
    memcpy(c->host_pin, probe, sizeof(probe));
    HIPCHK(c, hipMemcpyAsync(c->red_out, c->host_pin, sizeof(probe), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, ncclAllReduce(c->red_out, c->red_out, 4, ncclDouble, ncclSum, c->comm, c->stream));
    double back[4];
    if (read_scalars(c, 4, back)) return 1;
    for (int i = 0; i < 4; i++)
        if (back[i] != probe[i] * nranks) return fail(c, "gv_comm_init: RCCL all-reduce self-test failed (%g != %g)", back[i], probe[i] * nranks);
    return 0;
}
int gv_comm_init_local(gv_ctx* c, int group, int nranks, int rank) {
    NEED(c, nranks >= 1 && rank >= 0 && rank < nranks, "gv_comm_init_local: bad rank / nranks");
    comm_drop(c);
    std::lock_guard<std::mutex> lk(g_groups_mu);
    std::shared_ptr<LocalGroup>& g = g_groups[group];
    if (!g || g->n != nranks) {
        g = std::make_shared<LocalGroup>();
        g->n = nranks;
        g->slots.assign(nranks, nullptr);
    }
    c->local_keep = g;
    c->local = g.get();
    c->cb = nullptr;
    c->cb_user = nullptr;
    c->rank = rank;
    c->nranks = nranks;
    return 0;
}
int gv_comm_init_callback(gv_ctx* c, int nranks, int rank, gv_allreduce_fn fn, void* user) {
    NEED(c, nranks >= 1 && rank >= 0 && rank < nranks, "gv_comm_init_callback: bad rank / nranks");
    NEED(c, fn != nullptr, "gv_comm_init_callback: fn is NULL");
    comm_drop(c);
    c->cb = fn;
    c->cb_user = user;
    c->rank = rank;
    c->nranks = nranks;
    // self-test, as for RCCL: every rank must see nranks * (rank-independent value)
    double probe[4] = {1.0, 2.0, 3.0, 4.0}, back[4];
    memcpy(back, probe, sizeof(probe));
    if (nranks > 1) {
        if (allreduce_scalars(c, back, 4)) return 1;
        for (int i = 0; i < 4; i++)
            if (back[i] != probe[i] * nranks)
                return fail(c, "gv_comm_init_callback: all-reduce self-test failed (%g != %g)", back[i], probe[i] * nranks);
    }
    return 0;
}
// Test hook (include/gvamp.h): transport 0 = off, 1 = loop-back through scratch, 2 = the 1-rank RCCL communicator (created here when
// the context holds none), 3 = RCCL then the loop-back.  Only a context of a one-rank job may be forced.
int gv_debug_force_multi(gv_ctx* c, int transport, int delay_us) {
    NEED(c, transport >= 0 && transport <= 7 && (transport == 0 || (transport & 3)) && delay_us >= 0,
         "gv_debug_force_multi: transport 0..3 (+ 4: fault injection), delay_us >= 0");
    NEED(c, transport == 0 || (c->nranks == 1 && !c->local && !c->cb), "gv_debug_force_multi: only a one-rank context can be forced");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (c->comm_stream) HIPCHK(c, hipStreamSynchronize(c->comm_stream));     // work a dropped join (bit 4) left behind
    if ((transport & 2) && !c->comm) {
        ncclUniqueId id;
        NCCLCHK(c, ncclGetUniqueId(&id));
        if (gv_comm_init(c, 1, 0, &id)) return 1;
    }
    c->force_multi = transport;
    c->loop_delay_us = delay_us;
    return 0;
}
int gv_set_overlap(gv_ctx* c, int tiles) {
    NEED(c, tiles >= 0 && tiles <= 64, "gv_set_overlap: 0 <= tiles <= 64");
    c->overlap_tiles = tiles;
    return 0;
}
int gv_comm_rank(const gv_ctx* c) { return c->rank; }
int gv_comm_size(const gv_ctx* c) { return c->nranks; }

// ---- instrumentation ----------------------------------------------------------------------------------------------
int gv_set_timing(gv_ctx* c, int timing) {
    c->timing = timing;
    return 0;
}
int gv_get_counters(gv_ctx* c, gv_counters* out) {
    ev_resolve(c);
    *out = c->cnt;
    return 0;
}
int gv_reset_counters(gv_ctx* c) {
    ev_resolve(c);
    c->cnt = gv_counters{};
    return 0;
}
int gv_get_layout(const gv_ctx* c) { return c->have_stripes ? (c->plan.layout == 1 ? 2 : 1) : 0; }
int gv_ingest_info(gv_ctx* c, double* alloc_seconds, double* fill_seconds) {
    if (alloc_seconds) *alloc_seconds = c->ingest_alloc_s;
    if (fill_seconds) *fill_seconds = c->ingest_fill_s;
    return 0;
}
// The calling thread (and the threads it starts afterwards: the staging-copy helpers, the file readers) is restricted to the CPUs of
// the NUMA node the GPU hangs off -- /sys/bus/pci/devices/<bus id>/numa_node and /sys/devices/system/node/node<k>/cpulist --
// intersected with the CPUs it may already run on.  The host-paced sections of a VAMP iteration and the pinned staging copies
// cross the socket interconnect otherwise (18.9 % idle measured on a box whose host was the far socket).
int gv_bind_host_numa(int device, int* numa_node_out) {
    if (numa_node_out) *numa_node_out = -1;
    if (const char* e = getenv("GVAMP_NUMA_BIND"))
        if (atoi(e) == 0) return 0;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, "gv_bind_host_numa: no PCI bus id for device %d", device); }
    for (char* q = bus; *q; q++) *q = (char)tolower((unsigned char)*q);
    char path[256];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
    FILE* f = fopen(path, "r");
    int node = -1;
    if (f) { if (fscanf(f, "%d", &node) != 1) node = -1; fclose(f); }
    if (node < 0) return 0;                       // a single-node host (or a VM that hides the topology): nothing to do
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f) return 0;
    char list[4096] = {0};
    const bool got = fgets(list, sizeof(list), f) != nullptr;
    fclose(f);
    if (!got) return 0;
    cpu_set_t want, have, both;
    CPU_ZERO(&want);
    for (char* q = list; *q;) {                   // "0-15,128-143"
        char* end = nullptr;
        const long a = strtol(q, &end, 10);
        if (end == q) break;
        long b = a;
        if (*end == '-') { q = end + 1; b = strtol(q, &end, 10); }
        for (long k = a; k <= b && k < CPU_SETSIZE; k++) CPU_SET((int)k, &want);
        q = (*end == ',') ? end + 1 : end;
        if (*end != ',') break;
    }
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return 0;
    CPU_AND(&both, &want, &have);
    // fewer than 8 CPUs in common (a launcher that pinned this rank elsewhere, a cgroup that grants a sliver of the node): leave the
    // affinity alone -- the rank's main thread spins on the scalar mailbox, and the staging helpers, the file readers and RCCL's
    // proxy threads need cores of their own beside it
    if (CPU_COUNT(&both) < 8) return 0;
    if (sched_setaffinity(0, sizeof(both), &both) != 0) return 0;
    if (numa_node_out) *numa_node_out = node;
    return 0;
}
int gv_ingest_info2(gv_ctx* c, gv_ingest_stats* out) {
    NEED(c, out != nullptr, "gv_ingest_info2: out is NULL");
    out->alloc_seconds = c->ingest_alloc_s;
    out->fill_seconds = c->ingest_fill_s;
    out->overlap_seconds = c->ingest_overlap_s;
    out->resident_bytes = (double)c->ingest_bytes;
    out->layout = gv_get_layout(c);
    out->expected_passes = c->expected_passes;
    return 0;
}
int gv_set_expected_passes(gv_ctx* c, int64_t passes) {
    NEED(c, passes >= 0, "gv_set_expected_passes: passes >= 0 (0 = unknown)");
    c->expected_passes = passes;
    return 0;
}
int gv_tune_info(gv_ctx* c, double* seconds, int* source) {
    if (seconds) *seconds = c->tune_seconds;
    if (source) *source = c->ks_tuned ? c->tune_source : -1;
    return 0;
}
int gv_get_decomp(gv_ctx* c, gv_decomp_info* out4) {
    NEED(c, out4 != nullptr, "gv_get_decomp: out is NULL");
    const gvm::Decomp* d[4] = {&c->plan.dm[0], &c->plan.dm[1], &c->plan.dn[0], &c->plan.dn[1]};
    for (int k = 0; k < 4; k++) {
        out4[k].ks = d[k]->ks;
        out4[k].balanced_cells = d[k]->skL;
        out4[k].whole_quads = d[k]->skL > 0 ? d[k]->piv : 0;
        out4[k].prio = d[k]->prio;
        out4[k].taper = d[k]->taper;
        out4[k].geo = d[k]->geo;
        out4[k].wgs_per_cu = d[k]->occ == 2 ? 2 : 3;
        out4[k].xcd_skew = d[k]->xskew;
        out4[k].tuned = c->ks_tuned ? 1 : 0;
    }
    return 0;
}
int gv_set_decomp(gv_ctx* c, int cls, const gv_decomp_info* in) {
    NEED(c, cls >= 0 && cls <= 3 && in != nullptr, "gv_set_decomp: class 0..3 and a decomposition are required");
    NEED(c, c->have_stripes, "gv_set_decomp: no re-encoded layout resident yet (call it after the ingest)");
    gvm::Decomp d;
    d.ks = in->ks; d.skL = in->balanced_cells; d.piv = in->balanced_cells > 0 ? in->whole_quads : 0; d.prio = in->prio;
    d.taper = in->taper; d.geo = in->geo;
    NEED(c, in->wgs_per_cu == 0 || in->wgs_per_cu == 2 || in->wgs_per_cu == 3, "gv_set_decomp: wgs_per_cu is 0 (default), 2 or 3");
    d.occ = in->wgs_per_cu == 2 ? 2 : 0;
    d.xskew = in->balanced_cells > 0 ? 0.f : in->xcd_skew;
    if (d.skL > 0) d.ks = 1;
    NEED(c, decomp_ok(c, d, cls >> 1), "gv_set_decomp: the decomposition is not admissible for this shard (range, or too many pieces for the partial-sum buffer)");
    (cls >> 1 ? c->plan.dn : c->plan.dm)[cls & 1] = d;
    return 0;
}
int gv_copy_bandwidth(gv_ctx* c, size_t nbytes, int reps, double* gbps) {
    double *a = nullptr, *b = nullptr;
    int64_t n = (int64_t)(nbytes / 16) * 2;
    HIPCHK(c, hipMalloc(&a, n * 8));
    HIPCHK(c, hipMalloc(&b, n * 8));
    HIPCHK(c, hipMemsetAsync(a, 1, n * 8, c->stream));
    gvk::copy_bw(c->stream, a, b, n);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; i++) gvk::copy_bw(c->stream, a, b, n);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *gbps = 2.0 * n * 8 * reps / (ms * 1e-3) / 1e9;
    (void)hipFree(a);
    (void)hipFree(b);
    return 0;
}

// read-only stream probe: the resident stripes when there are any (the very bytes the matvecs stream), else a scratch
// buffer of nbytes.  ~45k waves of equal contiguous runs, like a matvec launch.
int gv_read_bandwidth(gv_ctx* c, size_t nbytes, int reps, double* gbps) {
    void* buf = nullptr;
    bool own = false;
    size_t have = 0;
    if (c->have_stripes && (c->plan.stripes_m || c->plan.tiles)) {
        buf = c->plan.layout == 1 ? c->plan.tiles : c->plan.stripes_m;
        have = (size_t)c->plan.nrg_m * c->plan.nkb_m * 4096;
    }
    if (have < (size_t)64 << 20) {
        have = nbytes < ((size_t)64 << 20) ? ((size_t)64 << 20) : nbytes;
        HIPCHK(c, hipMalloc(&buf, have));
        HIPCHK(c, hipMemsetAsync(buf, 1, have, c->stream));
        own = true;
    }
    const int64_t blocks = (int64_t)(have / 4096);
    int64_t nwaves = 45056;
    if (blocks / nwaves < 16) nwaves = blocks / 16 > 0 ? blocks / 16 : 1;
    const int64_t bpw = blocks / nwaves;
    unsigned int* sink = reinterpret_cast<unsigned int*>(c->red_out);
    // lane -> piece pattern of the loads: that of the resident layout's kernels (GV_READ_PERM = 0 / 1 / 2 overrides: linear, tile
    // layout ATx side, tile layout Ax side)
    int perm = (c->have_stripes && c->plan.layout == 1) ? 1 : 0;
    if (const char* e = getenv("GV_READ_PERM")) perm = atoi(e);
    gvk::read_bw(c->stream, buf, bpw, nwaves, sink, perm);
    HIPCHK(c, hipEventRecord(c->ev0, c->stream));
    for (int i = 0; i < reps; i++) gvk::read_bw(c->stream, buf, bpw, nwaves, sink, perm);
    HIPCHK(c, hipEventRecord(c->ev1, c->stream));
    HIPCHK(c, hipEventSynchronize(c->ev1));
    KCHK(c);
    float ms = 0;
    HIPCHK(c, hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *gbps = (double)bpw * nwaves * 4096.0 * reps / (ms * 1e-3) / 1e9;
    if (own) (void)hipFree(buf);
    return 0;
}

}  // extern "C"
