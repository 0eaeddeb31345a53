// shm_comm.cpp -- see shm_comm.hpp.
#include "shm_comm.hpp"

#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace {
constexpr uint32_t MAGIC = 0x67764853u;   // "gvHS"
struct alignas(64) Hdr {
    std::atomic<uint32_t> magic;          // set last by rank 0
    uint32_t nranks;
    uint64_t cap;                         // doubles per slot
    int32_t creator_pid;
    std::atomic<uint32_t> joined;         // ranks > 0 that have attached
    std::atomic<uint32_t> go;             // rank 0: everybody is here, the name is gone
    alignas(64) std::atomic<uint32_t> bar_count;
    alignas(64) std::atomic<uint32_t> bar_gen;
    alignas(64) std::atomic<uint32_t> failed;   // a rank gave up: everybody else stops waiting
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "process-shared atomics must be lock-free");
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// wait until pred() holds; false on timeout or when a peer flagged failure
template <class P>
bool spin_until(P pred, double timeout_s, const std::atomic<uint32_t>* failed) {
    const double t0 = now_s();
    for (unsigned long it = 0;; it++) {
        if (pred()) return true;
        if (it < 4000) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        } else {
            sched_yield();
            if ((it & 0x3FF) == 0) {
                if (failed && failed->load(std::memory_order_acquire)) return false;
                if (now_s() - t0 > timeout_s) return false;
            }
        }
    }
}
}  // namespace

struct gvh_shm_comm {
    int nranks = 1, rank = 0;
    size_t cap = 0, bytes = 0;
    Hdr* hdr = nullptr;
    double* slots = nullptr;
    bool barrier() {
        const uint32_t g = hdr->bar_gen.load(std::memory_order_acquire);
        if (hdr->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)nranks) {
            hdr->bar_count.store(0, std::memory_order_relaxed);
            hdr->bar_gen.fetch_add(1, std::memory_order_release);
            return true;
        }
        const bool ok = spin_until([&] { return hdr->bar_gen.load(std::memory_order_acquire) != g; }, 600.0, &hdr->failed);
        if (!ok) hdr->failed.store(1, std::memory_order_release);
        return ok;
    }
};

std::string gvh_shm_default_name() {
    std::string key;
    if (const char* r = getenv("GVAMP_RENDEZVOUS")) key = r;
    else key = std::string(getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0") + "_" + std::to_string((long)getppid());
    return "/gvamp_" + std::to_string(std::hash<std::string>{}(key));
}

gvh_shm_comm* gvh_shm_open_impl(const std::string& name, int nranks, int rank, size_t cap_doubles, std::string& err) {
    if (nranks < 1 || rank < 0 || rank >= nranks || cap_doubles == 0) { err = "gvh_shm_open: bad rank / nranks / capacity"; return nullptr; }
    const size_t hdr_bytes = (sizeof(Hdr) + 4095) / 4096 * 4096;
    const size_t bytes = hdr_bytes + sizeof(double) * cap_doubles * (size_t)nranks;
    int fd = -1;
    if (rank == 0) {
        shm_unlink(name.c_str());                                   // whatever a dead job left under this name
        fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { err = "gvh_shm_open: cannot create " + name + ": " + strerror(errno); if (fd >= 0) close(fd); return nullptr; }
    } else {
        const double t0 = now_s();
        for (;;) {
            fd = shm_open(name.c_str(), O_RDWR, 0600);
            if (fd >= 0) {
                struct stat sb;
                if (fstat(fd, &sb) == 0 && (size_t)sb.st_size == bytes) break;      // rank 0 has sized it
                close(fd);
                fd = -1;
            }
            if (now_s() - t0 > 120.0) { err = "gvh_shm_open: rank 0 never created " + name; return nullptr; }
            usleep(2000);
        }
    }
    void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { err = std::string("gvh_shm_open: mmap failed: ") + strerror(errno); if (rank == 0) shm_unlink(name.c_str()); return nullptr; }
    gvh_shm_comm* c = new gvh_shm_comm();
    c->nranks = nranks; c->rank = rank; c->cap = cap_doubles; c->bytes = bytes;
    c->hdr = static_cast<Hdr*>(p);
    c->slots = reinterpret_cast<double*>(static_cast<char*>(p) + hdr_bytes);
    Hdr* h = c->hdr;
    auto bail = [&](const std::string& m) { err = m; munmap(p, bytes); delete c; return (gvh_shm_comm*)nullptr; };
    if (rank == 0) {
        h->nranks = (uint32_t)nranks; h->cap = cap_doubles; h->creator_pid = (int32_t)getpid();
        h->joined.store(0); h->go.store(0); h->bar_count.store(0); h->bar_gen.store(0); h->failed.store(0);
        h->magic.store(MAGIC, std::memory_order_release);
        const bool ok = spin_until([&] { return h->joined.load(std::memory_order_acquire) == (uint32_t)(nranks - 1); }, 120.0, nullptr);
        shm_unlink(name.c_str());                                   // nobody can find it any more; it lives while it is mapped
        if (!ok) return bail("gvh_shm_open: not every rank attached to " + name + " within 120 s");
        h->go.store(1, std::memory_order_release);
    } else {
        if (!spin_until([&] { return h->magic.load(std::memory_order_acquire) == MAGIC; }, 120.0, nullptr))
            return bail("gvh_shm_open: segment " + name + " was never initialised");
        // a segment whose creator is gone, or that already started, belongs to another (dead) job
        if (h->nranks != (uint32_t)nranks || h->cap != cap_doubles || h->go.load() != 0 || kill((pid_t)h->creator_pid, 0) != 0)
            return bail("gvh_shm_open: stale or foreign segment under " + name);
        h->joined.fetch_add(1, std::memory_order_acq_rel);
        if (!spin_until([&] { return h->go.load(std::memory_order_acquire) == 1; }, 120.0, nullptr))
            return bail("gvh_shm_open: rank 0 never released " + name);
    }
    return c;
}

extern "C" int gvh_shm_allreduce(void* comm, double* buf, size_t n) {
    gvh_shm_comm* c = static_cast<gvh_shm_comm*>(comm);
    if (!c) return 1;
    if (c->nranks == 1) return 0;
    for (size_t off = 0; off < n; off += c->cap) {
        const size_t k = n - off < c->cap ? n - off : c->cap;
        memcpy(c->slots + (size_t)c->rank * c->cap, buf + off, sizeof(double) * k);
        if (!c->barrier()) return 1;
        // every rank adds the slots in rank order: bit-identical sums everywhere
        const double* s0 = c->slots;
        double* out = buf + off;
        for (size_t i = 0; i < k; i++) out[i] = s0[i];
        for (int r = 1; r < c->nranks; r++) {
            const double* s = c->slots + (size_t)r * c->cap;
            for (size_t i = 0; i < k; i++) out[i] += s[i];
        }
        if (!c->barrier()) return 1;                               // the slots may be overwritten again
    }
    return 0;
}

void gvh_shm_close_impl(gvh_shm_comm* c) {
    if (!c) return;
    if (c->hdr) munmap(c->hdr, c->bytes);
    delete c;
}
