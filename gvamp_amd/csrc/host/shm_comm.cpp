// shm_comm.cpp -- see shm_comm.hpp.
#include "shm_comm.hpp"

#include <fcntl.h>
#include <sched.h>
#include <signal.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <functional>

namespace {
constexpr uint32_t MAGIC = 0x67764853u;   // "gvHS"
struct alignas(64) Hdr {
    std::atomic<uint32_t> magic;          // set last by rank 0
    uint32_t nranks;
    uint64_t cap;                         // doubles per slot
    int32_t creator_pid;
    uint64_t creator_start;               // start time of the creator (clock ticks since boot, /proc/<pid>/stat field 22): a pid that
                                          // has been reused since does not match
    uint64_t name_hash;                   // of the segment name (the job's rendezvous key)
    std::atomic<uint32_t> joined;         // ranks > 0 that have attached
    std::atomic<uint32_t> go;             // rank 0: everybody is here, the name is gone
    alignas(64) std::atomic<uint32_t> bar_count;
    alignas(64) std::atomic<uint32_t> bar_gen;
    alignas(64) std::atomic<uint32_t> failed;   // a rank gave up: everybody else stops waiting
};
static_assert(std::atomic<uint32_t>::is_always_lock_free, "process-shared atomics must be lock-free");
// start time of a process in clock ticks since boot (0: unknown / gone)
uint64_t proc_start_ticks(long pid) {
    char path[64], buf[1024];
    snprintf(path, sizeof(path), "/proc/%ld/stat", pid);
    FILE* f = fopen(path, "r");
    if (!f) return 0;
    const size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    const char* q = strrchr(buf, ')');          // the command name may hold spaces and parentheses: fields resume after the last ')'
    if (!q) return 0;
    unsigned long long v = 0;
    int field = 2;                              // q points behind field 2 (comm)
    for (const char* t = q + 1; *t; ) {
        while (*t == ' ') t++;
        if (!*t) break;
        field++;
        if (field == 22) { v = strtoull(t, nullptr, 10); break; }
        while (*t && *t != ' ') t++;
    }
    return (uint64_t)v;
}
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

// What the ranks of ONE job agree on without talking to each other: $GVAMP_RENDEZVOUS when the launcher set it
// (scripts/run_sharded.py does); else the job id of a known launcher -- torchrun, Slurm, PMIx / PMI (mpirun), in that order -- with
// MASTER_ADDR:MASTER_PORT; only as the last resort MASTER_PORT + the parent's pid, which requires every rank to be a DIRECT child
// of one launcher process (a per-rank wrapper script or `bash -c` in between breaks that).
std::string gvh_job_key() {
    auto env = [](const char* k) { const char* v = getenv(k); return std::string(v ? v : ""); };
    if (!env("GVAMP_RENDEZVOUS").empty()) return env("GVAMP_RENDEZVOUS");
    const std::string where = env("MASTER_ADDR") + ":" + (env("MASTER_PORT").empty() ? "0" : env("MASTER_PORT"));
    if (!env("TORCHELASTIC_RUN_ID").empty()) return "torch." + env("TORCHELASTIC_RUN_ID") + "." + where;
    if (!env("SLURM_JOB_ID").empty()) return "slurm." + env("SLURM_JOB_ID") + "." + env("SLURM_STEP_ID") + "." + where;
    if (!env("PMIX_NAMESPACE").empty()) return "pmix." + env("PMIX_NAMESPACE") + "." + where;
    if (!env("PMI_JOBID").empty()) return "pmi." + env("PMI_JOBID") + "." + where;
    if (!env("OMPI_MCA_ess_base_jobid").empty()) return "ompi." + env("OMPI_MCA_ess_base_jobid") + "." + where;
    return "ppid." + std::to_string((long)getppid()) + "." + where;
}
// The RCCL unique id of a job travels from rank 0 to the others through a small file (one process per GPU, no MPI to broadcast it).
std::string gvh_id_file_default() {
    if (const char* rdv = getenv("GVAMP_RENDEZVOUS")) return std::string(rdv) + "." + std::string(getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0");
    return "/tmp/gvamp_rccl_id." + std::to_string(std::hash<std::string>{}(gvh_job_key()));
}
int gvh_exchange_id_impl(const std::string& path, int rank, unsigned char* id128, double timeout_s, std::string& err) {
    if (rank == 0) {
        const std::string tmp = path + ".tmp";
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id128, 1, 128, f) != 128) { if (f) fclose(f); err = "cannot write " + tmp; return 1; }
        fclose(f);
        if (rename(tmp.c_str(), path.c_str()) != 0) { err = "cannot rename " + tmp + " to " + path; return 1; }
        return 0;
    }
    // a file left behind by a run that died before rank 0 removed it must not be taken for this run's id: only a file written
    // after this process started (minus a launch skew of 30 s) counts
    const time_t born = time(nullptr) - 30;
    const double t0 = now_s();
    for (;;) {
        struct stat sb;
        if (stat(path.c_str(), &sb) == 0 && sb.st_mtime >= born && sb.st_size == 128) {
            FILE* f = fopen(path.c_str(), "rb");
            if (f) {
                const size_t n = fread(id128, 1, 128, f);
                fclose(f);
                if (n == 128) return 0;
            }
        }
        if (now_s() - t0 > timeout_s) break;
        usleep(10000);
    }
    err = "rank " + std::to_string(rank) + " waited " + std::to_string((int)timeout_s) + " s for the id of rank 0 in " + path + " (job key \"" +
          gvh_job_key() + "\"): if the ranks are not started by one launcher that sets a job id (torchrun, Slurm, mpirun) or as its direct "
          "children, export GVAMP_RENDEZVOUS=<a path private to the job> for all of them";
    return 1;
}
std::string gvh_shm_default_name() { return "/gvamp_" + std::to_string(std::hash<std::string>{}(gvh_job_key())); }

gvh_shm_comm* gvh_shm_open_impl(const std::string& name, int nranks, int rank, size_t cap_doubles, std::string& err) {
    if (nranks < 1 || rank < 0 || rank >= nranks || cap_doubles == 0) { err = "gvh_shm_open: bad rank / nranks / capacity"; return nullptr; }
    const size_t hdr_bytes = (sizeof(Hdr) + 4095) / 4096 * 4096;
    const size_t bytes = hdr_bytes + sizeof(double) * cap_doubles * (size_t)nranks;
    const uint64_t name_hash = (uint64_t)std::hash<std::string>{}(name);
    auto make = [&](void* p) {
        gvh_shm_comm* c = new gvh_shm_comm();
        c->nranks = nranks; c->rank = rank; c->cap = cap_doubles; c->bytes = bytes;
        c->hdr = static_cast<Hdr*>(p);
        c->slots = reinterpret_cast<double*>(static_cast<char*>(p) + hdr_bytes);
        return c;
    };
    if (rank == 0) {
        shm_unlink(name.c_str());                                   // whatever a dead job left under this name
        int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { err = "gvh_shm_open: cannot create " + name + ": " + strerror(errno); if (fd >= 0) close(fd); return nullptr; }
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { err = std::string("gvh_shm_open: mmap failed: ") + strerror(errno); shm_unlink(name.c_str()); return nullptr; }
        gvh_shm_comm* c = make(p);
        Hdr* h = c->hdr;
        h->nranks = (uint32_t)nranks; h->cap = cap_doubles; h->creator_pid = (int32_t)getpid();
        h->creator_start = proc_start_ticks((long)getpid()); h->name_hash = name_hash;
        h->joined.store(0); h->go.store(0); h->bar_count.store(0); h->bar_gen.store(0); h->failed.store(0);
        h->magic.store(MAGIC, std::memory_order_release);
        const bool ok = spin_until([&] { return h->joined.load(std::memory_order_acquire) == (uint32_t)(nranks - 1); }, 120.0, nullptr);
        shm_unlink(name.c_str());                                   // nobody can find it any more; it lives while it is mapped
        if (!ok) { err = "gvh_shm_open: not every rank attached to " + name + " within 120 s"; munmap(p, bytes); delete c; return nullptr; }
        h->go.store(1, std::memory_order_release);
        return c;
    }
    // A joiner may find, under this name, the segment a DEAD job of the same size left before rank 0 of this job has unlinked and
    // re-created it: such a segment (already started, creator gone, or a creator pid that has since been reused by another process)
    // is dropped and the name tried again until this job's segment appears or the deadline passes.
    const double t0 = now_s();
    std::string why = "rank 0 never created it";
    for (;;) {
        if (now_s() - t0 > 120.0) { err = "gvh_shm_open: no usable segment under " + name + " within 120 s (" + why + ")"; return nullptr; }
        int fd = shm_open(name.c_str(), O_RDWR, 0600);
        if (fd < 0) { usleep(2000); continue; }
        struct stat sb;
        if (fstat(fd, &sb) != 0 || (size_t)sb.st_size != bytes) { close(fd); why = "a segment of another size"; usleep(2000); continue; }   // (or: rank 0 has not sized it yet)
        void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { err = std::string("gvh_shm_open: mmap failed: ") + strerror(errno); return nullptr; }
        Hdr* h = static_cast<Hdr*>(p);
        const bool inited = spin_until([&] { return h->magic.load(std::memory_order_acquire) == MAGIC; }, 2.0, nullptr);
        const bool mine = inited && h->nranks == (uint32_t)nranks && h->cap == cap_doubles && h->name_hash == name_hash &&
                          h->go.load(std::memory_order_acquire) == 0 && kill((pid_t)h->creator_pid, 0) == 0 &&
                          proc_start_ticks((long)h->creator_pid) == h->creator_start;
        if (!mine) {
            why = inited ? "a stale or foreign segment" : "a segment that was never initialised";
            munmap(p, bytes);
            usleep(5000);
            continue;
        }
        gvh_shm_comm* c = make(p);
        h->joined.fetch_add(1, std::memory_order_acq_rel);
        if (!spin_until([&] { return h->go.load(std::memory_order_acquire) == 1; }, 120.0, nullptr)) {
            err = "gvh_shm_open: rank 0 never released " + name;
            munmap(p, bytes);
            delete c;
            return nullptr;
        }
        return c;
    }
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
