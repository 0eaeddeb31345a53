// gv_internal.h -- context layout and kernel-launch prototypes shared by the libgvamp translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <memory>
#include <string>
#include <unordered_set>
#include <vector>

#include "gvamp.h"
#include "gv_mfma.h"

struct gv_vec {
    gv_ctx* ctx;
    int space;      // GV_SPACE_M / GV_SPACE_N
    int64_t len;    // logical length (M or 4*mbytes)
    int64_t cap;    // allocated doubles (M or npad); the tail beyond len is kept at 0
    double* d;
    bool owns = true;   // false: d points into another vector's allocation (w_n2 behind w_n)
};

struct gv_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // dataset ------------------------------------------------------------------------------------
    int64_t N = 0, M = 0, Mt = 0, S = 0;
    int64_t mbytes = 0;   // ceil(N/4)
    int64_t pitch = 0;    // bytes per marker row in HBM (multiple of 64)
    int64_t npad = 0;     // 4 * pitch : device length of every N-space vector
    uint8_t* bed = nullptr;     // M * pitch, marker-major 2-bit, pad bytes 0
    uint32_t* mask2 = nullptr;  // pitch/4 words: bits 2q and 2q+1 set iff individual 16j+q has a phenotype
    int64_t nonas = 0;
    double* mave = nullptr;
    double* msig = nullptr;
    uint32_t* counts = nullptr;     // 3*M: present individuals with a = 2, 1, 0 per marker (marker statistics)
    double alpha_scale = 1.0;       // of the last gv_marker_stats
    bool have_stats = false;
    // Defaults = the engine bench.py measures: i8 MFMA kernels on a re-encoded layout picked at ingest, no raw rows resident.
    // The fp64 VALU family (parity anchor) and the raw rows (gv_download_bed) are opt-in: gv_set_kernel_mode(ctx, 0),
    // gv_set_layout(ctx, 1, ..).
    int kernel_mode = 1;            // 0 = fp64 VALU on raw rows, 1 = i8 MFMA on stripes, 2 = the same in two-level fixed point (gvm::ax_wide)
    bool want_raw = false, want_stripes = true;  // layouts built at ingest (gv_set_layout)
    bool want_auto = true;                       // gv_set_layout(.., 3): want_tile is decided at ingest from the free HBM
    bool want_tile = false;                      // the MFMA family's layout: false = two stripe sets, true = one tile layout
    bool have_raw = false, have_stripes = false;
    gvm::Plan plan;
    // candidate work decompositions of the ATx-side / Ax-side streaming kernels (default first) and whether the on-device
    // pick among them has been made (autotune_ks)
    std::vector<gvm::Decomp> dec_cand_m, dec_cand_n;
    bool ks_fixed_m = false, ks_fixed_n = false;   // an override fixed the decomposition: nothing to pick
    bool ks_tuned = false;
    double ingest_alloc_s = 0.0, ingest_fill_s = 0.0;   // last ingest: allocating the resident layouts / filling them
    double ingest_overlap_s = 0.0;  // ... of which the allocation ran beside the preparation of the source (helper thread)
    size_t ingest_bytes = 0;        // bytes of genotype layouts resident after the last ingest
    int64_t expected_passes = 0;    // gv_set_expected_passes: 0 = unknown
    double tune_seconds = 0.0;      // wall time the pick cost (0 when it came from the cache)
    int tune_source = 0;            // 0 model's first candidate, 1 measured, 2 cache, 3 fixed by an override / nothing to tune

    // workspaces ---------------------------------------------------------------------------------
    double* t3 = nullptr;          // 3*M: per-marker Ax table {(2-mu)c, (1-mu)c, (0-mu)c}
    double* ax_partial = nullptr;  // ax_chunks * npad
    int ax_chunks = 0;
    double* red_partial = nullptr; // RED_BLOCKS * RED_MAXK block partials
    double* red_out = nullptr;     // RED_MAXK device scalars
    double* host_pin = nullptr;    // pinned, RED_MAXK doubles
    // scalar mailbox: mapped coherent host memory the device writes into (k_publish), RED_MAXK doubles + a sequence flag
    double* mbox = nullptr;        // host address
    double* mbox_dev = nullptr;    // the same memory as the device sees it
    unsigned long long mbox_seq = 0;
    bool use_mbox = false;
    // a reduction whose finalisation publishes to the mailbox itself (arm_scalars -> gvk::arm_publish -> k_finalize_pub)
    unsigned int* pub_counter = nullptr;
    bool pub_armed = false;
    unsigned long long pub_seq = 0;
    void* xfer_pin = nullptr;      // 8 MiB pinned staging buffer of the whole-vector host transfers (to_host / to_device)
    hipEvent_t xfer_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // to_host: one per 2 MiB piece of the staging buffer
    // device-resident CG (cg_run_device): 2 state blocks of gvm::ST_SIZE doubles, the `go` / rider flags, residual traces
    // (2 x cgx_relcap doubles), a pinned staging block for the initial states
    double* cgx_state = nullptr;
    int* cgx_go = nullptr;
    double* cgx_rel = nullptr;     // (device view of mapped, coherent HOST memory: the decide kernels write a trace entry per step straight
    double* cgx_rel_h = nullptr;   //  into it and the host reads it behind the status it already waits for -- no copy, no event)
    int cgx_relcap = 0;
    // How far the device-resident loops enqueue ahead of their statuses: the steps (cg_run_device) / passes (gv_cg_solve_aat2w) the
    // previous solve of the same kind needed.  From that count on the host reads the status of the step it has just enqueued before
    // it enqueues another -- consecutive VAMP iterations repeat their step counts, and a step enqueued after the last one costs a
    // dozen dropped launches (~60 us) where a wrong guess the other way costs one host round trip (~15 us).  0 = no hint yet.
    // The hint holds for TWO counts (hint, hint + 1): a solve that outgrows it (cg_iters 2 -> 27 -> 31 on LD genotypes) goes back to
    // reading one step behind, so the run-ahead is not lost for the rest of the solve.  One hint per solver kind (LMMSE alone,
    // Onsager alone, the two in lock-step; the joint N-space solver): a --fuse-solves 0 run alternates kinds with different counts.
    // Reset with the data set (free_dataset).
    int spec_hint_steps[3] = {0, 0, 0}, spec_hint_passes = 0;
    void* stripes_slab = nullptr;  // owner of plan.stripes_n | plan.stripes_m when the two stripe sets share one allocation (ingest)
    double* aat_slab = nullptr;    // work vectors of the N-space solvers (gv_solvers.hip: aat_scratch), kept between calls
    size_t aat_slab_cap = 0;
    gv_vec *mave_p = nullptr, *msig_p = nullptr, *numb_p = nullptr;   // people statistics (gv_people_stats), N-space
    gv_vec *w_n = nullptr, *w_n2 = nullptr;                  // N-space scratch (lmmse_mult, two-vector form)
    gv_vec *cg2_r = nullptr, *cg2_z = nullptr, *cg2_p = nullptr, *cg2_d = nullptr;   // second CG system (gv_cg_solve2)
    gv_vec *cg_r = nullptr, *cg_z = nullptr, *cg_p = nullptr, *cg_d = nullptr;  // CG work vectors
    std::unordered_set<gv_vec*> live_vecs;   // every vector of this context (vec_new / vec_del): freed at gv_destroy

    // communicator ---------------------------------------------------------------------------------
    ncclComm_t comm = nullptr;
    std::shared_ptr<void> comm_keep;       // owns comm: ncclCommDestroy when the last context sharing it lets go (gv_comm_share)
    int rank = 0, nranks = 1;
    void* local = nullptr;                 // in-process test communicator (gv_comm_init_local)
    std::shared_ptr<void> local_keep;
    std::vector<double> local_buf;
    int overlap_tiles = 0;                 // > 1: data::Ax in that many individual chunks, exchange on comm_stream (GV_OVERLAP)
    hipStream_t comm_stream = nullptr;
    hipEvent_t ev_chunk = nullptr, ev_comm = nullptr;
    gv_allreduce_fn cb = nullptr;          // host-callback communicator (gv_comm_init_callback)
    void* cb_user = nullptr;
    // gv_debug_force_multi: a ONE-rank context takes every multi-rank branch; the exchange is an in-stream loop-back through
    // scratch (1), the 1-rank RCCL communicator (2) or both (3) -- never a host synchronisation
    int force_multi = 0;
    int loop_delay_us = 0;
    double* loop_buf[2] = {nullptr, nullptr};     // [0] the context's stream, [1] the side stream of the overlapped exchange
    size_t loop_cap[2] = {0, 0};

    // instrumentation ------------------------------------------------------------------------------
    int timing = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    gv_counters cnt{};
    struct EvRec { hipEvent_t a, b; int kind; };
    std::vector<EvRec> ev_pool;   // timing == 2: un-synchronised event pairs around the matvec kernels
    size_t ev_used = 0;
};

constexpr int RED_BLOCKS = 1024;
constexpr int RED_MAXK = 80;   // >= 1 + 2*(LMAX-1) with LMAX = 32
constexpr int GV_LMAX = 32;

struct gv_prior {
    int L;
    double probs[GV_LMAX];
    double vars[GV_LMAX];
};

// ---- kernel launchers (gv_kernels.hip) -----------------------------------------------------------
namespace gvk {
void synth_bed(hipStream_t s, uint8_t* bed, int64_t M, int64_t S, int64_t N, int64_t pitch, uint64_t seed,
               uint32_t miss_thr, uint32_t ld_block = 0, uint32_t ld_thr = 0);
void marker_stats(hipStream_t s, const uint8_t* bed, const uint32_t* mask2, int64_t M, int64_t pitch, double nonas,
                  double alpha_scale, double* mave, double* msig, uint32_t* counts);
void marker_sums2_f64(hipStream_t s, const uint8_t* bed, int64_t M, int64_t pitch, const double* p1, const double* p2,
                      double* out4);
void people_table(hipStream_t s, const double* mave, const double* msig, int64_t M, int kind, double* t3);
void people_finish(hipStream_t s, double* s1, double* s2, const double* cnt, const uint32_t* mask2, int64_t N,
                   int64_t npad);
void aat_diag(hipStream_t s, const double* mave_p, const double* msig_p, const double* numb_p, double tau, double gam2,
              double Nd, int64_t npad, double* diag);
void cg_step_b_diag(hipStream_t s, double* r, const double* d, double alpha, const double* diag, double* z, int64_t n,
                    double* partial, double* out);   // out = <r,z>, <r,r>
void probit_denoise(hipStream_t s, const double* p1, const double* y, const double* m_cov, int64_t N, int64_t npad,
                    double tau1, double probit_var, double* z1, double* partial, double* out);   // out = sum g1d, sum (z1-p1)^2
void mul(hipStream_t s, double* out, const double* x, const double* y, int64_t n);
void select_eq(hipStream_t s, double* out, const double* x, const int* key, int value, int64_t n);
void ax_table(hipStream_t s, const double* x, const double* mave, const double* msig, int64_t M, double* t3);
void ax_f64(hipStream_t s, const uint8_t* bed, int64_t M, int64_t pitch, const double* t3, int chunks,
            double* partial, int64_t npad);
void ax_reduce(hipStream_t s, const double* partial, int chunks, int64_t npad, const uint32_t* mask2, double scale,
               double* out);
void scale_vec(hipStream_t s, double* v, int64_t n, double a);
void atx_f64(hipStream_t s, const uint8_t* bed, int64_t M, int64_t pitch, const double* p, const double* mave,
             const double* msig, double scale, double* out);
void fill(hipStream_t s, double* v, int64_t n, double a);
void fill_hash(hipStream_t s, double* v, int64_t n, uint64_t seed);
void publish(hipStream_t s, const double* src, int K, double* mailbox, unsigned long long* flag, unsigned long long seq);
// the finalisation of the NEXT reduction launched from this thread also publishes its scalars (no k_publish launch)
void arm_publish(double* mailbox, unsigned long long* flag, unsigned long long seq, unsigned int* counter);
void disarm_publish();
void axpby(hipStream_t s, double* out, double a, const double* x, double b, const double* y, int64_t n);
void mask_copy(hipStream_t s, double* out, const double* in, const uint32_t* mask2, int64_t npad);
void copy(hipStream_t s, double* dst, const double* src, int64_t n);                  // dst = src (kernel, not hipMemcpyAsync)
void p_update(hipStream_t s, double* p, const double* z, double beta, int64_t n);     // p = fma(beta, p, z)
// one-rank all-reduce that moves the message through scratch and poisons it meanwhile (gv_debug_force_multi)
void loopback(hipStream_t s, double* buf, double* scratch, int64_t n, int delay_us);
// K dot products <x[k], y[k]> over n elements -> red_out[0..K)
void dots(hipStream_t s, int K, const double* const* x, const double* const* y, int64_t n, double* partial,
          double* out);
void dots_ex(hipStream_t s, int K, const double* const* xa, const double* const* xb, const double* const* ya,
             const double* const* yb, const int64_t* n, double* partial, double* out);   // <xa - xb, ya - yb>, lengths n[k]
// device-resident CG step (gv_solvers.hip: cg_run_device)
// dp_part != NULL (one rank): <d,p> of system k is still in dp_nb block partials at dp_part[k] and the three sums of this
// launch stay block partials too (k_cgx_decide adds them up: cgx_decide's part / part_nb); returns the number of blocks
int cgx_ab(hipStream_t s, int nsys, double* const* st, double* const* mu, const double* const* p, const double* const* v,
           double* const* r, const double* const* d, double* const* z, const double* const* dp, double* const* part,
           double* const* red, double diag, int64_t n, const double* const* dp_part = nullptr, int dp_nb = 0,
           double* const* az = nullptr, const double* const* aw = nullptr, int64_t npad = 0);   // az: A mu += alpha A p rides along
void finalize(hipStream_t s, const double* partial, int nb, int K, double* out);   // ordered sum of block partials
void state_init(hipStream_t s, double* dst, const double* q);   // q: gvm::ST_SIZE doubles, by value in the launch
void set_ints(hipStream_t s, int* dst, int a, int b);
void go_from_states(hipStream_t s, int* dst, const double* st0, const double* st1, int b);
// the state block of a system at its start from the block partials of its two opening reductions over n entries (one rank)
void state_from_partials(hipStream_t s, double* dst, const double* q, const double* part_rz, int K_rz, int k_rz, const double* part_vv,
                         int64_t n, bool sqrt_norm);
void state_from_scalars(hipStream_t s, double* dst, const double* q, const double* rz, const double* vv, bool sqrt_norm);
void cgx_decide(hipStream_t s, int nsys, double* const* st, const double* const* red, double* const* relres, double gam2,
                int max_iter, int* go, double* mailbox, unsigned long long* flag, unsigned long long seq, int* ride,
                const double* other_st = nullptr, const int* ride_report = nullptr, const double* const* part = nullptr,
                int part_nb = 0);
void ride_mark(hipStream_t s, const double* st0, const double* st1, int* ride);
void p_update_st(hipStream_t s, double* p, const double* z, const double* st, int64_t n);   // p = z + beta p while st is stepping
void aat_step(hipStream_t s, double* st, double* mu, double* p, double* r, double* d, double* z, const double* diag, double tau,
              double gam2, int64_t n, double* partial, double* relres, int max_iter, double* mailbox,
              unsigned long long* flag, unsigned long long seq, double* at_acc = nullptr,
              const double* at_p = nullptr, int64_t m = 0, const double* other_st = nullptr, int* go = nullptr,
              const int* ride = nullptr, int* ride_mark = nullptr, bool p_update = true, bool dq_done = false);
void ride_copy(hipStream_t s, double* out, const double* w0, const double* w1, const double* st0, const double* st1,
               const int* ride, int64_t n);
void cg_step_a(hipStream_t s, double* mu, const double* p, double alpha, const double* v, int64_t n,
               double* partial, double* out);                 // mu += alpha p ; out[0] = <v, mu>
void cg_step_b(hipStream_t s, double* r, const double* d, double alpha, double diag, double* z, const double* mu,
               int64_t n, double* partial, double* out);      // r -= alpha d ; z = r/diag ; out = <r,z>,<z,z>,<r,r>,<mu,mu>
void denoise(hipStream_t s, const double* r1, int64_t n, double gam1, const gv_prior& pr, double* x1, double* dd,
             double* partial, double* out);                   // out[0] = sum g1d, out[1] = sum (x1-r1)^2
void prior_estep(hipStream_t s, const double* r1, int64_t n, double gam1, double lambda, const gv_prior& om_vars,
                 double* partial, double* out);               // out[0..1+2(L-1))
void pvals_test(hipStream_t s, const uint32_t* cnt, const double* mave, const double* msig, const double* sums4,
                const double* xself, double self_scale, const int* chrom, int ch, int64_t M, double* pvals);
void copy_bw(hipStream_t s, const double* src, double* dst, int64_t n);
void read_bw(hipStream_t s, const void* src, int64_t blocks_per_wave, int64_t nwaves, unsigned int* sink, int perm = 0);
}  // namespace gvk

// ---- internals shared by the translation units of the C ABI (gv_capi.hip, gv_solvers.hip) ------------------------------
namespace gvi {
int fail(gv_ctx* c, const char* fmt, ...);
int vec_new(gv_ctx* c, int space, gv_vec** out);
void vec_del(gv_ctx* c, gv_vec* v);
int ensure_w2(gv_ctx* c);       // the second N-space scratch vector (behind w_n in one allocation)
int ensure_work(gv_ctx* c);     // scratch vectors of the matvecs and the CG
int read_scalars(gv_ctx* c, int K, double* out);                            // red_out[0..K) -> host (mailbox or copy)
// Call right before launching a reduction (gvk::dots, cg_step_*, denoise, ...) whose red_out[0..K) the very next
// read_scalars(c, K, .) fetches, with nothing all-reduced in between: saves the k_publish launch of that read-back.
void arm_scalars(gv_ctx* c);
int read_scalars_global(gv_ctx* c, int K, double* out, bool multi);         // the same, summed over the ranks first
int allreduce_scalars(gv_ctx* c, double* buf, int K);                       // MPI_Allreduce(SUM) of K host scalars
bool is_multi(const gv_ctx* c);
int comm_allreduce(gv_ctx* c, double* dev, size_t n);
int comm_allreduce_on(gv_ctx* c, double* dev, size_t n, hipStream_t stream);
int to_host(gv_ctx* c, void* dst, const void* src_dev, size_t nbytes);
int to_device(gv_ctx* c, void* dst_dev, const void* src, size_t nbytes, bool sync = true);
// data::Ax / data::ATx (and their two-vector forms) on device pointers, in the kernel family of the context
// (cg: the slots of the pass that belong to a CG system with device-resident scalars -- gvm::CgHook)
int ax_device(gv_ctx* c, const double* x, double* out, const gvm::CgHook* cg = nullptr);
int atx_device(gv_ctx* c, const double* p, double* out, const double* addx = nullptr, double tau = 1.0, double gam2 = 0.0,
               const gvm::CgHook* cg = nullptr);
int ax2_device(gv_ctx* c, const double* xa, const double* xb, double* outa, double* outb, const gvm::CgHook* cg = nullptr);
int atx2_device(gv_ctx* c, const double* pa, const double* pb, double* outa, double* outb, const double* addxa = nullptr,
                const double* addxb = nullptr, double tau = 1.0, double gam2 = 0.0, const gvm::CgHook* cg = nullptr);
int lmmse_device(gv_ctx* c, const double* v, double tau, double gam2, double* out);
bool use_overlap(const gv_ctx* c);    // data::Ax cut into chunks whose exchange runs on the side stream (GV_OVERLAP)
int ax_overlapped(gv_ctx* c, int nv, const double* xa, const double* xb, double* outa, double* outb, const gvm::CgHook* cg);
int autotune_ks(gv_ctx* c);     // picks the work decompositions of the streaming kernels (once per shard)

struct Timer {
    gv_ctx* c;
    double* acc;
    bool on;
    Timer(gv_ctx* c_, double* acc_) : c(c_), acc(acc_), on(c_->timing == 1) {
        if (on) (void)hipEventRecord(c->ev0, c->stream);
    }
    void stop() {
        if (!on) return;
        (void)hipEventRecord(c->ev1, c->stream);
        (void)hipEventSynchronize(c->ev1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, c->ev0, c->ev1);
        *acc += ms;
        on = false;
    }
};
}  // namespace gvi

#define HIPCHK(c, call)                                                                      \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) return fail(c, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define NCCLCHK(c, call)                                                                     \
    do {                                                                                     \
        ncclResult_t r_ = (call);                                                            \
        if (r_ != ncclSuccess) return fail(c, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)
#define KCHK(c) HIPCHK(c, hipGetLastError())
#define NEED(c, cond, msg)                 \
    do {                                   \
        if (!(cond)) return fail(c, msg);  \
    } while (0)

