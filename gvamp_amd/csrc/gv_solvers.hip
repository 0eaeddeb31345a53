// gv_solvers.hip -- the solvers of the C ABI (include/gvamp.h): vamp::precondCG_solver as a state machine that can advance two
// systems on shared passes over the genotype shard, and the N-space solvers of --use-XXT-denoiser 1.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "gv_internal.h"

using namespace gvi;

extern "C" {

// ---- solver ---------------------------------------------------------------------------------------------
int gv_lmmse_mult(gv_ctx* c, const gv_vec* v, double tau, double gam2, gv_vec* out) {
    NEED(c, v->space == GV_SPACE_M && out->space == GV_SPACE_M && v != out, "gv_lmmse_mult: M-space, no aliasing");
    return lmmse_device(c, v->d, tau, gam2, out->d);
}

// vamp::precondCG_solver (vamp.cpp:1130-1229) as a small state machine, so that TWO solves on the same operator
// (the LMMSE solve and the Onsager probe solve of one VAMP iteration, vamp.cpp:593-596 / :884) can share every pass
// over the genotype shard: each round gathers the pending operator applications Q p of the active systems and runs
// them as one two-vector Ax / ATx pair.  Per system the recurrences, stopping rules and results are exactly those of
// a stand-alone solve.  Deviations from the reference, all result-neutral: <r,z> and ||v|| are
// computed once per step instead of three times / every step; lmmse_mult's all-zero shortcut (:1079) is replaced by
// "mu_start == NULL"; the scalar reductions of one step travel in packed all-reduces.
struct CgSys {
    const double* v = nullptr;
    const double* mu0 = nullptr;   // warm start or NULL
    double *mu = nullptr, *r = nullptr, *z = nullptr, *p = nullptr, *d = nullptr;
    int denoiser = 1;
    double* relres = nullptr;
    // state
    bool active = true;
    int phase = 1;                 // 0: waiting for Q mu0 (initial residual), 1: waiting for Q p
    double rz = 0, norm_v = 0, prev_onsager = 0, onsager = 0, rel_err = 0;
    int iters = 0, n_relres = 0, converged = 0;
    const double* req = nullptr;   // operand of the pending operator application
    double* res = nullptr;         // where its result goes
    // by-products (gv_cg_solve2x)
    double* az = nullptr;          // N-space: A mu, accumulated from the A p_k of the operator applications
    const double* wslot = nullptr; // N-space buffer holding A req after the last application
    bool keep_resid = false;       // keep r = v - Q mu exact on the Onsager-rule exit too
    // warm start whose products are already known (gv_cg_solve2w): A^T A mu0 and A mu0 -- no pass for the initial residual
    const double* ata0 = nullptr;
    const double* amu0 = nullptr;
    // A^T A v of a zero-started solve whose right-hand side does not change between calls (the Onsager probe): input that
    // replaces the first operator application (ata_v_known), else captured from it
    double* ata_v = nullptr;
    bool ata_v_known = false;
};

// The first CG step of a zero-started solve applies the operator to p0 = z0 = v / diag.  With A^T A v at hand,
// d = Q p0 = (tau / diag) A^T A v + gam2 p0 needs no pass; the step itself is the host-driven one (same arithmetic as on the device).
static int cg_consume_all(gv_ctx* c, CgSys** act, int na, double gam2, double diag, int max_iter, bool multi);
static int cg_first_step_from_known_product(gv_ctx* c, CgSys& s, double tau, double gam2, double diag, int max_iter, bool multi) {
    if (!(s.ata_v && s.ata_v_known && s.active && s.phase == 1 && s.iters == 0 && !s.mu0 && max_iter > 0)) return 0;
    gvk::axpby(c->stream, s.d, tau / diag, s.ata_v, gam2, s.p, c->M);
    KCHK(c);
    CgSys* one[1] = {&s};
    return cg_consume_all(c, one, 1, gam2, diag, max_iter, multi);
}
// ... and where the product is not at hand yet, the first application delivers it: d = tau A^T A (v / diag) + gam2 v / diag
static void cg_capture_first_product(gv_ctx* c, const CgSys& s, double tau, double gam2, double diag) {
    if (!(s.ata_v && !s.ata_v_known && tau != 0.0)) return;
    gvk::axpby(c->stream, s.ata_v, diag / tau, s.d, -gam2 / tau, s.v, c->M);
}

static int cg_finish_init(gv_ctx* c, CgSys& s, double diag, bool multi) {
    // z = r / diag (:1152), <r,z>, ||v||^2 ; p = z (:1154)
    const int64_t M = c->M;
    double sc[5];
    gvk::cg_step_b(c->stream, s.r, s.d, 0.0, diag, s.z, s.mu, M, c->red_partial, c->red_out);          // red_out[0..3]
    const double* vv[1] = {s.v};
    gvk::dots(c->stream, 1, vv, vv, M, c->red_partial + 4 * RED_BLOCKS, c->red_out + 4);               // red_out[4]
    KCHK(c);
    if (read_scalars(c, 5, sc)) return 1;            // both reductions in ONE read-back
    double pk[2] = {sc[0], sc[4]};
    if (multi && allreduce_scalars(c, pk, 2)) return 1;
    s.rz = pk[0];
    s.norm_v = sqrt(pk[1]);
    gvk::copy(c->stream, s.p, s.z, M);
    s.phase = 1;
    s.req = s.p;
    s.res = s.d;
    return 0;
}

// One CG step (:1160-1223) of every system whose operator application has just come back.  The systems move through
// the three reductions of a step TOGETHER: their scalars sit in disjoint slots of red_out (8 doubles per system), so a
// step costs three all-reduce + read-back round trips however many systems are active (sharded jobs: the latency of
// these small messages is what remains of a CG step besides the two passes over the shard).
static int cg_consume_all(gv_ctx* c, CgSys** act, int na, double gam2, double diag, int max_iter, bool multi) {
    const int64_t M = c->M;
    hipStream_t st = c->stream;
    CgSys* stp[2];
    int slot[2];
    int ns = 0;
    for (int k = 0; k < na; k++) {
        CgSys& s = *act[k];
        if (s.phase == 0) {                                               // r = v - Q mu0 (:1142-1145)
            gvk::axpby(st, s.r, 1.0, s.v, -1.0, s.r, M);
            if (s.az) gvk::copy(st, s.az, s.wslot, c->npad);   // A mu0
            KCHK(c);
            if (cg_finish_init(c, s, diag, multi)) return 1;
            if (max_iter <= 0) s.active = false;
        } else {
            slot[ns] = k;
            stp[ns++] = &s;
        }
    }
    if (ns == 0) return 0;
    const int K = 8 * (slot[ns - 1] + 1);
    double sc[16], alpha[2] = {0, 0};
    auto part = [&](int j) { return c->red_partial + (size_t)slot[j] * RED_BLOCKS * 8; };
    auto outp = [&](int j) { return c->red_out + 8 * slot[j]; };
    // (one system in slot 0 of an unsharded job: each reduction's finalisation publishes its scalars itself)
    const bool solo = ns == 1 && slot[0] == 0 && !multi;
    // <d, p>
    for (int j = 0; j < ns; j++) {
        const double* xs[1] = {stp[j]->d};
        const double* ys[1] = {stp[j]->p};
        if (solo) arm_scalars(c);
        gvk::dots(st, 1, xs, ys, M, part(j), outp(j));
    }
    KCHK(c);
    if (read_scalars_global(c, K, sc, multi)) return 1;
    bool any_onsager = false;
    for (int j = 0; j < ns; j++) {
        CgSys& s = *stp[j];
        alpha[j] = s.rz / sc[8 * slot[j]];                                // :1167
        if (solo && s.denoiser == 0) arm_scalars(c);                      // read back only under the Onsager rule
        gvk::cg_step_a(st, s.mu, s.p, alpha[j], s.v, M, part(j), outp(j));   // mu += alpha p (:1169-1172)
        if (s.az) gvk::axpby(st, s.az, 1.0, s.az, alpha[j], s.wslot, c->npad);   // A mu += alpha A p
        any_onsager |= s.denoiser == 0;
    }
    KCHK(c);
    bool stepping[2] = {true, true};
    if (any_onsager) {                                                    // :1174-1193
        if (read_scalars_global(c, K, sc, multi)) return 1;
        for (int j = 0; j < ns; j++) {
            CgSys& s = *stp[j];
            s.iters++;
            if (s.denoiser != 0) continue;
            s.onsager = gam2 * sc[8 * slot[j]];
            const double oerr = (s.onsager != 0) ? fabs((s.onsager - s.prev_onsager) / s.onsager) : 1.0;
            if (oerr < 1e-8) {
                if (s.keep_resid) gvk::axpby(st, s.r, 1.0, s.r, -alpha[j], s.d, M);   // the reference leaves r one step behind
                s.converged = 1;
                s.active = false;
                stepping[j] = false;
            }
            s.prev_onsager = s.onsager;
        }
    } else
        for (int j = 0; j < ns; j++) stp[j]->iters++;
    bool any = false;
    for (int j = 0; j < ns; j++) {
        if (!stepping[j]) continue;
        CgSys& s = *stp[j];
        if (solo) arm_scalars(c);
        gvk::cg_step_b(st, s.r, s.d, alpha[j], diag, s.z, s.mu, M, part(j), outp(j));   // :1195-1216
        any = true;
    }
    KCHK(c);
    if (!any) return 0;
    if (read_scalars_global(c, K, sc, multi)) return 1;
    for (int j = 0; j < ns; j++) {
        if (!stepping[j]) continue;
        CgSys& s = *stp[j];
        const double* q = sc + 8 * slot[j];
        const double beta = q[0] / s.rz;                                  // (1/<r,z>_old) * <r,z>_new (:1198,:1207)
        s.rz = q[0];
        gvk::p_update(st, s.p, s.z, beta, M);                             // p = z + beta p (:1209-1210)
        s.rel_err = sqrt(q[2]) / s.norm_v;                                // :1215
        if (s.relres) s.relres[s.iters - 1] = s.rel_err;
        s.n_relres = s.iters;
        if (s.rel_err < 1e-5) {                                           // :1217,:1222
            s.converged = 1;
            s.active = false;
        } else if (s.iters >= max_iter)
            s.active = false;
    }
    KCHK(c);
    return 0;
}

// two-vector operator: outa = tau A^T A xa + gam2 xa, outb likewise, ONE Ax pass and ONE ATx pass (kernel mode 1)
static int lmmse2_device(gv_ctx* c, const double* xa, const double* xb, double tau, double gam2, double* outa,
                         double* outb) {
    if (ensure_work(c)) return 1;
    if (ensure_w2(c)) return 1;
    if (ax2_device(c, xa, xb, c->w_n->d, c->w_n2->d)) return 1;
    return atx2_device(c, c->w_n->d, c->w_n2->d, outa, outb, xa, xb, tau, gam2);
}

// ---- the steady state of cg_run with every scalar on the device (kernel mode 1) ------------------------------------------
// A CG step of the host-driven loop above costs three scalar round trips (alpha, the Onsager rule, beta / the residual
// rule): the host waits for a reduction, computes one division and launches the next small kernel -- ~15 us of idle GPU each
// through the mailbox on one GPU (1-2 % of a step at 12.5-100 GB shards: docs/history/rounds1-3.md section 5 has the A/B, which is a tie
// there), plus a small-message all-reduce latency each on a sharded job, which is what this loop is for.  Here the
// state of each system (gvm::ST_*) lives in HBM; the step is
//   Ax pass   [k_prep_ax: p <- z + beta p on the way in | k_quant | stream | k_fin_ax | all-reduce]
//   ATx pass  [k_prep_atx | k_quant | stream | k_fin_atx_dot: d = tau A^T A p + gam2 p and the partials of <d,p> | k_finalize]
//   k_cgx_ab  [alpha = <r,z>/<d,p>; mu += alpha p; r -= alpha d; z = r/diag; <v,mu>, <r,z>, <r,r>]
//   k_cgx_decide [Onsager rule, beta, residual rule, trace; go flag; status -> host mailbox]
// and the host only ENQUEUES: it reads the status of step s - 1 after it has enqueued step s, so the device always has
// a step queued.  A step enqueued after every system had finished is dropped on the device (the stream kernels return on
// *go == 0, the update kernels on the system's ACTIVE flag): one step of empty launches, ~50 us.  Sharded jobs: the
// reductions are all-reduced in-stream on the device buffer (RCCL), still without the host.  Per system the arithmetic is
// that of the host-driven loop -- the same reductions in the same order, the same IEEE divisions and square roots.
struct CgxStatus { double active, iters, conv, rel, ons, nrel, stepped, ride; };

static int cgx_alloc(gv_ctx* c, int max_iter) {
    if (!c->cgx_state) {
        HIPCHK(c, hipMalloc(&c->cgx_state, sizeof(double) * 2 * gvm::ST_SIZE));
        HIPCHK(c, hipMalloc(&c->cgx_go, sizeof(int) * 4));
    }
    if (max_iter > c->cgx_relcap) {
        // residual traces: mapped, coherent host memory like the mailbox.  A decide kernel stores its entry before it releases the
        // status flag the host waits for anyway, so the trace is in host memory when the solve returns -- the two device-to-host
        // copies a solve used to end with cost the host ~20 us each (profiles/r4_cfg5_gaps.txt: the gaps around copyBuffer)
        if (c->cgx_rel_h) { HIPCHK(c, hipStreamSynchronize(c->stream)); (void)hipHostFree(c->cgx_rel_h); c->cgx_rel = c->cgx_rel_h = nullptr; c->cgx_relcap = 0; }
        const int cap = max_iter < 64 ? 64 : max_iter;
        void *hp = nullptr, *dp = nullptr;
        HIPCHK(c, hipHostMalloc(&hp, sizeof(double) * 2 * cap, hipHostMallocMapped | hipHostMallocCoherent));
        if (hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) { (void)hipHostFree(hp); return fail(c, "cgx_alloc: no device view of the trace buffer"); }
        memset(hp, 0, sizeof(double) * 2 * cap);
        c->cgx_rel_h = static_cast<double*>(hp);
        c->cgx_rel = static_cast<double*>(dp);
        c->cgx_relcap = cap;
    }
    return 0;
}

// waits until the device has published status number `seq` (or a later one); copies the slot of `seq`
static int cgx_wait(gv_ctx* c, unsigned long long seq, CgxStatus* out2) {
    unsigned long long* flag = reinterpret_cast<unsigned long long*>(c->mbox + RED_MAXK);
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long spins = 0;
    while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) < seq) {
        if ((++spins & 0xFFFF) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 30.0) {
            HIPCHK(c, hipStreamSynchronize(c->stream));       // a kernel may have faulted: let the runtime say so
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) >= seq) break;
            return fail(c, "cg_run_device: the device never published the status of a CG step");
        }
    }
    const double* mb = c->mbox + (seq & 1ull) * 16;
    for (int v = 0; v < 2; v++) {
        const double* q = mb + 8 * v;
        out2[v] = CgxStatus{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7]};
    }
    return 0;
}

// sys[k] (k < nsys <= 2) are initialised (phase 1: r, z, p = z, rz, norm_v set; req = p, res = d).  Runs them to their
// stopping rules.  *ride_done tells whether the rider's product was taken along.
//
// The rider (ride_x -> ride_out = A ride_x, the z1 of --fuse-solves 2) wants the free slot of a two-vector pass in the first
// step after ONE of the two systems has finished.  The host learns that a step late, so the slot is re-assigned on the
// device: while the rider is pending every step is enqueued two-vector shaped (slot v = system v); k_prep_ax feeds ride_x
// into the slot of a system that has finished (if the other one is still running), k_ride_copy moves the product out and
// k_cgx_decide marks the rider done (status word 7) -- the same step in which the host-driven loop would have placed it.
// state_on_device: the state blocks were built on the device by the opening (cg_open_device): nothing to upload, and the residual
// traces start at entry 0 on the device.
static int cg_run_device(gv_ctx* c, CgSys* sys, int nsys, double tau, double gam2, int max_iter, const double* ride_x,
                         double* ride_out, bool* ride_done, bool state_on_device = false) {
    const int64_t M = c->M, npad = c->npad;
    hipStream_t st = c->stream;
    const bool multi = is_multi(c);
    const double diag = tau * (double)(c->N - 1) / (double)c->N + gam2;   // :1137-1138
    const double scale = 1.0 / sqrt((double)c->N);
    const bool ovl = use_overlap(c);
    if (cgx_alloc(c, max_iter)) return 1;
    if (ensure_work(c) || ensure_w2(c)) return 1;
    if (M > 0 && !c->ks_tuned && autotune_ks(c)) return 1;
    c->plan.ev0 = c->plan.ev1 = nullptr;       // no per-launch events inside the loop
    const bool riding = ride_x != nullptr && nsys == 2;
    const int hint_kind = nsys == 2 ? 2 : (sys[0].denoiser == 0 ? 1 : 0);      // gv_ctx::spec_hint_steps
    // ---- initial states -> device
    int go0 = 0;
    for (int k = 0; k < 2; k++) {
        double q[gvm::ST_SIZE];
        for (int i = 0; i < gvm::ST_SIZE; i++) q[i] = 0.0;
        if (k < nsys) {
            const CgSys& s = sys[k];
            q[gvm::ST_RZ] = s.rz; q[gvm::ST_NORMV] = s.norm_v; q[gvm::ST_PREV_ONS] = s.prev_onsager; q[gvm::ST_ONS] = s.onsager;
            q[gvm::ST_RELERR] = s.rel_err; q[gvm::ST_ACTIVE] = s.active ? 1.0 : 0.0; q[gvm::ST_ITERS] = s.iters;
            q[gvm::ST_CONV] = s.converged; q[gvm::ST_NRELRES] = s.n_relres; q[gvm::ST_DENOISER] = s.denoiser;
            go0 |= s.active ? 1 : 0;
        }
        if (!state_on_device || k >= nsys)
            gvk::state_init(st, c->cgx_state + k * gvm::ST_SIZE, q);    // (by value in the launch: no pinned staging to protect)
    }
    // rider word: 1 pending, 2 rode.  The go word: after a device-side opening the host has read no status and believes every system
    // active; the opening's first step may have ended them all (max_iter == 1, rel < 1e-5 in step 1) -- the state blocks know
    if (state_on_device) gvk::go_from_states(st, c->cgx_go, c->cgx_state, c->cgx_state + gvm::ST_SIZE, riding ? 1 : 0);
    else gvk::set_ints(st, c->cgx_go, go0, riding ? 1 : 0);
    KCHK(c);
    double* dst[2] = {c->cgx_state, c->cgx_state + gvm::ST_SIZE};
    double* drel[2] = {c->cgx_rel, c->cgx_rel + c->cgx_relcap};
    int* d_ride = c->cgx_go + 1;
    unsigned long long* flag_dev = reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK);
    bool host_active[2] = {nsys > 0 && sys[0].active, nsys > 1 && sys[1].active};
    const int done_iters[2] = {nsys > 0 ? sys[0].iters : 0, nsys > 1 ? sys[1].iters : 0};
    CgxStatus last[2] = {};
    bool have_last = false, ride_pending = riding, rode_seen = false;
    std::vector<unsigned long long> steps;      // status sequence number of every enqueued step
    auto apply_status = [&](const CgxStatus* s2) {
        for (int k = 0; k < nsys; k++) host_active[k] = s2[k].active != 0.0;
        last[0] = s2[0]; last[1] = s2[1];
        have_last = true;
        if (s2[0].ride == 2.0) { ride_pending = false; rode_seen = true; }
    };
    for (;;) {
        int act[2], na = 0;
        for (int k = 0; k < nsys; k++) if (host_active[k]) act[na++] = k;
        if (na == 0) break;
        // ---- enqueue one step: the systems the host believes active; two-vector shaped (slot v = system v) while both run
        // or while the rider still waits for a slot
        const bool two = na == 2 || ride_pending;
        if (two) { act[0] = 0; act[1] = 1; }
        const int ns = two ? 2 : 1;
        gvm::CgHook hk;
        hk.go = c->cgx_go;
        // one rank: the epilogue of the Ax pass delivers the rider's product itself (no k_ride_copy); sharded or overlapped: it is
        // copied out of the slot behind the cross-rank sum and the scaling of that slot's output
        const bool ride_direct = ride_pending && !multi && !ovl;
        if (ride_pending) { hk.ride = d_ride; hk.alt_x = ride_x; if (ride_direct) hk.ride_out = ride_out; }
        for (int j = 0; j < ns; j++) {
            CgSys& s = sys[act[j]];
            hk.state[j] = dst[act[j]];
            hk.p[j] = s.p;
            hk.z[j] = s.z;
            hk.dot_part[j] = c->red_partial + (size_t)act[j] * RED_BLOCKS * 8;
            hk.dot_out[j] = c->red_out + 8 * act[j];
        }
        // one rank: no all-reduce between a reduction and its consumer -- the consumer adds the block partials up itself and the
        // four k_finalize launches of a step go (gvk::cgx_ab / cgx_decide)
        const bool self = !multi && M > 0;
        hk.dot_self = self;
        double* wn[2] = {c->w_n->d, c->w_n2->d};
        if (two) {
            if (ovl) {             // chunks of individuals, each slice exchanged on the side stream behind the next chunk
                if (ax_overlapped(c, 2, sys[0].p, sys[1].p, wn[0], wn[1], &hk)) return 1;
            } else if (M == 0) {   // empty shard: zeros into the same collectives (the search directions are empty vectors)
                gvk::fill(st, wn[0], 2 * npad, 0.0);
            } else
                gvm::ax2(st, c->plan, sys[0].p, sys[1].p, c->mave, c->msig, c->mask2, npad, multi ? 1.0 : scale, c->red_partial, wn[0], wn[1], &hk);
            KCHK(c);
            if (multi && !ovl) {
                if (comm_allreduce(c, wn[0], 2 * npad)) return 1;      // w_n | w_n2: one message
                gvk::scale_vec(st, wn[0], 2 * npad, scale);
            }
            if (ride_pending && !ride_direct) gvk::ride_copy(st, ride_out, wn[0], wn[1], dst[0], dst[1], d_ride, npad);
        } else {
            if (ovl) {
                if (ax_overlapped(c, 1, sys[act[0]].p, nullptr, wn[0], nullptr, &hk)) return 1;
            } else if (M == 0) gvk::fill(st, wn[0], npad, 0.0);
            else gvm::ax(st, c->plan, sys[act[0]].p, c->mave, c->msig, c->mask2, npad, multi ? 1.0 : scale, c->red_partial, wn[0], &hk);
            KCHK(c);
            if (multi && !ovl) {
                if (comm_allreduce(c, wn[0], npad)) return 1;
                gvk::scale_vec(st, wn[0], npad, scale);
            }
        }
        if (M == 0)            // empty shard: no markers, <d,p> = 0 from this rank
            gvk::fill(st, c->red_out, 8 * (act[ns - 1] + 1), 0.0);
        else if (two)
            gvm::atx2(st, c->plan, wn[0], wn[1], npad, c->mave, c->msig, scale, c->red_partial, sys[0].d, sys[1].d,
                      sys[0].p, sys[1].p, tau, gam2, &hk);
        else
            gvm::atx(st, c->plan, wn[0], npad, c->mave, c->msig, scale, c->red_partial, sys[act[0]].d, sys[act[0]].p, tau, gam2, &hk);
        if (steps.empty())     // the first step of the loop: a zero-started system that has not stepped yet applies Q to v / diag
            for (int j = 0; j < ns; j++)
                if (done_iters[act[j]] == 0 && !sys[act[j]].mu0) cg_capture_first_product(c, sys[act[j]], tau, gam2, diag);
        KCHK(c);
        const int K = 8 * (act[ns - 1] + 1);
        if (multi && comm_allreduce(c, c->red_out, K)) return 1;               // <d,p>
        {
            double *a_st[2], *a_mu[2], *a_r[2], *a_z[2], *a_part[2], *a_red[2];
            const double *a_p[2], *a_v[2], *a_d[2], *a_dp[2], *a_dpp[2];
            for (int j = 0; j < ns; j++) {
                CgSys& s = sys[act[j]];
                a_st[j] = dst[act[j]]; a_mu[j] = s.mu; a_p[j] = s.p; a_v[j] = s.v; a_r[j] = s.r; a_d[j] = s.d; a_z[j] = s.z;
                a_dp[j] = c->red_out + 8 * act[j];
                a_dpp[j] = hk.dot_part[j];
                a_part[j] = c->red_partial + (size_t)act[j] * RED_BLOCKS * 8 + 4 * RED_BLOCKS;   // (clear of the <d,p> partials)
                a_red[j] = c->red_out + 8 * act[j];
            }
            double* a_az[2] = {nullptr, nullptr};
            const double* a_aw[2] = {nullptr, nullptr};
            for (int j = 0; j < ns; j++) { a_az[j] = sys[act[j]].az; a_aw[j] = wn[j]; }       // A mu += alpha A p rides in the same launch
            const int nb_ab = gvk::cgx_ab(st, ns, a_st, a_mu, a_p, a_v, a_r, a_d, a_z, a_dp, a_part, a_red, diag, M,
                                          self ? a_dpp : nullptr, self ? gvm::atx_dot_blocks(c->plan) : 0, a_az, a_aw, npad);
            KCHK(c);
            if (multi && comm_allreduce(c, c->red_out, K)) return 1;           // <v,mu>, <r,z>, <r,r>
            const double* c_red[2] = {c->red_out, c->red_out + 8};
            double* c_rel[2] = {sys[0].relres ? drel[0] : nullptr, (nsys > 1 && sys[1].relres) ? drel[1] : nullptr};
            const unsigned long long seq = ++c->mbox_seq;
            const double* c_part[2] = {c->red_partial + 4 * RED_BLOCKS, c->red_partial + (size_t)RED_BLOCKS * 8 + 4 * RED_BLOCKS};
            gvk::cgx_decide(st, nsys, dst, c_red, c_rel, gam2, max_iter, c->cgx_go, c->mbox_dev, flag_dev, seq,
                            (two && ride_pending) ? d_ride : nullptr, nullptr, nullptr, self ? c_part : nullptr, nb_ab);
            KCHK(c);
            steps.push_back(seq);
        }
        // ---- the status of the step BEFORE the one just enqueued -- or, once the solve has made as many steps as the previous one
        // needed (gv_ctx::spec_hint_steps), of the step just enqueued: it is probably the last
        const int hint = c->spec_hint_steps[hint_kind];
        const bool likely_last = hint > 0 && ((int)steps.size() == hint || (int)steps.size() == hint + 1);
        if (likely_last || steps.size() >= 2) {
            CgxStatus s2[2];
            if (cgx_wait(c, likely_last ? steps.back() : steps[steps.size() - 2], s2)) return 1;
            apply_status(s2);
        }
    }
    // ---- drain: the last enqueued step's status is the final one (a dropped step republishes the unchanged state)
    if (!steps.empty()) {
        CgxStatus s2[2];
        if (cgx_wait(c, steps.back(), s2)) return 1;
        apply_status(s2);
    }
    bool rode = false;
    if (have_last) {
        int executed = 0;                       // steps that ran with *go == 1 = the largest iteration count reached in here
        for (int k = 0; k < nsys; k++) {
            const int it = (int)last[k].iters - done_iters[k];
            if (it > executed) executed = it;
            c->cnt.n_ax += it;
            c->cnt.n_atx += it;
        }
        rode = riding && rode_seen;
        if (rode) c->cnt.n_ax += 1;
        c->cnt.n_ax_pass += executed;
        c->cnt.n_atx_pass += executed;
        c->spec_hint_steps[hint_kind] = executed;
        for (int k = 0; k < nsys; k++) {
            CgSys& s = sys[k];
            s.active = false;
            s.iters = (int)last[k].iters;
            s.converged = (int)last[k].conv;
            s.rel_err = last[k].rel;
            s.onsager = last[k].ons;
            s.n_relres = (int)last[k].nrel;
            // the device trace is indexed by the iteration number of the solve; entries below done_iters[k] were written by
            // the host-driven rounds before the hand-over
            const int from = state_on_device ? 0 : done_iters[k];
            if (s.relres && s.n_relres > from)      // (in host memory already: written ahead of the status just read)
                memcpy(s.relres + from, c->cgx_rel_h + (size_t)k * c->cgx_relcap + from, sizeof(double) * (size_t)(s.n_relres - from));
        }
    }
    if (ride_done) *ride_done = rode;
    return 0;
}

// ---- device-resident scalars, one read-back per operator application: the solvers of --use-XXT-denoiser 1 -----------------
// (gv_cg_solve_aat, gv_cg_solve_aat2).  Their passes alternate between two systems half an application out of phase, so the
// host keeps scheduling pass by pass; but a CG step no longer costs two or three scalar round trips and a dozen
// host-paced launches: the step's kernels (alpha, updates, beta, stopping rule) run back to back on the device and the host
// reads ONE status word set per completed application.
// GV_CG_DEVICE=0 selects the host-driven loops (scalars read back three times per step): the form kernel mode 0 and a runtime
// without a mapped mailbox run anyway, kept switchable as the reference point of tests/test_gpu_cgdevice.py and for the
// host-synchronising transports (gv_comm_init_callback / _local), whose collectives wait for the host whichever loop issues them.
static bool cgx_usable(const gv_ctx* c) {
    const char* cgdev = getenv("GV_CG_DEVICE");
    // (nothing rank-local in here: an empty shard, M == 0, must take the same sequence of collectives as its peers)
    return c->kernel_mode == 1 && c->have_stripes && c->use_mbox && !(cgdev && atoi(cgdev) == 0);
}
// initial state of one system -> device block blk (0 / 1); normv: ||v|| (M-space systems) or ||v||^2 (CG_solverAAT)
static int cgx_upload_state(gv_ctx* c, int blk, double rz, double normv, int denoiser, bool active) {
    double q[gvm::ST_SIZE];
    for (int i = 0; i < gvm::ST_SIZE; i++) q[i] = 0.0;
    q[gvm::ST_RZ] = rz; q[gvm::ST_NORMV] = normv; q[gvm::ST_ACTIVE] = active ? 1.0 : 0.0; q[gvm::ST_DENOISER] = denoiser;
    gvk::state_init(c->stream, c->cgx_state + blk * gvm::ST_SIZE, q);      // (by value in the launch: nothing to wait for)
    KCHK(c);
    return 0;
}
// the same for an M-space system that may already have taken steps on the host (iteration count, Onsager memory, trace length)
static int cgx_upload_sys(gv_ctx* c, int blk, const CgSys& s) {
    double q[gvm::ST_SIZE];
    for (int i = 0; i < gvm::ST_SIZE; i++) q[i] = 0.0;
    q[gvm::ST_RZ] = s.rz; q[gvm::ST_NORMV] = s.norm_v; q[gvm::ST_PREV_ONS] = s.prev_onsager; q[gvm::ST_ONS] = s.onsager;
    q[gvm::ST_RELERR] = s.rel_err; q[gvm::ST_ACTIVE] = s.active ? 1.0 : 0.0; q[gvm::ST_ITERS] = s.iters;
    q[gvm::ST_CONV] = s.converged; q[gvm::ST_NRELRES] = s.n_relres; q[gvm::ST_DENOISER] = s.denoiser;
    gvk::state_init(c->stream, c->cgx_state + blk * gvm::ST_SIZE, q);
    KCHK(c);
    return 0;
}
// one CG_solverAAT step (state block 0) after d = A (A^T p) has arrived
static int aat_step_device(gv_ctx* c, double* mu, double* p, double* r, double* d, double* z, const double* DG, double tau,
                           double gam2, int max_iter, bool want_rel, CgxStatus* out, double* at_acc = nullptr,
                           const double* at_p = nullptr) {
    const unsigned long long seq = ++c->mbox_seq;
    gvk::aat_step(c->stream, c->cgx_state, mu, p, r, d, z, DG, tau, gam2, c->npad, c->red_partial,
                  want_rel ? c->cgx_rel : nullptr, max_iter, c->mbox_dev, reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK), seq,
                  at_acc, at_p, c->M);
    KCHK(c);
    CgxStatus s2[2];
    if (cgx_wait(c, seq, s2)) return 1;
    *out = s2[0];
    return 0;
}
// one rank: the consumers of the M-space system's reductions add the block partials up themselves (no k_finalize launches)
static bool mspace_self_reduce(const gv_ctx* c, bool multi) { return !multi && c->M > 0; }
// one precondCG_solver step of the M-space system in state block 1 after d = Q p has arrived through an ATx pass whose
// epilogue (CgHook: dot_part / dot_out = red_partial / red_out) left <d,p>
static int mspace_step_device(gv_ctx* c, CgSys& sb, double gam2, double diag, int max_iter, bool multi, CgxStatus* out) {
    double* dst1 = c->cgx_state + gvm::ST_SIZE;
    if (multi && comm_allreduce(c, c->red_out, 8)) return 1;
    const bool self = mspace_self_reduce(c, multi);        // (the pass before this step was enqueued with CgHook::dot_self = self)
    double *a_st[1] = {dst1}, *a_mu[1] = {sb.mu}, *a_r[1] = {sb.r}, *a_z[1] = {sb.z}, *a_part[1] = {c->red_partial + 4 * RED_BLOCKS},
           *a_red[1] = {c->red_out};
    const double *a_p[1] = {sb.p}, *a_v[1] = {sb.v}, *a_d[1] = {sb.d}, *a_dp[1] = {c->red_out}, *a_dpp[1] = {c->red_partial};
    const int nb_ab = gvk::cgx_ab(c->stream, 1, a_st, a_mu, a_p, a_v, a_r, a_d, a_z, a_dp, a_part, a_red, diag, c->M,
                                  self ? a_dpp : nullptr, self ? gvm::atx_dot_blocks(c->plan) : 0);
    KCHK(c);
    if (multi && comm_allreduce(c, c->red_out, 8)) return 1;
    const double* c_red[1] = {c->red_out};
    const double* c_part[1] = {a_part[0]};
    double* c_rel[1] = {sb.relres ? c->cgx_rel + c->cgx_relcap : nullptr};
    const unsigned long long seq = ++c->mbox_seq;
    gvk::cgx_decide(c->stream, 1, a_st, c_red, c_rel, gam2, max_iter, c->cgx_go, c->mbox_dev,
                    reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK), seq, nullptr, nullptr, nullptr,
                    self ? c_part : nullptr, nb_ab);
    KCHK(c);
    CgxStatus s2[2];
    if (cgx_wait(c, seq, s2)) return 1;
    *out = s2[0];
    return 0;
}

// ---- the opening of a solve on the device (device-resident loop): z = r / diag with <r,z> and ||v||^2, the state block of each
// system built on the device from those sums -- one rank: from the block partials (k_state_from_partials); sharded: finalised,
// all-reduced IN STREAM (one message for every system of the call) and picked up by k_state_from_scalars -- and with A^T A v at
// hand the first step of a zero-started solve (d = (tau / diag) A^T A v + gam2 p: no pass) is a device step like every later one.
// The host-driven opening (cg_finish_init, cg_first_step_from_known_product) reads scalars back five times for the two solves of a
// VAMP iteration -- ~280 us of a 13 ms iteration at N = 400k x M = 125k (profiles/r5_forced_multi_gaps.txt), and on a sharded job
// each of them with a host-side all-reduce round trip behind it; this one never touches the host.  Same kernels, same sums in the
// same order, same IEEE divisions and square roots: the same bits.  An empty shard (M == 0) enters the same collectives with zeros.
static int cg_open_device(gv_ctx* c, CgSys* sys, int nsys, double tau, double gam2, double diag, int max_iter, bool multi) {
    const int64_t M = c->M;
    hipStream_t st = c->stream;
    auto part_of = [&](int k) { return c->red_partial + (size_t)k * RED_BLOCKS * 8; };   // the system's own region, as in cg_run_device
    auto red_of = [&](int k) { return c->red_out + 8 * k; };
    for (int k = 0; k < nsys; k++) {
        CgSys& s = sys[k];
        if (M > 0) {
            gvk::cg_step_b(st, s.r, s.d, 0.0, diag, s.z, s.mu, M, part_of(k), multi ? red_of(k) : nullptr);   // z = r / diag (:1152), <r,z> ...
            const double* vv[1] = {s.v};
            gvk::dots(st, 1, vv, vv, M, part_of(k) + 4 * RED_BLOCKS, multi ? red_of(k) + 4 : nullptr);         // ... and ||v||^2
            gvk::copy(st, s.p, s.z, M);                                                                         // p = z (:1154)
        } else
            gvk::fill(st, red_of(k), 8, 0.0);
        s.phase = 1;
        s.req = s.p;
        s.res = s.d;
    }
    KCHK(c);
    if (multi && comm_allreduce(c, c->red_out, (size_t)8 * nsys)) return 1;
    for (int k = 0; k < nsys; k++) {
        CgSys& s = sys[k];
        double* stb = c->cgx_state + k * gvm::ST_SIZE;
        double q[gvm::ST_SIZE];
        for (int i = 0; i < gvm::ST_SIZE; i++) q[i] = 0.0;
        q[gvm::ST_ACTIVE] = 1.0;
        q[gvm::ST_DENOISER] = s.denoiser;
        if (multi) gvk::state_from_scalars(st, stb, q, red_of(k), red_of(k) + 4, true);
        else gvk::state_from_partials(st, stb, q, part_of(k), 4, 0, part_of(k) + 4 * RED_BLOCKS, M, true);
        KCHK(c);
        if (!(s.ata_v && s.ata_v_known && !s.mu0)) continue;
        // the first step from the known product (cg_first_step_from_known_product)
        if (M > 0) {
            gvk::axpby(st, s.d, tau / diag, s.ata_v, gam2, s.p, M);
            const double *xs[1] = {s.d}, *ys[1] = {s.p};
            gvk::dots(st, 1, xs, ys, M, part_of(k), multi ? red_of(k) : nullptr);
        } else
            gvk::fill(st, red_of(k), 8, 0.0);
        KCHK(c);
        if (multi && comm_allreduce(c, red_of(k), 8)) return 1;                    // <d,p>
        double *a_st[1] = {stb}, *a_mu[1] = {s.mu}, *a_r[1] = {s.r}, *a_z[1] = {s.z}, *a_part[1] = {part_of(k) + 4 * RED_BLOCKS},
               *a_red[1] = {red_of(k)};
        const double *a_p[1] = {s.p}, *a_v[1] = {s.v}, *a_d[1] = {s.d}, *a_dp[1] = {red_of(k)}, *a_dpp[1] = {part_of(k)};
        int nb_ab = 0;
        if (M > 0)
            nb_ab = gvk::cgx_ab(st, 1, a_st, a_mu, a_p, a_v, a_r, a_d, a_z, a_dp, a_part, a_red, diag, M, multi ? nullptr : a_dpp,
                                multi ? 0 : gvm::atx_dot_blocks(c->plan));
        else
            gvk::fill(st, red_of(k), 8, 0.0);
        KCHK(c);
        if (multi && comm_allreduce(c, red_of(k), 8)) return 1;                    // <v,mu>, <r,z>, <r,r>
        const double* c_red[1] = {red_of(k)};
        const double* c_part[1] = {a_part[0]};
        double* c_rel[1] = {s.relres ? c->cgx_rel + (size_t)k * c->cgx_relcap : nullptr};
        // (nobody waits for this status: the loop's first step publishes the state of every system)
        gvk::cgx_decide(st, 1, a_st, c_red, c_rel, gam2, max_iter, c->cgx_go, c->mbox_dev,
                        reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK), ++c->mbox_seq, nullptr, nullptr, nullptr,
                        multi ? nullptr : c_part, nb_ab);
        KCHK(c);
        s.iters = 1;                                // (the host's count of steps that cost no product: cg_run_device starts from it)
    }
    return 0;
}

// ride_x / ride_out (may be NULL): out = data::Ax(ride_x), taken along in the free slot of the first round in which only
// one system is still active (a two-vector pass costs what a one-vector pass costs), else by a pass of its own.
static int cg_run(gv_ctx* c, CgSys* sys, int nsys, double tau, double gam2, int max_iter, const double* ride_x = nullptr,
                  double* ride_out = nullptr) {
    const int64_t M = c->M;
    const bool multi = is_multi(c);
    const double diag = tau * (double)(c->N - 1) / (double)c->N + gam2;   // :1137-1138
    const bool device_loop = cgx_usable(c);
    // every system opening without an operator application of its own (zero start, or a warm start whose products are known): the
    // opening runs on the device too (cg_open_device) -- chosen by nothing rank-local (an empty shard takes it with its peers)
    bool dev_open = device_loop && max_iter > 0 && (multi || M > 0);
    for (int k = 0; k < nsys; k++) dev_open = dev_open && !(sys[k].mu0 && !sys[k].ata0);
    if (dev_open && cgx_alloc(c, max_iter)) return 1;
    for (int k = 0; k < nsys; k++) {
        CgSys& s = sys[k];
        if (s.mu0 && s.ata0) {
            // r = v - Q mu0 with A^T A mu0 handed in by the caller (the previous solve left it: Q' mu0 = v' - r'), so the
            // initial residual of a warm start (vamp.cpp:1142-1145) costs no pass over the shard
            gvk::copy(c->stream, s.mu, s.mu0, M);
            gvk::axpby(c->stream, s.r, 1.0, s.v, -tau, s.ata0, M);
            gvk::axpby(c->stream, s.r, 1.0, s.r, -gam2, s.mu0, M);
            if (s.az && s.amu0 != s.az)
                gvk::copy(c->stream, s.az, s.amu0, c->npad);   // A mu0
            KCHK(c);
            if (!dev_open && cg_finish_init(c, s, diag, multi)) return 1;
        } else if (s.mu0) {
            gvk::copy(c->stream, s.mu, s.mu0, M);
            s.phase = 0;
            s.req = s.mu;
            s.res = s.r;
        } else {
            gvk::fill(c->stream, s.mu, M, 0.0);
            if (s.az) gvk::fill(c->stream, s.az, c->npad, 0.0);
            gvk::copy(c->stream, s.r, s.v, M);
            if (!dev_open && cg_finish_init(c, s, diag, multi)) return 1;
        }
        if (max_iter <= 0 && s.phase == 1) s.active = false;
    }
    if (dev_open && cg_open_device(c, sys, nsys, tau, gam2, diag, max_iter, multi)) return 1;
    for (int k = 0; k < nsys && !dev_open; k++)
        if (cg_first_step_from_known_product(c, sys[k], tau, gam2, diag, max_iter, multi)) return 1;
    for (;;) {
        CgSys* act[2];
        int na = 0;
        bool all_stepping = true;
        for (int k = 0; k < nsys; k++)
            if (sys[k].active) {
                act[na++] = &sys[k];
                all_stepping &= sys[k].phase == 1;
            }
        if (na == 0) break;
        if (device_loop && all_stepping) {
            // every active system is past its initial residual: hand the steady state to the device-resident loop
            bool rode = false;
            if (cg_run_device(c, sys, nsys, tau, gam2, max_iter, ride_x, ride_out, &rode, dev_open)) return 1;
            if (rode) ride_x = nullptr;
            break;
        }
        if (na == 2) {
            if (lmmse2_device(c, act[0]->req, act[1]->req, tau, gam2, act[0]->res, act[1]->res)) return 1;
            act[0]->wslot = c->w_n->d;
            act[1]->wslot = c->w_n2->d;
        } else {
            if (ride_x) {                                                 // Ax of the rider in the free slot
                if (ensure_work(c)) return 1;
                if (ax2_device(c, act[0]->req, ride_x, c->w_n->d, ride_out)) return 1;
                if (atx_device(c, c->w_n->d, act[0]->res, act[0]->req, tau, gam2)) return 1;
                ride_x = nullptr;
            } else if (lmmse_device(c, act[0]->req, tau, gam2, act[0]->res))
                return 1;
            act[0]->wslot = c->w_n->d;
        }
        for (int k = 0; k < na; k++)
            if (act[k]->phase == 1 && act[k]->iters == 0 && !act[k]->mu0) cg_capture_first_product(c, *act[k], tau, gam2, diag);
        if (cg_consume_all(c, act, na, gam2, diag, max_iter, multi)) return 1;
    }
    if (ride_x && ax_device(c, ride_x, ride_out)) return 1;
    return 0;
}

static void cg_fill_stats(const CgSys& s, gv_cg_stats* st) {
    if (!st) return;
    st->iters = s.iters;
    st->converged = s.converged;
    st->rel_res = s.rel_err;
    st->onsager = s.onsager;
    st->n_relres = s.n_relres;
}

int gv_cg_solve(gv_ctx* c, const gv_vec* v, const gv_vec* mu_start, double tau, double gam2, int denoiser,
                int max_iter, gv_vec* mu_out, gv_cg_stats* st, double* relres) {
    NEED(c, v->space == GV_SPACE_M && mu_out->space == GV_SPACE_M, "gv_cg_solve: M-space vectors required");
    NEED(c, mu_out != v && mu_out != mu_start, "gv_cg_solve: mu_out must not alias v or mu_start");
    if (ensure_work(c)) return 1;
    const int64_t ax0 = c->cnt.n_ax, atx0 = c->cnt.n_atx;
    CgSys s;
    s.v = v->d; s.mu0 = mu_start ? mu_start->d : nullptr; s.mu = mu_out->d;
    s.r = c->cg_r->d; s.z = c->cg_z->d; s.p = c->cg_p->d; s.d = c->cg_d->d;
    s.denoiser = denoiser; s.relres = relres;
    if (cg_run(c, &s, 1, tau, gam2, max_iter)) return 1;
    cg_fill_stats(s, st);
    if (st) {
        st->n_ax = (int)(c->cnt.n_ax - ax0);
        st->n_atx = (int)(c->cnt.n_atx - atx0);
    }
    return 0;
}

// The LMMSE solve (denoiser = 1, optional warm start) and the Onsager probe solve (denoiser = 0, zero start) of one VAMP
// iteration in lock-step.  n_ax / n_atx of the stats count vector products (2 per shared pass).
int gv_cg_solve2x(gv_ctx* c, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                  int max_iter, gv_vec* mu_a, gv_vec* mu_b, gv_cg_stats* st_a, gv_cg_stats* st_b, double* relres_a,
                  double* relres_b, const gv_cg_extras* ex) {
    return gv_cg_solve2w(c, v_a, mu_start_a, v_b, tau, gam2, max_iter, mu_a, mu_b, st_a, st_b, relres_a, relres_b, ex, nullptr);
}

int gv_cg_solve2w(gv_ctx* c, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                  int max_iter, gv_vec* mu_a, gv_vec* mu_b, gv_cg_stats* st_a, gv_cg_stats* st_b, double* relres_a,
                  double* relres_b, const gv_cg_extras* ex, const gv_cg_warm* wm) {
    NEED(c, v_a->space == GV_SPACE_M && v_b->space == GV_SPACE_M && mu_a->space == GV_SPACE_M && mu_b->space == GV_SPACE_M,
         "gv_cg_solve2: M-space vectors required");
    NEED(c, mu_a != v_a && mu_a != mu_start_a && mu_b != v_b && mu_a != mu_b, "gv_cg_solve2: outputs must not alias inputs");
    gv_cg_extras none{};
    if (!ex) ex = &none;
    NEED(c, (ex->ride_x == nullptr) == (ex->ride_out == nullptr), "gv_cg_solve2x: ride_x and ride_out go together");
    NEED(c, !ex->ride_x || (ex->ride_x->space == GV_SPACE_M && ex->ride_out->space == GV_SPACE_N),
         "gv_cg_solve2x: ride_x is M-space, ride_out N-space");
    NEED(c, !ex->a_mu_a || ex->a_mu_a->space == GV_SPACE_N, "gv_cg_solve2x: a_mu_a is N-space");
    NEED(c, !ex->ata_mu_b || (ex->ata_mu_b->space == GV_SPACE_M && ex->ata_mu_b != mu_b && ex->ata_mu_b != v_b),
         "gv_cg_solve2x: ata_mu_b is M-space and must not alias v_b / mu_b");
    NEED(c, !ex->ata_mu_b || tau != 0.0, "gv_cg_solve2x: ata_mu_b needs tau != 0");
    gv_cg_warm nowarm{};
    if (!wm) wm = &nowarm;
    NEED(c, !wm->ata_mu_start_a || (mu_start_a && wm->ata_mu_start_a->space == GV_SPACE_M && wm->ata_mu_start_a != mu_a),
         "gv_cg_solve2w: ata_mu_start_a goes with mu_start_a, is M-space and must not alias mu_a");
    NEED(c, !wm->ata_mu_start_a || !ex->a_mu_a || (wm->a_mu_start_a && wm->a_mu_start_a->space == GV_SPACE_N),
         "gv_cg_solve2w: a_mu_a with a known warm start needs a_mu_start_a (N-space)");
    NEED(c, !wm->ata_mu_a || (wm->ata_mu_a->space == GV_SPACE_M && wm->ata_mu_a != mu_a && wm->ata_mu_a != v_a &&
                              wm->ata_mu_a != mu_start_a && wm->ata_mu_a != mu_b && wm->ata_mu_a != v_b && tau != 0.0),
         "gv_cg_solve2w: ata_mu_a is M-space, must not alias the systems' vectors, and needs tau != 0");
    NEED(c, !wm->ata_v_b || (wm->ata_v_b->space == GV_SPACE_M && wm->ata_v_b != v_b && wm->ata_v_b != mu_b && wm->ata_v_b != mu_a &&
                             wm->ata_v_b != v_a && tau != 0.0),
         "gv_cg_solve2w: ata_v_b is M-space, must not alias the systems' vectors, and needs tau != 0");
    NEED(c, !wm->have_ata_v_b || wm->ata_v_b, "gv_cg_solve2w: have_ata_v_b without ata_v_b");
    if (ensure_work(c)) return 1;
    if (ensure_w2(c)) return 1;
    for (gv_vec** w : {&c->cg2_r, &c->cg2_z, &c->cg2_p, &c->cg2_d})
        if (!*w && vec_new(c, GV_SPACE_M, w)) return 1;
    const int64_t ax0 = c->cnt.n_ax, atx0 = c->cnt.n_atx;
    CgSys s[2];
    s[0].v = v_a->d; s[0].mu0 = mu_start_a ? mu_start_a->d : nullptr; s[0].mu = mu_a->d;
    s[0].r = c->cg_r->d; s[0].z = c->cg_z->d; s[0].p = c->cg_p->d; s[0].d = c->cg_d->d;
    s[0].denoiser = 1; s[0].relres = relres_a;
    s[0].az = ex->a_mu_a ? ex->a_mu_a->d : nullptr;
    if (wm->ata_mu_start_a) {
        s[0].ata0 = wm->ata_mu_start_a->d;
        s[0].amu0 = wm->a_mu_start_a ? wm->a_mu_start_a->d : nullptr;
    }
    s[1].v = v_b->d; s[1].mu0 = nullptr; s[1].mu = mu_b->d;
    s[1].r = c->cg2_r->d; s[1].z = c->cg2_z->d; s[1].p = c->cg2_p->d; s[1].d = c->cg2_d->d;
    s[1].denoiser = 0; s[1].relres = relres_b;
    s[1].keep_resid = ex->ata_mu_b != nullptr;
    if (wm->ata_v_b) { s[1].ata_v = wm->ata_v_b->d; s[1].ata_v_known = wm->have_ata_v_b != 0; }
    if (cg_run(c, s, 2, tau, gam2, max_iter, ex->ride_x ? ex->ride_x->d : nullptr, ex->ride_out ? ex->ride_out->d : nullptr))
        return 1;
    if (ex->ata_mu_b) {   // Q mu_b = v_b - r_b  =>  A^T A mu_b = (v_b - r_b - gam2 mu_b) / tau
        double* o = ex->ata_mu_b->d;
        gvk::axpby(c->stream, o, 1.0 / tau, s[1].v, -1.0 / tau, s[1].r, c->M);
        gvk::axpby(c->stream, o, 1.0, o, -gam2 / tau, s[1].mu, c->M);
        KCHK(c);
    }
    if (wm->ata_mu_a) {   // the same identity for solve a (its r is current on every exit: residual rule or iteration cap)
        double* o = wm->ata_mu_a->d;
        gvk::axpby(c->stream, o, 1.0 / tau, s[0].v, -1.0 / tau, s[0].r, c->M);
        gvk::axpby(c->stream, o, 1.0, o, -gam2 / tau, s[0].mu, c->M);
        KCHK(c);
    }
    cg_fill_stats(s[0], st_a);
    cg_fill_stats(s[1], st_b);
    if (st_a) { st_a->n_ax = (int)(c->cnt.n_ax - ax0); st_a->n_atx = (int)(c->cnt.n_atx - atx0); }
    if (st_b) { st_b->n_ax = st_a ? st_a->n_ax : 0; st_b->n_atx = st_a ? st_a->n_atx : 0; }
    return 0;
}

int gv_cg_solve2(gv_ctx* c, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                 int max_iter, gv_vec* mu_a, gv_vec* mu_b, gv_cg_stats* st_a, gv_cg_stats* st_b, double* relres_a,
                 double* relres_b) {
    return gv_cg_solve2x(c, v_a, mu_start_a, v_b, tau, gam2, max_iter, mu_a, mu_b, st_a, st_b, relres_a, relres_b, nullptr);
}

// Work vectors of the N-space solvers: nn N-space vectors and nm M-space ones as views into ONE allocation the context keeps
// between calls, zeroed by one fill per call (six allocations, six fills and six releases per solve before: ~0.1 ms of a
// 5 ms VAMP iteration at N = 50k).  vec_del on a view releases the handle only.
static int aat_scratch(gv_ctx* c, int nn, int nm, gv_vec** out) {
    const size_t ln = (size_t)c->npad, lm = (size_t)(c->M > 0 ? c->M : 1);
    const size_t lm_pad = (lm + 63) / 64 * 64;
    const size_t need = ln * (size_t)nn + lm_pad * (size_t)nm;
    if (c->aat_slab_cap < need) {
        if (c->aat_slab) {
            HIPCHK(c, hipStreamSynchronize(c->stream));
            (void)hipFree(c->aat_slab);
            c->aat_slab = nullptr;
            c->aat_slab_cap = 0;
        }
        if (hipMalloc(&c->aat_slab, sizeof(double) * need) != hipSuccess) {
            (void)hipGetLastError();
            c->aat_slab = nullptr;
            return fail(c, "no room for the work vectors of the N-space solver (%zu doubles)", need);
        }
        c->aat_slab_cap = need;
    }
    HIPCHK(c, hipMemsetAsync(c->aat_slab, 0, sizeof(double) * need, c->stream));
    double* at = c->aat_slab;
    for (int k = 0; k < nn + nm; k++) {
        gv_vec* v = new gv_vec();
        v->ctx = c;
        v->space = k < nn ? GV_SPACE_N : GV_SPACE_M;
        v->len = k < nn ? 4 * c->mbytes : c->M;
        v->cap = k < nn ? c->npad : (int64_t)lm;
        v->d = at;
        v->owns = false;
        c->live_vecs.insert(v);
        out[k] = v;
        at += k < nn ? ln : lm_pad;
    }
    return 0;
}

// vamp::lmmse_multAAT (denoiserXXT.cpp:15-35): out = tau A A^T u + gam2 u on N-space device pointers
static int lmmse_aat_device(gv_ctx* c, const double* u, double tau, double gam2, double* tmpM, double* out) {
    if (atx_device(c, u, tmpM)) return 1;
    if (ax_device(c, tmpM, out)) return 1;
    gvk::axpby(c->stream, out, tau, out, gam2, u, c->npad);
    KCHK(c);
    return 0;
}

// z = r / diag (denoiserXXT.cpp:76-77) with <r,z>, and ||v||^2: two reductions into neighbouring scalars, ONE read-back
static int aat_init_scalars(gv_ctx* c, double* r, double* d, const double* diag, double* z, const double* v, double* rz,
                            double* vn2) {
    const int64_t n = c->npad;
    gvk::cg_step_b_diag(c->stream, r, d, 0.0, diag, z, n, c->red_partial, c->red_out);          // red_out[0..1]
    const double* vv[1] = {v};
    gvk::dots(c->stream, 1, vv, vv, n, c->red_partial + 2 * RED_BLOCKS, c->red_out + 2);         // red_out[2]
    KCHK(c);
    double sc[3];
    if (read_scalars(c, 3, sc)) return 1;
    *rz = sc[0];
    *vn2 = sc[2];
    return 0;
}

// vamp::CG_solverAAT (denoiserXXT.cpp:52-130): per-individual diagonal preconditioner, tolerance 1e-4, inner
// products not all-reduced (N-vectors are replicated).  mu_start may be NULL (zeros).
int gv_cg_solve_aat(gv_ctx* c, const gv_vec* v, const gv_vec* mu_start, double tau, double gam2, int max_iter,
                    gv_vec* mu_out, gv_cg_stats* st, double* relres) {
    NEED(c, v->space == GV_SPACE_N && mu_out->space == GV_SPACE_N, "gv_cg_solve_aat: N-space vectors required");
    NEED(c, mu_out != v && mu_out != mu_start, "gv_cg_solve_aat: mu_out must not alias v or mu_start");
    NEED(c, c->mave_p, "gv_cg_solve_aat: gv_people_stats must run first");
    if (ensure_work(c)) return 1;
    hipStream_t s = c->stream;
    const int64_t n = c->npad;
    gv_vec* W[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    auto cleanup = [&]() { for (gv_vec* x : W) vec_del(c, x); };
    if (aat_scratch(c, 5, 0, W)) return 1;
    gv_vec *R = W[0], *Z = W[1], *P = W[2], *D = W[3], *DG = W[4];
    double *r = R->d, *z = Z->d, *p = P->d, *d = D->d, *mu = mu_out->d, *tmpM = c->cg_d->d;
    const int64_t ax0 = c->cnt.n_ax, atx0 = c->cnt.n_atx;
    int rc = 0, iters = 0, converged = 0, n_relres = 0;
    double sc[2], rel_err = 0;
#define AAT_TRY(expr) do { if ((expr) != 0) { rc = 1; goto done; } } while (0)
#define AAT_HIP(expr) do { if ((expr) != hipSuccess) { rc = fail(c, "%s failed", #expr); goto done; } } while (0)
    {
        gvk::aat_diag(s, c->mave_p->d, c->msig_p->d, c->numb_p->d, tau, gam2, (double)c->N, n, DG->d);
        if (mu_start) {
            gvk::copy(s, mu, mu_start->d, n);
            AAT_TRY(lmmse_aat_device(c, mu, tau, gam2, tmpM, r));
            gvk::axpby(s, r, 1.0, v->d, -1.0, r, n);                       // r = v - Q mu (:71-73)
        } else {
            gvk::fill(s, mu, n, 0.0);
            gvk::copy(s, r, v->d, n);
        }
        double rz, vn2;
        AAT_TRY(aat_init_scalars(c, r, d, DG->d, z, v->d, &rz, &vn2));      // z = r / diag (:76-77), <r,z>, ||v||^2
        gvk::copy(s, p, z, n);
        if (cgx_usable(c) && max_iter > 0) {      // scalars on the device, one read-back per step
            AAT_TRY(cgx_alloc(c, max_iter));
            AAT_TRY(cgx_upload_state(c, 0, rz, vn2, 1, true));
            for (int i = 0; i < max_iter; i++) {
                AAT_TRY(atx_device(c, p, tmpM));                                   // A^T p, then A (A^T p) (:22-23)
                AAT_TRY(ax_device(c, tmpM, d));
                CgxStatus stt;
                AAT_TRY(aat_step_device(c, mu, p, r, d, z, DG->d, tau, gam2, max_iter, relres != nullptr, &stt));
                iters = (int)stt.iters;
                rel_err = stt.rel;
                n_relres = (int)stt.nrel;
                if (stt.active == 0.0) { converged = (int)stt.conv; break; }
            }
            if (relres && n_relres > 0) memcpy(relres, c->cgx_rel_h, sizeof(double) * (size_t)n_relres);
            AAT_HIP(hipGetLastError());
            goto done;
        }
        for (int i = 0; i < max_iter; i++) {
            iters = i + 1;
            AAT_TRY(lmmse_aat_device(c, p, tau, gam2, tmpM, d));              // d = Q p (:86)
            const double* xs[1] = {d};
            const double* ys[1] = {p};
            arm_scalars(c);
            gvk::dots(s, 1, xs, ys, n, c->red_partial, c->red_out);
            double dp;
            AAT_TRY(read_scalars(c, 1, &dp));
            const double alpha = rz / dp;                                     // :88
            gvk::axpby(s, mu, 1.0, mu, alpha, p, n);                           // mu += alpha p (:90-93)
            arm_scalars(c);
            gvk::cg_step_b_diag(s, r, d, alpha, DG->d, z, n, c->red_partial, c->red_out);   // :95-105
            AAT_TRY(read_scalars(c, 2, sc));
            const double beta = sc[0] / rz;                                    // :98,:107
            rz = sc[0];
            gvk::axpby(s, p, 1.0, z, beta, p, n);                              // :109-110
            rel_err = sqrt(sc[1] / vn2);                                       // :113-114
            if (relres) relres[i] = rel_err;
            n_relres = i + 1;
            if (rel_err < 1e-4) {                                              // :117,:120
                converged = 1;
                break;
            }
        }
        AAT_HIP(hipGetLastError());
    }
done:
#undef AAT_TRY
#undef AAT_HIP
    if (st) {
        st->iters = iters;
        st->converged = converged;
        st->rel_res = rel_err;
        st->onsager = 0;
        st->n_ax = (int)(c->cnt.n_ax - ax0);
        st->n_atx = (int)(c->cnt.n_atx - atx0);
        st->n_relres = n_relres;
    }
    (void)hipStreamSynchronize(s);
    cleanup();
    return rc;
}

// ---- --use-XXT-denoiser 1: the N-space LMMSE solve and the M-space Onsager solve of one iteration on shared passes ------
// System A (vamp::CG_solverAAT): Q_A u = tau A (A^T u) + gam2 u -- an ATx followed by an Ax.
// System B (precondCG_solver, denoiser = 0, zero start): Q_B x = tau A^T (A x) + gam2 x -- an Ax followed by an ATx.
// The two operators are made of the same two passes in opposite order, so run in step -- A half an application out of
// phase with B -- every pass over the shard after the first carries one half-application of each (two-vector
// kernels).  Per system the iterates, scalars, stopping rules and results are those of gv_cg_solve_aat / gv_cg_solve.
// When A has finished its A^T mu_a (wanted for x2_hat = r2 + gamw A^T mu_a, denoiserXXT.cpp:47-49) is taken the same way.
namespace {
struct HalfOp {
    bool pending = false;
    int first = 0;          // kind of the first half: 0 = Ax, 1 = ATx
    int stage = 0;          // 0: first half to do, 1: second half to do
    bool one_half = false;  // only the first half is wanted (A^T mu_a)
    const double* src = nullptr;
    double* mid = nullptr;
    double* dst = nullptr;
    int kind() const { return stage == 0 ? first : 1 - first; }
    const double* in() const { return stage == 0 ? src : mid; }
    double* out() const { return (stage == 0 && !one_half) ? mid : dst; }
};
}  // namespace

int gv_cg_solve_aat2(gv_ctx* c, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                     int max_iter, gv_vec* mu_a, gv_vec* at_mu_a, gv_vec* mu_b, gv_cg_stats* st_a, gv_cg_stats* st_b,
                     double* relres_a, double* relres_b, gv_vec* aat_mu_a, gv_vec* ata_mu_b) {
    // (without gv_aat_warm::pre_x nothing writes v_a)
    return gv_cg_solve_aat2w(c, const_cast<gv_vec*>(v_a), mu_start_a, v_b, tau, gam2, max_iter, mu_a, at_mu_a, mu_b, st_a, st_b, relres_a,
                             relres_b, aat_mu_a, ata_mu_b, nullptr);
}

// gv_aat_warm (gvamp.h).  aat_mu_start_a: A A^T mu_start_a, as the previous call left it in aat_mu_a -- the initial residual of
// the warm-started N-space solve (denoiserXXT.cpp:76-78) is then formed without its ATx + Ax pair.  accumulate_at_mu_a: A^T mu_a
// is built from the A^T p_k every application starts with (A^T mu_a = A^T mu_start_a + sum_k alpha_k A^T p_k) instead of by a
// closing ATx pass; A^T mu_start_a comes from at_mu_start_a, or from the explicit opening application, or is 0 for a zero start.
int gv_cg_solve_aat2w(gv_ctx* c, gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                      int max_iter, gv_vec* mu_a, gv_vec* at_mu_a, gv_vec* mu_b, gv_cg_stats* st_a, gv_cg_stats* st_b,
                      double* relres_a, double* relres_b, gv_vec* aat_mu_a, gv_vec* ata_mu_b, const gv_aat_warm* wm) {
    gv_aat_warm nowarm{};
    if (!wm) wm = &nowarm;
    const gv_vec* aat_mu_start_a = wm->aat_mu_start_a;
    NEED(c, !aat_mu_start_a || (mu_start_a && aat_mu_start_a->space == GV_SPACE_N && aat_mu_start_a != mu_a),
         "gv_cg_solve_aat2w: aat_mu_start_a goes with mu_start_a, is N-space and must not alias mu_a");
    NEED(c, !wm->at_mu_start_a || (wm->accumulate_at_mu_a && mu_start_a && wm->at_mu_start_a->space == GV_SPACE_M),
         "gv_cg_solve_aat2w: at_mu_start_a goes with mu_start_a and accumulate_at_mu_a, and is M-space");
    NEED(c, !(wm->accumulate_at_mu_a && aat_mu_start_a && !wm->at_mu_start_a),
         "gv_cg_solve_aat2w: accumulating A^T mu_a from a start whose opening application is skipped needs at_mu_start_a");
    NEED(c, !wm->ata_v_b || (wm->ata_v_b->space == GV_SPACE_M && wm->ata_v_b != v_b && wm->ata_v_b != mu_b && wm->ata_v_b != at_mu_a &&
                             tau != 0.0),
         "gv_cg_solve_aat2w: ata_v_b is M-space, must not alias the systems' vectors, and needs tau != 0");
    NEED(c, !wm->have_ata_v_b || wm->ata_v_b, "gv_cg_solve_aat2w: have_ata_v_b without ata_v_b");
    NEED(c, (wm->pre_x == nullptr) == (wm->pre_out == nullptr) && (wm->ride_x == nullptr) == (wm->ride_out == nullptr),
         "gv_cg_solve_aat2w: pre_x / pre_out and ride_x / ride_out go together");
    NEED(c, !wm->pre_x || (wm->pre_x->space == GV_SPACE_M && wm->pre_out->space == GV_SPACE_N && wm->pre_out != v_a &&
                           wm->pre_out != mu_a && wm->pre_out != aat_mu_a),
         "gv_cg_solve_aat2w: pre_x is M-space, pre_out N-space and neither v_a, mu_a nor aat_mu_a");
    NEED(c, !wm->ride_x || (wm->ride_x->space == GV_SPACE_M && wm->ride_out->space == GV_SPACE_N && wm->ride_out != v_a &&
                            wm->ride_out != mu_a && wm->ride_out != aat_mu_a && wm->ride_out != wm->pre_out),
         "gv_cg_solve_aat2w: ride_x is M-space, ride_out N-space and none of the solver's N-space vectors");
    double* const at_acc = wm->accumulate_at_mu_a ? at_mu_a->d : nullptr;
    NEED(c, v_a->space == GV_SPACE_N && mu_a->space == GV_SPACE_N && (!mu_start_a || mu_start_a->space == GV_SPACE_N),
         "gv_cg_solve_aat2: system a lives in N-space");
    NEED(c, v_b->space == GV_SPACE_M && mu_b->space == GV_SPACE_M && at_mu_a->space == GV_SPACE_M,
         "gv_cg_solve_aat2: system b and A^T mu_a live in M-space");
    NEED(c, mu_a != v_a && mu_a != mu_start_a && mu_b != v_b && at_mu_a != mu_b && at_mu_a != v_b,
         "gv_cg_solve_aat2: outputs must not alias inputs");
    NEED(c, !aat_mu_a || (aat_mu_a->space == GV_SPACE_N && aat_mu_a != mu_a && aat_mu_a != v_a), "gv_cg_solve_aat2: aat_mu_a");
    NEED(c, !ata_mu_b || (ata_mu_b->space == GV_SPACE_M && ata_mu_b != mu_b && ata_mu_b != v_b && ata_mu_b != at_mu_a),
         "gv_cg_solve_aat2: ata_mu_b");
    NEED(c, !(aat_mu_a || ata_mu_b) || tau != 0.0, "gv_cg_solve_aat2: the by-products need tau != 0");
    NEED(c, c->mave_p, "gv_cg_solve_aat2: gv_people_stats must run first");
    if (ensure_work(c)) return 1;
    hipStream_t s = c->stream;
    const int64_t n = c->npad, M = c->M;
    const bool multi = is_multi(c);
    const bool dev = cgx_usable(c) && max_iter > 0;      // scalars of both systems on the device (state blocks 0 = a, 1 = b)
    if (dev && cgx_alloc(c, max_iter)) return 1;
    gv_vec* W[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    auto cleanup = [&]() { for (gv_vec* x : W) vec_del(c, x); };
    if (aat_scratch(c, 5, 1, W)) return 1;
    gv_vec *R = W[0], *Z = W[1], *P = W[2], *D = W[3], *DG = W[4], *MA = W[5];
    double *r = R->d, *z = Z->d, *p = P->d, *d = D->d, *mu = mu_a->d;
    const int64_t ax0 = c->cnt.n_ax, atx0 = c->cnt.n_atx;
    int rc = 0;
#define MIX_TRY(expr) do { if ((expr) != 0) { rc = 1; goto done; } } while (0)
#define MIX_HIP(expr) do { if ((expr) != hipSuccess) { rc = fail(c, "%s failed", #expr); goto done; } } while (0)
    // ---- system A state (denoiserXXT.cpp:52-130)
    int a_iters = 0, a_conv = 0, a_nrel = 0, a_phase = 0;   // phase 0: waiting for Q mu0, 1: waiting for Q p, 2: A^T mu, 3: done
    double a_rz = 0, a_vn2 = 0, a_rel = 0;
    HalfOp ha, hb;
    ha.first = 1;
    hb.first = 0;
    // ---- system B state: the M-space machinery of gv_cg_solve
    CgSys sb;
    const double diag_b = tau * (double)(c->N - 1) / (double)c->N + gam2;
    {
        sb.v = v_b->d; sb.mu0 = nullptr; sb.mu = mu_b->d;
        sb.r = c->cg_r->d; sb.z = c->cg_z->d; sb.p = c->cg_p->d; sb.d = c->cg_d->d;
        sb.denoiser = 0; sb.relres = relres_b;
        sb.keep_resid = ata_mu_b != nullptr;
        gvk::fill(s, sb.mu, M, 0.0);
        gvk::copy(s, sb.r, sb.v, M);
        if (wm->ata_v_b) { sb.ata_v = wm->ata_v_b->d; sb.ata_v_known = wm->have_ata_v_b != 0; }
        if (dev && !multi && M > 0) {
            // One rank, scalars on the device: nothing of the opening needs the host.  z = r / diag with <r,z> and ||v||^2 stay
            // block partials, the state block is built from them on the device (k_state_from_partials), and with A^T A v_b at
            // hand the first step -- d = (tau / diag) A^T A v + gam2 p, no pass -- is a device step like every later one.  ONE
            // status read-back where the host-driven opening took four (cg_finish_init, then <d,p>, <v,mu> and the norms).
            double* stB = c->cgx_state + gvm::ST_SIZE;
            gvk::cg_step_b(s, sb.r, sb.d, 0.0, diag_b, sb.z, sb.mu, M, c->red_partial, nullptr);
            const double* vv[1] = {sb.v};
            gvk::dots(s, 1, vv, vv, M, c->red_partial + 4 * RED_BLOCKS, nullptr);
            gvk::copy(s, sb.p, sb.z, M);
            sb.phase = 1; sb.req = sb.p; sb.res = sb.d;
            double q[gvm::ST_SIZE];
            for (int i = 0; i < gvm::ST_SIZE; i++) q[i] = 0.0;
            q[gvm::ST_ACTIVE] = 1.0;                        // (dev implies max_iter > 0)
            gvk::state_from_partials(s, stB, q, c->red_partial, 4, 0, c->red_partial + 4 * RED_BLOCKS, M, true);
            if (sb.ata_v && sb.ata_v_known) {
                gvk::axpby(s, sb.d, tau / diag_b, sb.ata_v, gam2, sb.p, M);
                const double *xs[1] = {sb.d}, *ys[1] = {sb.p};
                gvk::dots(s, 1, xs, ys, M, c->red_partial, nullptr);
                double *a_st[1] = {stB}, *a_mu[1] = {sb.mu}, *a_r[1] = {sb.r}, *a_z[1] = {sb.z},
                       *a_part[1] = {c->red_partial + 4 * RED_BLOCKS}, *a_red[1] = {c->red_out};
                const double *a_p[1] = {sb.p}, *a_v[1] = {sb.v}, *a_d[1] = {sb.d}, *a_dp[1] = {c->red_out}, *a_dpp[1] = {c->red_partial};
                const int nb_ab = gvk::cgx_ab(s, 1, a_st, a_mu, a_p, a_v, a_r, a_d, a_z, a_dp, a_part, a_red, diag_b, M, a_dpp,
                                              gvm::atx_dot_blocks(c->plan));
                const double* c_red[1] = {c->red_out};
                const double* c_part[1] = {a_part[0]};
                double* c_rel[1] = {sb.relres ? c->cgx_rel + c->cgx_relcap : nullptr};
                const unsigned long long seq = ++c->mbox_seq;
                gvk::cgx_decide(s, 1, a_st, c_red, c_rel, gam2, max_iter, c->cgx_go, c->mbox_dev,
                                reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK), seq, nullptr, nullptr, nullptr, c_part, nb_ab);
                MIX_HIP(hipGetLastError());
                CgxStatus s2[2];
                MIX_TRY(cgx_wait(c, seq, s2));
                sb.iters = (int)s2[0].iters; sb.converged = (int)s2[0].conv; sb.onsager = s2[0].ons; sb.n_relres = (int)s2[0].nrel;
                sb.rel_err = s2[0].rel; sb.active = s2[0].active != 0.0;
            }
        } else {
            MIX_TRY(cg_finish_init(c, sb, diag_b, multi));
            if (max_iter <= 0) sb.active = false;
            // (A^T A v_b at hand: the first step of solve b without its Ax + ATx; the residual trace of a device run lives on the
            // device, its first entry goes there)
            double rel0 = 0;
            double* keep_rel = sb.relres;
            if (dev && sb.relres) sb.relres = &rel0;
            MIX_TRY(cg_first_step_from_known_product(c, sb, tau, gam2, diag_b, max_iter, multi));
            sb.relres = keep_rel;
            if (dev && keep_rel && sb.n_relres > 0) c->cgx_rel_h[c->cgx_relcap] = rel0;      // (host memory: the device writes the later entries)
            if (dev) MIX_TRY(cgx_upload_sys(c, 1, sb));
        }
    }
    {
        auto a_init_scalars = [&]() -> int {   // z = r / diag (:76-77), <r,z>, ||v||^2, p = z
            if (dev) {      // the scalars stay on the device: the state block is built from the block partials (no read-back; N-space
                            // sums are not all-reduced, so this holds for a sharded job too)
                gvk::cg_step_b_diag(s, r, d, 0.0, DG->d, z, n, c->red_partial, nullptr);
                const double* vv[1] = {v_a->d};
                gvk::dots(s, 1, vv, vv, n, c->red_partial + 2 * RED_BLOCKS, nullptr);
                double q[gvm::ST_SIZE];
                for (int i = 0; i < gvm::ST_SIZE; i++) q[i] = 0.0;
                q[gvm::ST_ACTIVE] = 1.0;
                q[gvm::ST_DENOISER] = 1.0;
                gvk::state_from_partials(s, c->cgx_state, q, c->red_partial, 2, 0, c->red_partial + 2 * RED_BLOCKS, n, false);
                if (hipGetLastError() != hipSuccess) return 1;
            } else if (aat_init_scalars(c, r, d, DG->d, z, v_a->d, &a_rz, &a_vn2))
                return 1;
            gvk::copy(s, p, z, n);
            return 0;
        };
        auto a_post = [&](const double* src, double* dst) {   // request Q_A src -> dst
            ha.pending = true; ha.stage = 0; ha.one_half = false; ha.src = src; ha.mid = MA->d; ha.dst = dst;
        };
        auto a_finish = [&]() {                                // the solve is over: A^T mu_a is the last request
            if (at_acc) { a_phase = 3; ha.pending = false; return; }   // ... unless it has been accumulated along the way
            a_phase = 2;
            ha.pending = true; ha.stage = 0; ha.one_half = true; ha.src = mu; ha.mid = nullptr; ha.dst = at_mu_a->d;
        };
        gvk::aat_diag(s, c->mave_p->d, c->msig_p->d, c->numb_p->d, tau, gam2, (double)c->N, n, DG->d);
        if (at_acc) {
            if (wm->at_mu_start_a) {
                if (wm->at_mu_start_a->d != at_acc)
                    gvk::copy(s, at_acc, wm->at_mu_start_a->d, M);
            } else if (!mu_start_a)
                gvk::fill(s, at_acc, M, 0.0);
            // (explicit warm start: the first half of its opening application is A^T mu0 -- copied when it arrives, below)
        }
        // solve a starts here -- at once, or (gv_aat_warm.pre_x) when the pass that completes its right-hand side has come back
        auto a_start = [&]() -> int {
            if (mu_start_a && aat_mu_start_a) {
                // r = v - (tau A A^T mu0 + gam2 mu0) from the product the previous solve left (Q' mu0 = v' - r'): no pass
                gvk::copy(s, mu, mu_start_a->d, n);
                gvk::axpby(s, r, 1.0, v_a->d, -tau, aat_mu_start_a->d, n);
                gvk::axpby(s, r, 1.0, r, -gam2, mu_start_a->d, n);
                if (a_init_scalars()) return 1;
                a_phase = 1;
                if (max_iter > 0) a_post(p, d); else a_finish();
            } else if (mu_start_a) {
                gvk::copy(s, mu, mu_start_a->d, n);
                a_phase = 0;
                a_post(mu, r);
            } else {
                gvk::fill(s, mu, n, 0.0);
                gvk::copy(s, r, v_a->d, n);
                if (a_init_scalars()) return 1;
                a_phase = 1;
                if (max_iter > 0) a_post(p, d); else a_finish();
            }
            return 0;
        };
        // gv_aat_warm.pre_x: v_a <- v_a - A pre_x first (v_a = y - A r2 of denoiserXXT.cpp:40-42, the A r2 taken in the pass
        // that also carries the first half-application of solve b, so that the two solves run in phase from the start);
        // gv_aat_warm.ride_x: ride_out = A ride_x (z1 = A x1_hat of vamp.cpp:429) in whatever Ax pass has a slot free
        HalfOp hr, hz;
        if (wm->pre_x) { hr.pending = true; hr.first = 0; hr.one_half = true; hr.src = wm->pre_x->d; hr.dst = wm->pre_out->d; }
        else if (a_start()) { rc = fail(c, "gv_cg_solve_aat2w: starting solve a failed"); goto done; }
        if (wm->ride_x) { hz.pending = true; hz.first = 0; hz.one_half = true; hz.src = wm->ride_x->d; hz.dst = wm->ride_out->d; }
        auto b_post = [&]() {
            hb.pending = sb.active;
            hb.stage = 0; hb.one_half = false; hb.src = sb.req; hb.mid = c->w_n->d; hb.dst = sb.res;
        };
        b_post();

        bool pipelined = false;
        for (;;) {
            // ---- the steady state, enqueued ahead of its statuses ---------------------------------------------------------------
            // Once both systems are stepping and share their passes -- X: an ATx pass carrying the first half of Q_A (A^T p_a) and the
            // second half of Q_B; Y: an Ax pass carrying the second half of Q_A and the first half of Q_B -- nothing the host does
            // between two passes depends on a scalar except WHICH systems are still running.  So the host enqueues pass k + 1 and its
            // step kernels before it has read the status of the step behind pass k (one mailbox slot per parity, as cg_run_device):
            // the device never waits for a launch, whatever the host's launch rate (profiles/r3_cfg5_*.txt: 26 waits of 15-40 us per
            // three iterations on a fast host, every small-kernel section host-paced on a slow one).  What speculation costs:
            //   * a system that finished at step k still has its next first half in the pass enqueued meanwhile: a wasted SLOT of a
            //     two-vector pass (free) whose step kernels return on the system's ACTIVE flag; products and passes are counted as the
            //     host-paced loop would have made them (the wasted slots and the dropped passes are taken back below);
            //   * a pass enqueued after BOTH had finished is dropped on the device (*go == 0: the k_aat_decide / k_cgx_decide of the
            //     two systems keep the flag together);
            //   * the rider (z1 = A x1_hat) wants the slot a finished system leaves in an Ax pass, in the very pass the host-paced
            //     loop would have used -- one pass before the host knows: it is placed on the device (gvm::CgHook::ride, as in
            //     cg_run_device), every Ax pass keeps both slots while it is pending.
            // Per system the kernels, their order and their operands are those of the loop below: bit-identical results.
            {
                // (a pass or two of the host-paced loop come first: the right-hand side of solve a, and one pass that aligns the two
                // systems when they start half an application apart)
                if (!pipelined && dev && at_acc && max_iter > 0 && !hr.pending && a_phase == 1 && ha.pending && hb.pending &&
                    sb.active && !ha.one_half && !hb.one_half && ha.kind() == hb.kind()) {
                    pipelined = true;
                    double* stA = c->cgx_state;
                    double* stB = c->cgx_state + gvm::ST_SIZE;
                    unsigned long long* flag_dev = reinterpret_cast<unsigned long long*>(c->mbox_dev + RED_MAXK);
                    int* d_ride = c->cgx_go + 1;
                    bool ride_pending = hz.pending;
                    gvk::set_ints(s, c->cgx_go, 1, ride_pending ? 1 : 0);
                    // from here on p_a <- z_a + beta p_a belongs to the k_prep_atx of the next X pass; a host-paced step before this
                    // point has advanced p_a already and must not be seen as pending
                    gvk::fill(s, stA + gvm::ST_STEPPED, 1, 0.0);
                    struct PassRec { int kind; bool slotA, slotB; };
                    struct StepRec { int sys; unsigned long long seq; int after_pass; };
                    std::vector<PassRec> passes;
                    std::vector<StepRec> steps;
                    size_t applied = 0;                         // statuses read so far
                    bool hostA = true, hostB = true, rode = false;
                    int endA = -1, endB = -1;                   // index of the pass whose step finished the system
                    const int itersA0 = a_iters;
                    bool b_first_capture = sb.iters == 0;       // its first application delivers A^T A v_b (cg_capture_first_product)
                    auto apply = [&](const StepRec& r, const CgxStatus& stt) {
                        if (r.sys == 0) {
                            a_iters = (int)stt.iters; a_rel = stt.rel; a_nrel = (int)stt.nrel;
                            if (stt.active == 0.0 && hostA) { hostA = false; a_conv = (int)stt.conv; endA = r.after_pass; }
                        } else {
                            sb.iters = (int)stt.iters; sb.converged = (int)stt.conv; sb.onsager = stt.ons; sb.n_relres = (int)stt.nrel;
                            sb.rel_err = stt.rel;
                            if (stt.active == 0.0 && hostB) { hostB = false; sb.active = false; endB = r.after_pass; }
                        }
                        if (stt.ride == 2.0 && ride_pending) { ride_pending = false; rode = true; }
                    };
                    int kind = ha.kind();                       // 1: X (ATx), 0: Y (Ax)
                    const int64_t ax0p = c->cnt.n_ax, atx0p = c->cnt.n_atx, axp0 = c->cnt.n_ax_pass, atxp0 = c->cnt.n_atx_pass;
                    while (hostA || hostB) {
                        // both slots are kept while the rider waits for one (Ax passes); otherwise the slots of the systems the host
                        // believes running
                        const bool slotA = hostA || (kind == 0 && ride_pending), slotB = hostB || (kind == 0 && ride_pending);
                        gvm::CgHook hk;
                        hk.go = c->cgx_go;
                        const int iB = slotA ? 1 : 0;           // B's slot index in the pass
                        const bool self = mspace_self_reduce(c, multi);
                        if (kind == 1) {                        // X: MA = A^T p_a | d_b = tau A^T w + gam2 p_b, <d_b, p_b>
                            // (p_a <- z_a + beta p_a on the way in: the A-step behind the Y pass leaves it to this launch)
                            if (slotA) { hk.state[0] = stA; hk.pn[0] = p; hk.zn[0] = z; }
                            if (slotB) { hk.state[iB] = stB; hk.dot_part[iB] = c->red_partial; hk.dot_out[iB] = c->red_out; hk.dot_self = self; }
                            if (slotA && slotB) MIX_TRY(atx2_device(c, p, c->w_n->d, MA->d, sb.d, nullptr, sb.p, tau, gam2, &hk));
                            else if (slotA) MIX_TRY(atx_device(c, p, MA->d, nullptr, tau, gam2, &hk));
                            else MIX_TRY(atx_device(c, c->w_n->d, sb.d, sb.p, tau, gam2, &hk));
                            passes.push_back(PassRec{1, slotA, slotB});
                            if (slotB && hostB) {
                                if (b_first_capture) { cg_capture_first_product(c, sb, tau, gam2, diag_b); b_first_capture = false; }
                                if (multi && comm_allreduce(c, c->red_out, 8)) { rc = 1; goto done; }
                                double *a_st[1] = {stB}, *a_mu[1] = {sb.mu}, *a_r[1] = {sb.r}, *a_z[1] = {sb.z},
                                       *a_part[1] = {c->red_partial + 4 * RED_BLOCKS}, *a_red[1] = {c->red_out};
                                const double *a_p[1] = {sb.p}, *a_v[1] = {sb.v}, *a_d[1] = {sb.d}, *a_dp[1] = {c->red_out},
                                             *a_dpp[1] = {c->red_partial};
                                const int nb_ab = gvk::cgx_ab(s, 1, a_st, a_mu, a_p, a_v, a_r, a_d, a_z, a_dp, a_part, a_red, diag_b, M,
                                                              self ? a_dpp : nullptr, self ? gvm::atx_dot_blocks(c->plan) : 0);
                                if (multi && comm_allreduce(c, c->red_out, 8)) { rc = 1; goto done; }
                                const double* c_red[1] = {c->red_out};
                                const double* c_part[1] = {a_part[0]};
                                double* c_rel[1] = {sb.relres ? c->cgx_rel + c->cgx_relcap : nullptr};
                                const unsigned long long seq = ++c->mbox_seq;
                                gvk::cgx_decide(s, 1, a_st, c_red, c_rel, gam2, max_iter, c->cgx_go, c->mbox_dev, flag_dev, seq, nullptr, stA,
                                                d_ride, self ? c_part : nullptr, nb_ab);
                                MIX_HIP(hipGetLastError());
                                steps.push_back(StepRec{1, seq, (int)passes.size() - 1});
                            }
                        } else {                                // Y: d_a = A MA | w_n = A p_b (p_b <- z_b + beta p_b on the way in)
                            if (slotA) hk.state[0] = stA;       // (for the rider's sake: no search direction to advance in this slot)
                            // one rank, a vector that fits RED_BLOCKS blocks: the epilogue of the pass leaves d_a = tau d_a + gam2 p_a
                            // and the block partials of <d_a, p_a> itself (k_aat_dq's work, bit for bit)
                            const bool dq_fused = slotA && hostA && !multi && M > 0 && (n + 255) / 256 <= RED_BLOCKS;
                            if (dq_fused) { hk.dq_p[0] = p; hk.dq_part[0] = c->red_partial; hk.dq_tau = tau; hk.dq_gam2 = gam2; }
                            if (slotB) { hk.state[iB] = stB; hk.p[iB] = sb.p; hk.z[iB] = sb.z; }
                            const bool may_ride = ride_pending && slotA && slotB;
                            // one rank: the epilogue delivers the rider's product itself; sharded: it is copied out of the slot behind the
                            // cross-rank sum and the scaling of that slot's output
                            if (may_ride) { hk.ride = d_ride; hk.alt_x = wm->ride_x->d; if (!multi) hk.ride_out = wm->ride_out->d; }
                            if (slotA && slotB) MIX_TRY(ax2_device(c, MA->d, sb.p, d, c->w_n->d, &hk));
                            else if (slotA) MIX_TRY(ax_device(c, MA->d, d, &hk));
                            else MIX_TRY(ax_device(c, sb.p, c->w_n->d, &hk));
                            passes.push_back(PassRec{0, slotA, slotB});
                            // (the rider's product went straight to ride_out; the A-step's decide marks it done, else a launch of its own)
                            if (may_ride && multi) gvk::ride_copy(s, wm->ride_out->d, d, c->w_n->d, stA, stB, d_ride, n);
                            if (may_ride && !hostA) gvk::ride_mark(s, stA, stB, d_ride);
                            if (slotA && hostA) {
                                const unsigned long long seq = ++c->mbox_seq;
                                gvk::aat_step(s, stA, mu, p, r, d, z, DG->d, tau, gam2, n, c->red_partial,
                                              relres_a ? c->cgx_rel : nullptr, max_iter, c->mbox_dev, flag_dev, seq, at_acc, MA->d, M, stB,
                                              c->cgx_go, d_ride, may_ride ? d_ride : nullptr, false, dq_fused);
                                MIX_HIP(hipGetLastError());
                                steps.push_back(StepRec{0, seq, (int)passes.size() - 1});
                            }
                        }
                        kind = 1 - kind;
                        // the status of everything but the step just enqueued -- of that one too once the solve has made as many
                        // passes as the previous one needed (gv_ctx::spec_hint_passes): it is probably the last
                        // -- for the next two passes only (one more A-step): a solve that outgrows the hint reads one step behind again
                        const bool likely_last = c->spec_hint_passes > 0 && (int)passes.size() >= c->spec_hint_passes &&
                                                 (int)passes.size() <= c->spec_hint_passes + 2;
                        while (applied + (likely_last ? 0 : 1) < steps.size()) {
                            CgxStatus s2[2];
                            MIX_TRY(cgx_wait(c, steps[applied].seq, s2));
                            apply(steps[applied], s2[0]);
                            applied++;
                        }
                    }
                    while (applied < steps.size()) {            // drain (a dropped step republishes the unchanged state)
                        CgxStatus s2[2];
                        MIX_TRY(cgx_wait(c, steps[applied].seq, s2));
                        apply(steps[applied], s2[0]);
                        applied++;
                    }
                    // ---- accounting: what the host-paced loop would have counted.  A pass after the one whose step finished the
                    // LAST system was dropped on the device; a slot of a system in a pass after the one that finished it was wasted
                    // (unless the rider took it: it rode in the first Ax pass in which exactly one system had finished).
                    {
                        const int last_useful = endA > endB ? endA : endB;
                        c->spec_hint_passes = last_useful + 1;
                        int64_t d_ax = 0, d_atx = 0, d_axp = 0, d_atxp = 0;
                        bool rider_slot_found = false;
                        for (int i = 0; i < (int)passes.size(); i++) {
                            const PassRec& q = passes[i];
                            const int nslots = (q.slotA ? 1 : 0) + (q.slotB ? 1 : 0);
                            if (i > last_useful) {              // dropped entirely
                                (q.kind ? d_atx : d_ax) += nslots;
                                (q.kind ? d_atxp : d_axp) += 1;
                                continue;
                            }
                            int wasted = 0;
                            if (q.slotA && endA >= 0 && i > endA) wasted++;
                            if (q.slotB && endB >= 0 && i > endB) wasted++;
                            if (wasted && rode && !rider_slot_found && q.kind == 0 && wasted == 1 && nslots == 2) { rider_slot_found = true; wasted--; }
                            (q.kind ? d_atx : d_ax) += wasted;
                        }
                        c->cnt.n_ax -= d_ax; c->cnt.n_atx -= d_atx; c->cnt.n_ax_pass -= d_axp; c->cnt.n_atx_pass -= d_atxp;
                        (void)ax0p; (void)atx0p; (void)axp0; (void)atxp0; (void)itersA0;
                    }
                    a_phase = 3;
                    ha.pending = false;
                    hb.pending = false;
                    sb.active = false;
                    if (rode) hz.pending = false;
                }
            }

            HalfOp* todo[2];
            int nt = 0;
            // up to two pending half-applications of the same kind share a pass.  Priority: the right-hand side of solve a,
            // solve a, solve b, the rider (which only ever fills a slot that would have stayed empty, or closes the call)
            HalfOp* cand[4] = {&hr, &ha, &hb, &hz};
            for (HalfOp* h : cand)
                if (h->pending && (nt == 0 || (nt == 1 && h->kind() == todo[0]->kind()))) todo[nt++] = h;
            if (nt == 0) break;
            // system b's second half is the ATx of lmmse_mult: its tau * . + gam2 * req epilogue is fused as in gv_cg_solve
            auto addx = [&](const HalfOp* h) -> const double* { return (h == &hb && h->stage == 1) ? sb.req : nullptr; };
            // device scalars: system b's slot of the pass advances its search direction on the way in (Ax half) and leaves
            // d = Q_B p and <d,p> on the way out (ATx half)
            gvm::CgHook hk;
            const gvm::CgHook* hkp = nullptr;
            for (int k = 0; k < nt && dev; k++)
                if (todo[k] == &hb) {
                    hk.state[k] = c->cgx_state + gvm::ST_SIZE;
                    if (hb.stage == 0) { hk.p[k] = sb.p; hk.z[k] = sb.z; }
                    else { hk.dot_part[k] = c->red_partial; hk.dot_out[k] = c->red_out; hk.dot_self = mspace_self_reduce(c, multi); }
                    hkp = &hk;
                }
            if (nt == 2) {
                if (todo[0]->kind() == 0) MIX_TRY(ax2_device(c, todo[0]->in(), todo[1]->in(), todo[0]->out(), todo[1]->out(), hkp));
                else MIX_TRY(atx2_device(c, todo[0]->in(), todo[1]->in(), todo[0]->out(), todo[1]->out(), addx(todo[0]),
                                         addx(todo[1]), tau, gam2, hkp));
            } else {
                if (todo[0]->kind() == 0) MIX_TRY(ax_device(c, todo[0]->in(), todo[0]->out(), hkp));
                else MIX_TRY(atx_device(c, todo[0]->in(), todo[0]->out(), addx(todo[0]), tau, gam2, hkp));
            }
            for (int k = 0; k < nt; k++) {
                HalfOp* h = todo[k];
                if (h->stage == 0 && !h->one_half) {                               // second half still to come
                    if (h == &ha && a_phase == 0 && at_acc)                         // A^T mu0 of an explicit warm start
                        gvk::copy(s, at_acc, MA->d, M);
                    h->stage = 1;
                    continue;
                }
                h->pending = false;
                if (h == &hz) continue;                                            // the rider's product is in place
                if (h == &hr) {                                                    // v_a = y - A r2 is complete: solve a can start
                    gvk::axpby(s, v_a->d, 1.0, v_a->d, wm->pre_scale != 0.0 ? -wm->pre_scale : -1.0, hr.dst, n);
                    MIX_TRY(a_start());
                    continue;
                }
                if (h == &hb) {                                                    // Q_B req complete (epilogue fused above)
                    sb.wslot = c->w_n->d;
                    if (sb.iters == 0) cg_capture_first_product(c, sb, tau, gam2, diag_b);   // its first application: A^T A v_b
                    if (dev) {
                        CgxStatus stt;
                        MIX_TRY(mspace_step_device(c, sb, gam2, diag_b, max_iter, multi, &stt));
                        sb.iters = (int)stt.iters; sb.converged = (int)stt.conv; sb.onsager = stt.ons;
                        sb.n_relres = (int)stt.nrel; sb.rel_err = stt.rel; sb.active = stt.active != 0.0;
                    } else {
                        CgSys* one[1] = {&sb};
                        MIX_TRY(cg_consume_all(c, one, 1, gam2, diag_b, max_iter, multi));
                    }
                    b_post();
                    continue;
                }
                if (a_phase == 2) { a_phase = 3; continue; }                        // A^T mu_a done
                if (a_phase == 0 || !dev) gvk::axpby(s, ha.dst, tau, ha.dst, gam2, ha.src, n);   // Q_A src complete
                if (a_phase == 0) {                                                // r = v - Q mu0 (:71-73)
                    gvk::axpby(s, r, 1.0, v_a->d, -1.0, r, n);
                    MIX_TRY(a_init_scalars());
                    a_phase = 1;
                    if (max_iter > 0) a_post(p, d); else a_finish();
                    continue;
                }
                if (dev) {                                                         // one CG step (:86-120), scalars on the device
                    CgxStatus stt;
                    MIX_TRY(aat_step_device(c, mu, p, r, d, z, DG->d, tau, gam2, max_iter, relres_a != nullptr, &stt, at_acc, MA->d));
                    a_iters = (int)stt.iters; a_rel = stt.rel; a_nrel = (int)stt.nrel;
                    if (stt.active == 0.0) { a_conv = (int)stt.conv; a_finish(); }
                    else a_post(p, d);
                    continue;
                }
                const int i = a_iters++;                                           // one CG step (:86-120)
                const double* xs[1] = {d};
                const double* ys[1] = {p};
                arm_scalars(c);
                gvk::dots(s, 1, xs, ys, n, c->red_partial, c->red_out);
                double dp, sc[2];
                MIX_TRY(read_scalars(c, 1, &dp));
                const double alpha = a_rz / dp;
                gvk::axpby(s, mu, 1.0, mu, alpha, p, n);
                if (at_acc) gvk::axpby(s, at_acc, 1.0, at_acc, alpha, MA->d, M);       // A^T mu += alpha A^T p
                arm_scalars(c);
                gvk::cg_step_b_diag(s, r, d, alpha, DG->d, z, n, c->red_partial, c->red_out);
                MIX_TRY(read_scalars(c, 2, sc));
                const double beta = sc[0] / a_rz;
                a_rz = sc[0];
                gvk::axpby(s, p, 1.0, z, beta, p, n);
                a_rel = sqrt(sc[1] / a_vn2);
                if (relres_a) relres_a[i] = a_rel;
                a_nrel = i + 1;
                if (a_rel < 1e-4) { a_conv = 1; a_finish(); }
                else if (a_iters >= max_iter) a_finish();
                else a_post(p, d);
            }
        }
        if (dev) {        // residual traces were written on the device
            // (every status has been read: the entries are in host memory)
            if (relres_a && a_nrel > 0) memcpy(relres_a, c->cgx_rel_h, sizeof(double) * (size_t)a_nrel);
            if (relres_b && sb.n_relres > 0) memcpy(relres_b, c->cgx_rel_h + c->cgx_relcap, sizeof(double) * (size_t)sb.n_relres);
        }
        if (aat_mu_a) {   // Q_A mu_a = v_a - r  =>  A A^T mu_a = (v_a - r - gam2 mu_a) / tau
            gvk::axpby(s, aat_mu_a->d, 1.0 / tau, v_a->d, -1.0 / tau, r, n);
            gvk::axpby(s, aat_mu_a->d, 1.0, aat_mu_a->d, -gam2 / tau, mu, n);
        }
        if (ata_mu_b) {
            gvk::axpby(s, ata_mu_b->d, 1.0 / tau, sb.v, -1.0 / tau, sb.r, M);
            gvk::axpby(s, ata_mu_b->d, 1.0, ata_mu_b->d, -gam2 / tau, sb.mu, M);
        }
        MIX_HIP(hipGetLastError());
    }
done:
#undef MIX_TRY
#undef MIX_HIP
    if (st_a) {
        st_a->iters = a_iters; st_a->converged = a_conv; st_a->rel_res = a_rel; st_a->onsager = 0; st_a->n_relres = a_nrel;
        st_a->n_ax = (int)(c->cnt.n_ax - ax0);
        st_a->n_atx = (int)(c->cnt.n_atx - atx0);
    }
    cg_fill_stats(sb, st_b);
    if (st_b) { st_b->n_ax = (int)(c->cnt.n_ax - ax0); st_b->n_atx = (int)(c->cnt.n_atx - atx0); }
    if (rc) (void)hipStreamSynchronize(s);
    cleanup();          // (views into the context's slab: nothing is released, so the host need not wait for the last kernels)
    return rc;
}

}  // extern "C"
