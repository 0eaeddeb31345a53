// gv_mfma.h -- host-side plan and launchers of the fixed-point i8 MFMA family (gv_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace gvm {

struct Decomp {
    int ks = 1;          // uniform: K-segments per quad (workgroups per group of 4 row groups)
    int64_t skL = 0;     // > 0: balanced decomposition, cells (quad x K-block) per workgroup; 0: uniform K-split
    int64_t piv = 0;     // balanced only, > 0: hybrid -- the first piv quads go one per workgroup over the whole K range (in
                         // step, like a uniform split with ks = 1) and only the remaining quads are cut into ranges of skL cells
    int prio = 0;        // 1: waves lower their issue priority as they progress (k_mfma_matvec)
    float taper = 0.f;   // uniform split only: K-segment j is (1 + taper (ks-1-2j)/(ks-1)) times the mean length, so the
                         // workgroups dispatched last (the last segment) are the shortest and the launch's tail is short
    float geo = 0.f;     // uniform split only, 0 < geo < 1: K-segment j is geo^j times segment 0 (big first, geometrically
                         // smaller behind): the dispatcher hands the short workgroups out as slots free up, which evens out
                         // the 10-20 % spread in the time equal shares take (takes precedence over taper)
    int occ = 0;         // workgroups a CU may hold: 0 / 3 = what the registers allow (three), 2 = two (the launch reserves dynamic
                         // LDS to that end).  Fewer resident workgroups keep fewer K ranges and rows open at a time: the two-vector
                         // Ax of the 8-GPU shard on the tile layout streams 4 % faster with 512 than with 768 (profiles/r6_launch_dist_shard.txt),
                         // the 100 GB headline kernels 2-6 % slower -- one more thing the tuner measures
    float xskew = 0.f;   // uniform split, ks >= 2: workgroups are dealt to the eight XCDs round-robin by block index and four of the XCDs
                         // of an MI355X -- those that get the ODD block indices (hardware XCC ids 0, 2, 4, 6) -- finish equal shares
                         // 4-6 % earlier than the other four (per-workgroup clocks, every box and round: profiles/r6_shard_wgtime.txt),
                         // then idle for the rest of the launch.  The segments of a quad alternate between the two kinds; those with an
                         // odd block index are made 1 + xskew, the others 1 - xskew times their nominal length.  Headline Ax: 15.09 ->
                         // 14.90 ms at 0.025 (profiles/r6_xcd_skew.txt); the opposite sign loses as much.
};
// dynamic LDS bytes that cap a CU at d.occ workgroups of the streaming kernels (160 KiB per CU, <= 8 KiB static per workgroup)
inline unsigned lds_pad_of(const Decomp& d) { return d.occ == 2 ? 60000u : 0u; }

// workgroups of a launch over nq quads x nkb K-blocks, and pieces (int32 partial sums per row) the epilogues add up at most
inline int64_t piv_of(const Decomp& d, int64_t nq) { return d.skL > 0 && d.piv > 0 ? (d.piv < nq ? d.piv : nq) : 0; }
inline int64_t grid_of(const Decomp& d, int64_t nq, int64_t nkb) {
    if (d.skL <= 0) return nq * d.ks;
    const int64_t piv = piv_of(d, nq);
    return piv + ((nq - piv) * nkb + d.skL - 1) / d.skL;
}
inline int64_t pieces_max(const Decomp& d, int64_t nkb) { return d.skL > 0 ? (nkb + d.skL - 1) / d.skL + 1 : d.ks; }

// ---- device-resident CG (gv_solvers.hip, cg_run_device) ------------------------------------------------------------
// State block of one CG system in device memory (doubles).  The kernels of a CG step read alpha / beta / the activity
// flags from it, so a step is enqueued without the host knowing the scalars of the previous one.
enum { ST_RZ = 0, ST_ALPHA, ST_BETA, ST_NORMV, ST_PREV_ONS, ST_ONS, ST_RELERR, ST_ACTIVE, ST_ITERS, ST_CONV, ST_NRELRES,
       ST_DENOISER, ST_STEPPED, ST_SIZE = 16 };
// What a pass over the shard needs to know when it is one half of a CG step.  Slot v = vector v of the pass.
struct CgHook {
    const int* go = nullptr;                 // the pass kernels return at once when *go == 0 (every system has finished)
    const double* state[2] = {nullptr, nullptr};   // slot v is CG system state[v]; NULL: not a CG system (a rider)
    // Ax side (k_prep_ax): the search direction is advanced on the way in, p <- z + beta p, when the system stepped
    double* p[2] = {nullptr, nullptr};
    const double* z[2] = {nullptr, nullptr};
    // ATx side (k_prep_atx): the same for an N-space system whose search direction is the operand of slot v (gv_cg_solve_aat2w)
    double* pn[2] = {nullptr, nullptr};
    const double* zn[2] = {nullptr, nullptr};
    // ATx side (k_fin_atx_dot): <out, addx> = <Q p, p>, block partials, then gvk::finalize -> dot_out[v][0]
    // Rider (gv_cg_extras.ride_x): while *ride == 1 and exactly one of the two systems has finished, the finished system's
    // slot of a two-vector Ax pass carries alt_x instead (decided on the device: the host learns of a finished system one
    // step late); the product lands in that slot's output, k_ride_copy moves it out and k_cgx_decide sets *ride = 2.
    const int* ride = nullptr;
    const double* alt_x = nullptr;
    double* ride_out = nullptr;              // != NULL: the Ax epilogue writes the rider's product there itself (no k_ride_copy)
    double* dot_part[2] = {nullptr, nullptr};
    double* dot_out[2] = {nullptr, nullptr};
    // Ax side (k_fin_ax), slot v closing an application of tau A A^T + gam2 I to dq_p[v] (gv_cg_solve_aat2w, one rank, vectors of at
    // most RED_BLOCKS * 256 entries): out = dq_tau * product + dq_gam2 * dq_p and the block partials of <out, dq_p> in dq_part[v],
    // while the system state[v] is running -- gvk::aat_step is then told that its k_aat_dq has been done (dq_done)
    const double* dq_p[2] = {nullptr, nullptr};
    double* dq_part[2] = {nullptr, nullptr};
    double dq_tau = 0.0, dq_gam2 = 0.0;
    bool dot_self = false;                   // the consumer adds the block partials of <out, addx> up itself: no gvk::finalize launch
};

struct Plan {
    int64_t M = 0, N = 0;
    int64_t nrg_m = 0, nkb_m = 0;   // stripes_m: row groups of 64 markers x K-blocks of 256 individuals
    int64_t nrg_n = 0, nkb_n = 0;   // stripes_n: row groups of 64 individuals x K-blocks of 256 markers
    // work decomposition of the streaming kernel, per kernel class: dm[0] ATx (MODE 0), dm[1] two-vector ATx / p-value sums
    // (MODE 2), dn[0] Ax and the people sums (MODE 1, 4), dn[1] two-vector Ax (MODE 3)
    Decomp dm[2], dn[2];
    void* stripes_m = nullptr;
    void* stripes_n = nullptr;
    // layout 1 ("tile", gv_set_layout(.., 2)): ONE resident re-encoding, 64 markers x 256 individuals per 4 KiB super-block,
    // serves both products (gv_mfma.hip).  Then nrg_m / nkb_m describe it (marker groups x individual blocks) and the Ax
    // side walks it transposed: nrg_n = nkb_m row groups of rows_n = 256 individuals, nkb_n = nrg_m K-steps of 64 markers.
    void* tiles = nullptr;
    int layout = 0;
    int rows_n = 64;                // rows per row group on the Ax side
    int64_t rstride_n = 0;          // tile layout, Ax side: row groups per K-step in memory when a launch covers a sub-range (0: nrg_n)
    void* dig0 = nullptr;           // digit buffers, max(nkb_m, nkb_n) * 2048 bytes each
    void* dig1 = nullptr;
    double* cv = nullptr;           // M doubles: c = msig * x
    double* ev = nullptr;           // M doubles: e = (mave - 3) * c
    double* cv2 = nullptr;          // the same for the second vector of a two-vector Ax
    double* ev2 = nullptr;
    double* scal = nullptr;         // 2 x 4 doubles: amax, sum, 2^(54-e), 2^(e-54) (second set: marker_sums2's p2)
    int32_t* partial = nullptr;     // per-(K-split, plane, row) digit sums
    size_t partial_bytes = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;   // when set: recorded around the matvec kernel launch (roofline timing)
};

void stripes_m_chunk(hipStream_t s, const uint8_t* raw, int64_t pitch, int64_t mc, int64_t N, void* stripes,
                     int64_t rg0, int64_t nkb);
void stripes_n_chunk(hipStream_t s, const uint8_t* raw, int64_t pitch, int64_t mc, int64_t N, void* stripes,
                     int64_t kb0, int64_t nkb, int64_t nrg_n);
void tile_chunk(hipStream_t s, const uint8_t* raw, int64_t pitch, int64_t mc, int64_t N, void* tiles, int64_t rg0, int64_t nkb);
void stats_from_tiles(hipStream_t s, const void* tiles, const uint32_t* mask2, int64_t M, int64_t nrg, int64_t nkb, int64_t P4,
                      double nonas, double alpha_scale, double* mave, double* msig, uint32_t* counts);
// counts (may be NULL): 3 per marker = present individuals with a = 2, 1, 0
void stats_from_stripes(hipStream_t s, const void* stripes_m, const uint32_t* mask2, int64_t M, int64_t nkb,
                        int64_t P4, double nonas, double alpha_scale, double* mave, double* msig, uint32_t* counts);
// The p-value pass (data::pvals_calc / pvals_calc_LOCO): p = mask (y - z1) + add (add may be NULL) and p^2 go through ONE pass over
// the marker-major layout, the per-marker regression test runs in the epilogue: pvals[m] for every local marker (chrom == NULL) or
// for the markers of chromosome ch.  pa / pb: npad doubles of scratch each (they receive p and p^2).
struct PvArgs {
    const uint32_t* cnt;      // 3 per marker: present individuals with a = 2, 1, 0 (marker statistics)
    const double* xself;      // leave-one-out: the marker's own effect x1_hat[m] * self_scale is added back; NULL: none
    double self_scale;
    const int* chrom;         // LOCO: chromosome of every marker, or NULL
    int ch;
};
void marker_pvals(hipStream_t s, const Plan& pl, const double* y, const double* z1, const double* add, const uint32_t* mask2,
                  int64_t npad, const double* mave, const double* msig, double* pa, double* pb, double* red_partial, const PvArgs& a,
                  double* pvals);
// out[M] = data::ATx(p); p has npad entries (zero at NA / pad slots)
// addx != NULL: out = tau * ATx(p) + gam2 * addx, the whole of vamp::lmmse_mult's epilogue (vamp.cpp:1112-1116)
void atx(hipStream_t s, const Plan& pl, const double* p, int64_t npad, const double* mave, const double* msig,
         double inv_sqrt_n, double* red_partial, double* out, const double* addx = nullptr, double tau = 1.0,
         double gam2 = 0.0, const CgHook* cg = nullptr);
// out[npad] = mask * (A~ x) * post   (post = 1/sqrt(N), or 1 when a cross-rank all-reduce follows)
void ax(hipStream_t s, const Plan& pl, const double* x, const double* mave, const double* msig, const uint32_t* mask2,
        int64_t npad, double post, double* red_partial, double* out, const CgHook* cg = nullptr);

// the two stages of ax / ax2, for callers that cut the product into individual-range chunks (row groups [rg0, rg1) of pl.rows_n)
void ax_prep(hipStream_t s, const Plan& pl, const double* xa, const double* xb, const double* mave, const double* msig,
             double* red_partial, const CgHook* cg = nullptr);
void ax_rows(hipStream_t s, const Plan& pl, int nv, int64_t rg0, int64_t rg1, const uint32_t* mask2, int64_t npad, double post,
             double* outa, double* outb, const CgHook* cg = nullptr);

int atx_dot_blocks(const Plan& pl);   // number of block partials of the fused <d, p> (CgHook::dot_part)

// two vectors per pass (the LMMSE and the Onsager CG of one VAMP iteration share the operator, vamp.cpp:593-596,:884)
void atx2(hipStream_t s, const Plan& pl, const double* pa, const double* pb, int64_t npad, const double* mave,
          const double* msig, double inv_sqrt_n, double* red_partial, double* outa, double* outb,
          const double* addxa = nullptr, const double* addxb = nullptr, double tau = 1.0, double gam2 = 0.0,
          const CgHook* cg = nullptr);
void ax_people(hipStream_t s, const Plan& pl, int kind, const double* mave, const double* msig, const uint32_t* mask2,
               int64_t npad, double* red_partial, double* out);
void ax2(hipStream_t s, const Plan& pl, const double* xa, const double* xb, const double* mave, const double* msig,
         const uint32_t* mask2, int64_t npad, double post, double* red_partial, double* outa, double* outb,
         const CgHook* cg = nullptr);

// kernel mode 2 (two-level fixed point, gv_mfma.hip MODE 5 / 6): the same products with ~108 bits below the vector's largest entry
// and exact zeros at missing genotypes, one two-vector-shaped pass each
void ax_wide(hipStream_t s, const Plan& pl, const double* x, const double* mave, const double* msig, const uint32_t* mask2,
             int64_t npad, double post, double* red_partial, double* out);
void atx_wide(hipStream_t s, const Plan& pl, const double* p, int64_t npad, const double* mave, const double* msig, double inv_sqrt_n,
              double* red_partial, double* out, const double* addx = nullptr, double tau = 1.0, double gam2 = 0.0);

}  // namespace gvm
