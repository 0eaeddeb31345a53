/* gvamp.h -- C ABI of libgvamp.so: the MI355X (gfx950) engine behind gVAMP's `class data` / `class vamp`
 * seam for the linear model.
 *
 * The reference has no plugin ABI; its seam is the C++ class `data` handed to `vamp` by pointer
 * (/root/reference/vamp.hpp:85-143, data.hpp:93-140).  Each entry point below names the reference
 * interface it replaces.  INTEGRATION.md shows the reference-side binding (a `data` subclass whose
 * Ax/ATx forward here).
 *
 * Conventions: every function returns 0 on success, non-zero on failure (message via gv_last_error);
 * no exceptions cross the ABI; all `const T*` / `T*` arguments are caller-owned HOST memory unless the
 * parameter is a gv_vec handle (device-resident fp64 vector owned by the context).  One context = one
 * marker shard on one GPU (the reference: one MPI rank, utilities.cpp:259-291).  Not thread-safe per
 * context; different contexts may be driven from different threads.
 *
 * There is NO CPU fallback: without a usable HIP device gv_create fails.
 */
#ifndef GVAMP_H
#define GVAMP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gv_ctx gv_ctx;
typedef struct gv_vec gv_vec;

/* Bumped whenever a struct of this header grows or a default changes, so that a binding compiled against an older header fails
 * its version check instead of overrunning a buffer.  History: 1 rounds 1-2; 2 round 3 (gv_decomp_info grew by whole_quads,
 * the context defaults became kernel mode 1 / no raw rows / layout auto) and round 4 (gv_ingest_info2, gv_set_expected_passes);
 * 3 round 5 (gv_cg_solve_aat2w takes v_a as the in/out vector it is; layout auto takes ONE resident layout unless the caller
 * announces a long run -- gv_set_expected_passes; gv_debug_force_multi); 4 round 6 (gv_decomp_info grew by wgs_per_cu and xcd_skew;
 * gv_set_kernel_mode accepts 2, the two-level fixed point).
 * A binding checks  gv_abi_version() == GV_ABI_VERSION  once, before anything else (INTEGRATION.md section B does). */
#define GV_ABI_VERSION 4

/* ---- context ------------------------------------------------------------------------------------ */
int gv_abi_version(void);
/* device: HIP ordinal.  Replaces nothing in the reference (it has no device boundary). */
int gv_create(int device, gv_ctx** out);
void gv_destroy(gv_ctx* ctx);
/* ctx may be NULL: returns the message of the last failed gv_create on this thread. */
const char* gv_last_error(const gv_ctx* ctx);
int gv_synchronize(gv_ctx* ctx);

/* ---- dataset: `data` constructors (data.cpp:30-113) ------------------------------------------------
 * N individuals, M local markers, Mt total markers, S first local marker (divide_work, utilities.cpp:259). */
int gv_set_dims(gv_ctx* ctx, int64_t N, int64_t M, int64_t Mt, int64_t S);
int64_t gv_mbytes(const gv_ctx* ctx);  /* ceil(N/4), data.cpp:40 */
/* bed: M*mbytes bytes, marker-major PLINK 2-bit, WITHOUT the 3 magic bytes: what read_genotype_data
 * (data.cpp:201-234) leaves in bed_data.  Copied to HBM (row pitch padded to 64 B) and kept resident. */
int gv_upload_bed(gv_ctx* ctx, const uint8_t* bed, size_t nbytes);
/* The same from a file, streamed through a bounded pinned buffer (host memory stays O(8192 * mbytes) however large
 * the shard): M*mbytes bytes starting at byte `offset` of `path` -- offset = 3 + S*mbytes for a PLINK .bed
 * (data.cpp:215).  The magic bytes are skipped, not validated, as in the reference. */
int gv_upload_bed_file(gv_ctx* ctx, const char* path, int64_t offset);
/* Synthetic shard generated on the device (bench / tests; SURVEY 8d recipe with an integer hash so that
 * gvamp_amd.synth.synth_bed() reproduces it bit for bit on the host).  miss_ppm: missing rate in 1e-6. */
int gv_synth_bed(gv_ctx* ctx, uint64_t seed, uint32_t miss_ppm);
/* The same with linkage disequilibrium: inside every block of ld_block consecutive (global) markers a genotype copies, with
 * probability ld_ppm / 1e6, the block's per-individual latent draw instead of its own -- block-correlated columns, as real
 * genotypes have them (the LMMSE CG then needs tens of steps instead of 4-5).  ld_block = 0: gv_synth_bed. */
int gv_synth_bed_ld(gv_ctx* ctx, uint64_t seed, uint32_t miss_ppm, uint32_t ld_block, uint32_t ld_ppm);
/* The PLINK rows back from HBM: only when the raw row layout is resident, i.e. gv_set_layout(ctx, 1, ..) was called before
 * the ingest (not the default). */
int gv_download_bed(gv_ctx* ctx, uint8_t* bed, size_t nbytes);
/* mask4: mbytes nibbles (data.hpp:36; bit k of mask4[j] = individual 4j+k has a phenotype and 4j+k < N);
 * NULL = every individual present (vector-phenotype ctor, data.cpp:86-100).  nonas: data.cpp:100/:150. */
int gv_set_mask(gv_ctx* ctx, const uint8_t* mask4, int64_t nonas);
/* compute_markers_statistics (data.cpp:392-546, scalar-path semantics :451-484). */
int gv_marker_stats(gv_ctx* ctx, double alpha_scale);
int gv_get_marker_stats(gv_ctx* ctx, double* mave, double* msig); /* M doubles each (data.hpp:95-98) */

/* ---- matvecs with the reference's host signatures ---------------------------------------------------
 * data::Ax (data.cpp:848-1009): x[M] -> out[4*mbytes], summed over ranks (if a communicator is attached),
 * masked, scaled by 1/sqrt(N).  data::ATx (data.cpp:810-835): p[4*mbytes] -> out[M], scaled by 1/sqrt(N). */
int gv_ax(gv_ctx* ctx, const double* x, double* out);
int gv_atx(gv_ctx* ctx, const double* p, double* out);

/* Resident HBM layouts built at the next gv_upload_bed / gv_synth_bed: raw_rows = the PLINK rows (pitch-padded;
 * needed by kernel mode 0 and gv_download_bed); stripes selects the re-encoded 2-bit layout of kernel mode 1:
 *   0 none, 1 two stripe sets (marker-major for ATx, individual-major for Ax: 2 x M*N/4 bytes resident),
 *   2 ONE tile layout that serves both products (M*N/4 bytes resident; bit-identical results),
 *   3 auto: the tile layout, unless the caller announced a long run (gv_set_expected_passes >= 1000) and two stripe sets (their
 *     ATx is 2-5 % faster) fit the free HBM at ingest.
 * Default: raw_rows = 0, stripes = 3 -- what bench.py measures and what a binding that never calls this gets (INTEGRATION.md
 * section B): the rows stream through a chunk buffer at ingest and only the re-encoded layout stays resident, so a 100 GB
 * shard occupies 100 GB (or 200 GB) of HBM, not 300.  raw_rows = 1 is needed by kernel mode 0 and gv_download_bed only. */
int gv_set_layout(gv_ctx* ctx, int raw_rows, int stripes);
int gv_get_layout(const gv_ctx* ctx);   /* the re-encoded layout resident now: 0 none, 1 two stripe sets, 2 tile layout */
/* kernel family for Ax/ATx: 1 (default) = i8 MFMA fixed-point kernels on the re-encoded layout (0.8 of the HBM roofline,
 * results within 2e-14 of fp64 sums, bit-reproducible); 0 = fp64 VALU kernels on the raw rows (parity anchor, 4-9 % of the
 * roofline; needs gv_set_layout(ctx, 1, ..) before ingest); 2 = two-level fixed point on the re-encoded layout (below). */
/* Accuracy contract of kernel mode 1.  A vector enters a product in fixed point with ONE exponent for the whole vector
 * (|q| < 2^54 relative to its largest entry); the product itself is exact integer arithmetic with one final rounding.  Hence
 *   Ax : |out[n] - exact[n]| <= M * 2^-50 * max_i |msig[i] x[i]| / sqrt(N),
 *   ATx: |out[m] - exact[m]| <= N * 2^-50 * msig[m] * max_n |p[n]| / sqrt(N)
 * (worst case; typical errors are sqrt(M) resp. sqrt(N) times smaller), i.e. an ABSOLUTE error relative to the vector's largest
 * entry -- below the error of fp64 summation in relative l2 (measured <= 2e-14 against the fp64 oracle over 60 decades of overall
 * scale), but not relative to each output entry: an entry more than 2^54 below the vector's maximum is flushed to zero, and an
 * output that does not see the large entries (Ax at an individual whose genotype is MISSING at the one marker carrying a huge
 * effect) is accurate to the bound above, not to its own last bits as the reference's fp64 sums are there.  The vectors of a VAMP
 * run span a few decades and never come near this; input with more than ~2^40 of dynamic range that needs per-entry relative
 * accuracy should use kernel mode 2 (below).  tests/test_gpu_matvec.py asserts both bounds on such input.
 * Kernel mode 2 (ABI 4) = the fast remedy: TWO-LEVEL fixed point on the same resident layout.  The vector is quantised as in mode 1
 * (the head), the exact fp64 residual v - head is quantised once more with the exponent 54 below (v = head + residual to within
 * 2^-109 of the largest entry: ~108 bits instead of 54), and head and residual travel as the two slots of ONE two-vector pass; the
 * planes of that pass give a missing genotype an exact zero (a' = a if present else 0; b = present), as the reference's table does
 * (data.cpp:951-988), instead of "3 c + (mave - 3) c - mave c", which cancels only to rounding.  An output entry that does not see
 * the huge entries is then accurate to its own magnitude (the adversarial case above: 1e-9 ... 1e-3 relative in mode 1, < 1e-12 in
 * mode 2) for up to ~2^55 of in-vector dynamic range.  Cost: one two-vector pass per product (1.0-1.1 x a one-vector pass); the
 * two-vector entry points run two such passes; the solvers take the host-driven loops (the device-resident CG and GV_OVERLAP are
 * mode-1 machinery); marker / people statistics and p-values are those of mode 1 (exact counts; their sums are not affected).
 * Results are bit-reproducible and independent of layout and decomposition, as in mode 1 -- but not bit-equal to mode 1. */
int gv_set_kernel_mode(gv_ctx* ctx, int mode);
int gv_get_kernel_mode(const gv_ctx* ctx);

/* ---- device-resident vectors (the std::vector<double> temporaries of vamp.cpp live here) ------------ */
enum { GV_SPACE_M = 0, GV_SPACE_N = 1 }; /* length M, or length 4*mbytes (padded internally) */
int gv_vec_alloc(gv_ctx* ctx, int space, gv_vec** out);
void gv_vec_free(gv_ctx* ctx, gv_vec* v);
int64_t gv_vec_len(const gv_vec* v); /* logical length: M or 4*mbytes */
int gv_vec_upload(gv_ctx* ctx, gv_vec* v, const double* src);
int gv_vec_download(gv_ctx* ctx, const gv_vec* v, double* dst);
int gv_vec_fill(gv_ctx* ctx, gv_vec* v, double value);
int gv_vec_copy(gv_ctx* ctx, gv_vec* dst, const gv_vec* src);
/* out = a*x + b*y (y may be NULL when b == 0; out may alias x or y) */
int gv_vec_axpby(gv_ctx* ctx, gv_vec* out, double a, const gv_vec* x, double b, const gv_vec* y);
/* out = x .* y element-wise (out may alias x or y): the freeze mask of vamp.cpp:308,:353 applied on the device */
int gv_vec_mul(gv_ctx* ctx, gv_vec* out, const gv_vec* x, const gv_vec* y);
/* inner_prod (utilities.cpp:190-210): sync != 0 adds the cross-rank all-reduce of the scalar. */
int gv_vec_dot(gv_ctx* ctx, const gv_vec* x, const gv_vec* y, int sync, double* out);
/* several dots in one pass and ONE all-reduce: out[k] = <x[k], y[k]> */
int gv_vec_dots(gv_ctx* ctx, int n, const gv_vec* const* x, const gv_vec* const* y, int sync, double* out);
/* The scalars one VAMP iteration reads after its solves (vamp.cpp:631, :686-693, :892-927, :741-749, :1295-1317), in one launch
 * and one read-back: out[k] = <xa - xb, ya - yb> (xb / yb may be NULL: the plain vector); pairs of either space may be mixed;
 * sync != 0 adds the cross-rank all-reduce of that scalar.  Each scalar is bit-identical to gv_vec_axpby(t, 1, xa, -1, xb) ...
 * gv_vec_dot(t, u): the difference is rounded once, the sum runs in the same order. */
typedef struct gv_dot_spec {
    const gv_vec* xa; const gv_vec* xb;
    const gv_vec* ya; const gv_vec* yb;
    int sync;
} gv_dot_spec;
int gv_vec_dots_ex(gv_ctx* ctx, int n, const gv_dot_spec* spec, double* out);
int gv_ax_dev(gv_ctx* ctx, const gv_vec* x, gv_vec* out);  /* data::Ax on handles */
int gv_atx_dev(gv_ctx* ctx, const gv_vec* p, gv_vec* out); /* data::ATx on handles */
/* Two vectors per pass over the shard (kernel mode 1; two single passes otherwise): outa = A xa, outb = A xb, and
 * outa = A^T pa, outb = A^T pb.  Results are bit-identical to the one-vector calls (exact integer accumulation). */
int gv_ax2_dev(gv_ctx* ctx, const gv_vec* xa, const gv_vec* xb, gv_vec* outa, gv_vec* outb);
int gv_atx2_dev(gv_ctx* ctx, const gv_vec* pa, const gv_vec* pb, gv_vec* outa, gv_vec* outb);
/* phenotype y (length N) -> N-space handle with NA / pad slots zeroed: data::filter_pheno (data.cpp:1065-1079) */
int gv_set_phen(gv_ctx* ctx, gv_vec* y_out, const double* y_host);

/* ---- solver: vamp::lmmse_mult (vamp.cpp:1074-1118), vamp::precondCG_solver (vamp.cpp:1130-1229) ------ */
int gv_lmmse_mult(gv_ctx* ctx, const gv_vec* v, double tau, double gam2, gv_vec* out);
typedef struct {
    int iters;           /* CG steps executed */
    int converged;       /* 1 if a stopping rule fired before max_iter */
    double rel_res;      /* ||r|| / ||v|| at exit (vamp.cpp:1215) */
    double onsager;      /* gam2 * <v, mu> at exit (denoiser == 0 only, vamp.cpp:1176) */
    int n_ax, n_atx;     /* matvec calls made */
    int n_relres;        /* entries written to relres (an Onsager-rule exit skips the last residual update) */
} gv_cg_stats;
/* mu_start may be NULL (zeros).  denoiser: 1 = LMMSE solve, 0 = Onsager probe solve (extra stopping rule).
 * relres (may be NULL): max_iter doubles receiving ||r||/||v|| after every step. */
int gv_cg_solve(gv_ctx* ctx, const gv_vec* v, const gv_vec* mu_start, double tau, double gam2, int denoiser,
                int max_iter, gv_vec* mu_out, gv_cg_stats* stats, double* relres);

/* The LMMSE solve (v_a, optional warm start, denoiser = 1 rules) and the Onsager probe solve (v_b, zero start, denoiser = 0
 * rules) of one VAMP iteration run in lock-step on the shared operator (tau, gam2): every round applies the operator to
 * the pending direction of each still-active solve in ONE two-vector Ax + ATx pair.  Per solve the iterates, stopping
 * rules and results are those of gv_cg_solve. */
int gv_cg_solve2(gv_ctx* ctx, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                 int max_iter, gv_vec* mu_a, gv_vec* mu_b, gv_cg_stats* stats_a, gv_cg_stats* stats_b,
                 double* relres_a, double* relres_b);

/* gv_cg_solve2 plus by-products that cost no further pass over the shard.  Every member may be NULL (= not wanted):
 *   ride_x / ride_out : ride_out = data::Ax(ride_x) (data.cpp:848) for an M-vector unrelated to the solves -- e.g.
 *                       z1 = A x1_hat of vamp.cpp:429 -- computed in the free slot of the first round in which only one
 *                       solve is still active (a two-vector pass costs what a one-vector pass costs), else by one pass of
 *                       its own after the solves.  Bit-identical to gv_ax_dev.
 *   a_mu_a   (N-space): A mu_a, accumulated from the products A p_k that CG forms anyway
 *                       (A mu_a = A mu_start + sum_k alpha_k A p_k): the A x2_hat of vamp.cpp:897 without its Ax.
 *   ata_mu_b (M-space): A^T A mu_b = (v_b - r_b - gam2 mu_b) / tau from the final residual of solve b: the
 *                       A^T (A invQ u) of vamp.cpp:913-914 without its Ax and ATx.
 * a_mu_a and ata_mu_b are algebraic identities of the CG recurrences: they agree with the explicit products to rounding
 * (~1e-13 relative), not bit for bit. */
typedef struct gv_cg_extras {
    const gv_vec* ride_x;
    gv_vec* ride_out;
    gv_vec* a_mu_a;
    gv_vec* ata_mu_b;
} gv_cg_extras;
int gv_cg_solve2x(gv_ctx* ctx, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                  int max_iter, gv_vec* mu_a, gv_vec* mu_b, gv_cg_stats* stats_a, gv_cg_stats* stats_b,
                  double* relres_a, double* relres_b, const gv_cg_extras* extras);
/* gv_cg_solve2x whose warm start costs no pass.  precondCG_solver opens with r = v - Q mu_start (vamp.cpp:1142-1145: one Ax
 * and one ATx); when mu_start_a is the mu_a of the PREVIOUS call, that call's final residual already holds the product:
 * Q' mu = v' - r'  =>  A^T A mu = (v' - r' - gam2' mu) / tau'.  Every member may be NULL:
 *   ata_mu_a       (M-space, out): A^T A mu_a of THIS solve from its final residual -- hand it to the next call;
 *   ata_mu_start_a (M-space, in) : A^T A mu_start_a (the previous call's ata_mu_a; may be the same handle as ata_mu_a).
 *                                  r = v_a - tau * ata_mu_start_a - gam2 * mu_start_a replaces the operator application;
 *   a_mu_start_a   (N-space, in) : A mu_start_a, required with ata_mu_start_a when extras->a_mu_a is wanted (its
 *                                  accumulation starts there; may be the same handle as extras->a_mu_a, i.e. the previous
 *                                  call's a_mu_a left in place).
 * An identity of the CG recurrences like a_mu_a / ata_mu_b: equal to the explicit products to rounding, not bit for bit;
 * the rounding of successive calls adds up (no call re-anchors on an explicit product), ~1e-16 relative per CG step. */
typedef struct gv_cg_warm {
    const gv_vec* ata_mu_start_a;
    const gv_vec* a_mu_start_a;
    gv_vec* ata_mu_a;
    /* Solve b starts from zero, so its first step applies the operator to v_b / diag.  When v_b is the same vector call after call
     * (the Onsager probe of vamp.cpp:875 is re-seeded identically every iteration), A^T A v_b never changes:
     *   ata_v_b (M-space): with have_ata_v_b == 0 an OUTPUT -- A^T A v_b, taken from solve b's first application (written when
     *                      max_iter > 0); with have_ata_v_b != 0 an INPUT -- that first application becomes
     *                      (tau / diag) * ata_v_b + gam2 * v_b / diag and costs no pass, so solve b is one pass pair shorter.
     * The caller owns the invariant "same v_b as when ata_v_b was written".  Equal to the explicit product to rounding. */
    gv_vec* ata_v_b;
    int have_ata_v_b;
} gv_cg_warm;
int gv_cg_solve2w(gv_ctx* ctx, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                  int max_iter, gv_vec* mu_a, gv_vec* mu_b, gv_cg_stats* stats_a, gv_cg_stats* stats_b,
                  double* relres_a, double* relres_b, const gv_cg_extras* extras, const gv_cg_warm* warm);

/* ---- denoiser side (fused element-wise kernels) ------------------------------------------------------
 * vamp::g1 / g1d over a vector (vamp.cpp:805-869; loops :292-310): x1 = g1(r1), sums[0] = sum g1d(r1) (local),
 * sums[1] = sum (x1-r1)^2 (local).  vars already multiplied by N (vamp.cpp:154-155).  d_out may be NULL. */
int gv_denoise(gv_ctx* ctx, const gv_vec* r1, double gam1, const double* probs, const double* vars, int L,
               gv_vec* x1_out, gv_vec* d_out, double* sums2);
/* one E-step of vamp::updatePrior (vamp.cpp:953-1013): sums[0] = sum_i pi_i, sums[1+2j] = sum_i beta_ij pi_i,
 * sums[2+2j] = sum_i beta_ij (m_ij^2 + v_j) pi_i for j = 0..L-2 (local sums; caller all-reduces 1+2(L-1)). */
/* The same two with their sums taken over the ranks (the MPI_Allreduce of vamp.cpp:313 / :326 and of :990 / :1012-1013 folded in):
 * the device buffer is all-reduced in stream before the one read-back, where gv_denoise + gv_allreduce_host make a second round trip
 * (host -> device -> all-reduce -> host) per call -- about seven per VAMP iteration on a sharded job.  One rank: identical to the
 * local forms. */
int gv_denoise_global(gv_ctx* ctx, const gv_vec* r1, double gam1, const double* probs, const double* vars, int L,
                      gv_vec* x1_out, gv_vec* d_out, double* sums2);
int gv_prior_estep_global(gv_ctx* ctx, const gv_vec* r1, double gam1, double lambda, const double* omegas,
                          const double* vars, int L, double* sums);
int gv_prior_estep(gv_ctx* ctx, const gv_vec* r1, double gam1, double lambda, const double* omegas,
                   const double* vars, int L, double* sums);
/* ---- --model bin_class (vamp_probit.cpp): the z-side probit denoiser over the N individuals ------------------------
 * vamp::g1_bin_class / g1d_bin_class (vamp_probit.cpp:661-705; loops :335-352) with m_cov = 0: z1 = E[z | y, cavity
 * N(p1, 1/tau1)], sums2[0] = sum_n g1d_bin_class, sums2[1] = sum_n (z1 - p1)^2.  y holds 0 / 1 (N-space handle). */
int gv_probit_denoise(gv_ctx* ctx, const gv_vec* p1, const gv_vec* y, double tau1, double probit_var, gv_vec* z1_out,
                      double* sums2);
/* the same with covariates (--C > 0): m_cov[i] = <Z[i], cov_eff> (vamp_probit.cpp:347,:364) shifts the argument of the
 * probit likelihood, c = (p1 + m_cov) / sqrt(probit_var + 1/tau1).  m_cov: N-space handle or NULL. */
int gv_probit_denoise_cov(gv_ctx* ctx, const gv_vec* p1, const gv_vec* y, const gv_vec* m_cov, double tau1,
                          double probit_var, gv_vec* z1_out, double* sums2);

/* ---- --use-XXT-denoiser 1 (vamp.cpp:169-170, :599-606; denoiserXXT.cpp): LMMSE through CG in N-space, matrix-free --
 * data::compute_people_statistics (data.cpp:558-716): per-individual mean, inverse std and count of the standardised
 * genotypes, all-reduced over ranks; kept on the device for gv_cg_solve_aat.  Host copies (4*mbytes doubles each) are
 * optional (NULL).  Kernel mode 0: three fp64 passes over the raw rows.  Kernel mode 1 (or no raw rows resident): four
 * exact fixed-point passes over stripes_n, the quadratic one on the a^2 plane of the 2-bit codes. */
int gv_people_stats(gv_ctx* ctx, double* mave_people, double* msig_people, double* numb_people);
/* vamp::CG_solverAAT (denoiserXXT.cpp:52-130): solves (tau A A^T + gam2 I) mu = v in N-space with the per-individual
 * diagonal preconditioner, stopping at ||r||/||v|| < 1e-4.  mu_start may be NULL (zeros). */
int gv_cg_solve_aat(gv_ctx* ctx, const gv_vec* v, const gv_vec* mu_start, double tau, double gam2, int max_iter,
                    gv_vec* mu_out, gv_cg_stats* stats, double* relres);
/* The N-space solve above (system a: v_a, mu_start_a, N-space) and the Onsager probe solve of the same iteration (system
 * b: gv_cg_solve with denoiser = 0 from a zero start, M-space; vamp.cpp:631 in the --use-XXT-denoiser branch) on shared
 * passes: Q_a = tau A A^T + gam2 I is an ATx followed by an Ax, Q_b = tau A^T A + gam2 I an Ax followed by an ATx, so run
 * half an application out of phase every pass over the shard (after the first) serves both through the two-vector kernels.
 * Per solve the iterates, stopping rules and results are those of the stand-alone calls (bit-identical).
 *   at_mu_a (M-space)           : A^T mu_a, the ATx of denoiserXXT.cpp:47-49 (x2_hat = r2 + gamw A^T mu_a), taken along too;
 *   aat_mu_a (N-space, or NULL) : A A^T mu_a = (v_a - r_a - gam2 mu_a) / tau from the final residual of solve a;
 *   ata_mu_b (M-space, or NULL) : A^T A mu_b likewise (gv_cg_extras) -- identities of the recurrences, equal to rounding. */
int gv_cg_solve_aat2(gv_ctx* ctx, const gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                     int max_iter, gv_vec* mu_a, gv_vec* at_mu_a, gv_vec* mu_b, gv_cg_stats* stats_a, gv_cg_stats* stats_b,
                     double* relres_a, double* relres_b, gv_vec* aat_mu_a, gv_vec* ata_mu_b);
/* The same with products the caller already holds.  Every member may be NULL / 0:
 *   aat_mu_start_a (N-space): A A^T mu_start_a as the previous call left it in aat_mu_a (may be that very handle);
 *                             r = v_a - tau * aat_mu_start_a - gam2 * mu_start_a then replaces the ATx + Ax pair that opens a
 *                             warm-started CG_solverAAT (denoiserXXT.cpp:76-78);
 *   accumulate_at_mu_a      : at_mu_a = A^T mu_a is accumulated from the A^T p_k every operator application starts with
 *                             (A^T mu_a = A^T mu_start_a + sum_k alpha_k A^T p_k) instead of by a closing ATx pass;
 *   at_mu_start_a (M-space) : A^T mu_start_a for that sum (the previous call's at_mu_a; may be the same handle as at_mu_a) --
 *                             required when aat_mu_start_a skips the opening application, taken from that application otherwise,
 *                             0 for a zero start.
 * Identities of the recurrences: equal to the explicit products to rounding, not bit for bit. */
typedef struct gv_aat_warm {
    const gv_vec* aat_mu_start_a;
    const gv_vec* at_mu_start_a;
    int accumulate_at_mu_a;
    gv_vec* ata_v_b;            /* as in gv_cg_warm: A^T A v_b of the zero-started M-space solve b, captured or handed in */
    int have_ata_v_b;
    /* pre_x / pre_out: the right-hand side of solve a is completed inside the call -- pre_out = A pre_x, then v_a <- v_a - pre_out
     * (v_a is an in/out argument of gv_cg_solve_aat2w for this reason): v = y - A r2 of denoiserXXT.cpp:40-42 with v_a = y, pre_x = r2.
     * That Ax shares its pass with the first half-application of solve b, which puts the two solves in phase from the start
     * (one pass fewer than forming v_a outside).  ride_x / ride_out: ride_out = A ride_x for an M-vector unrelated to the solves
     * (z1 = A x1_hat, vamp.cpp:429), taken in an Ax pass that has a slot free, else by a pass of its own at the end. */
    const gv_vec* pre_x;
    gv_vec* pre_out;
    const gv_vec* ride_x;
    gv_vec* ride_out;
    /* v_a <- v_a - pre_scale * pre_out (0 = 1).  With pre_x = x1_hat, pre_out = z1 and the caller's v_a = y + (gam1/gam2) A r1 the solve
     * gets its right-hand side y - A r2 from A r2 = (eta1/gam2) A x1_hat - (gam1/gam2) A r1 (r2 is that combination, vamp.cpp:485-486)
     * and the iteration's z1 = A x1_hat (vamp.cpp:429) for free: no pass for A r2 (host/vamp.cpp keeps A r1 up to date from A x2_hat). */
    double pre_scale;
} gv_aat_warm;
/* v_a: read only, unless warm->pre_x is set -- then it is updated in place (see pre_x above). */
int gv_cg_solve_aat2w(gv_ctx* ctx, gv_vec* v_a, const gv_vec* mu_start_a, const gv_vec* v_b, double tau, double gam2,
                      int max_iter, gv_vec* mu_a, gv_vec* at_mu_a, gv_vec* mu_b, gv_cg_stats* stats_a, gv_cg_stats* stats_b,
                      double* relres_a, double* relres_b, gv_vec* aat_mu_a, gv_vec* ata_mu_b, const gv_aat_warm* warm);

/* ---- association tests after the loop (vamp.cpp:761-776) ------------------------------------------------------
 * data::pvals_calc (data.cpp:1108-1226, one estimator): leave-one-out t-test p-value of every local marker,
 * pvals[M].  z1 = A x1_hat and y (filtered phenotype) are N-space handles, x1_hat the M-space estimate (the
 * sqrt(N)-scaled x1_hat of vamp.cpp, as passed at :765).  One pass over the genotype shard (two digit vectors, y_mod
 * and y_mod^2, share the 16 MFMA columns) + exact per-marker genotype counts; the regression test and its Student-t tail run in
 * the epilogue of that pass (k_fin_pvals, gv_pval_dev.h), only the p-values cross PCIe. */
int gv_pvals_loo(gv_ctx* ctx, const gv_vec* z1, const gv_vec* y, const gv_vec* x1_hat, double* pvals);
/* data::pvals_calc_LOCO (data.cpp:1235-1353): chrom[M] in 1..23 (read_chromosome_info, data.cpp:346-380); per
 * chromosome one Ax (with its cross-rank all-reduce) + one marker pass.  Markers of other chromosomes get 0. */
int gv_pvals_loco(gv_ctx* ctx, const gv_vec* z1, const gv_vec* y, const gv_vec* x1_hat, const int* chrom,
                  double* pvals);
/* The same, also returning the per-chromosome genetic predictors the reference dumps as <file>_LOCO_chr_<ch>.csv
 * (data.cpp:1276-1281): chrom_pred[(ch-1) * 4*mbytes + n] = (A x1_hat restricted to chromosome ch)[n], summed over ranks,
 * ch = 1..23 (zeros for a chromosome no rank holds).  chrom_pred: 23 * 4*mbytes host doubles, or NULL. */
int gv_pvals_loco_pred(gv_ctx* ctx, const gv_vec* z1, const gv_vec* y, const gv_vec* x1_hat, const int* chrom,
                       double* pvals, double* chrom_pred);

/* SUM all-reduce of n host doubles over the attached communicator (identity when none): MPI_Allreduce of
 * scalars in vamp.cpp:313,990,1012-1013 */
int gv_allreduce_host(gv_ctx* ctx, double* buf, int n);

/* ---- communicator: stands in for MPI_COMM_WORLD (data.cpp:928/:995; utilities.cpp:203) ---------------
 * RCCL over xGMI, one process per GPU.  Rank 0 calls gv_comm_unique_id and ships the 128 bytes to the
 * other ranks by any out-of-band channel (bench.py: torch.distributed broadcast). */
int gv_comm_unique_id(void* id128);
/* nranks == 1 with id128 == NULL attaches nothing; with an id a 1-rank RCCL communicator is created (self-test). */
int gv_comm_init(gv_ctx* ctx, int nranks, int rank, const void* id128);
/* In-process communicator for tests: `nranks` contexts of one process (one thread each, any GPUs) that name the same
 * `group` behave like nranks ranks; sums are taken in rank order on the host.  Every collective must be entered by
 * all members concurrently (from different threads). */
int gv_comm_init_local(gv_ctx* ctx, int group, int nranks, int rank);
/* Communicator over a caller-supplied transport: every collective of the context becomes
 *   device -> host copy, fn(user, buf, n) [in-place SUM of n host doubles over the nranks ranks; 0 = ok], host -> device copy.
 * For a host that already owns a communicator -- the reference's MPI_COMM_WORLD (`MPI_Allreduce(MPI_IN_PLACE, buf, n,
 * MPI_DOUBLE, MPI_SUM, ...)`, data.cpp:928) -- and for tests that run the ranks as processes sharing one GPU
 * (torch.distributed gloo).  Slower than RCCL (two PCIe hops per message); the sums must be identical on every rank. */
typedef int (*gv_allreduce_fn)(void* user, double* buf, size_t n);
int gv_comm_init_callback(gv_ctx* ctx, int nranks, int rank, gv_allreduce_fn fn, void* user);
/* One communicator per process: ctx joins the communicator `owner` already holds (RCCL, in-process or callback) instead of
 * building another one -- a process that keeps several `data` objects (main_real --run-mode both: a training and a test set)
 * has ONE MPI_COMM_WORLD in the reference, too.  The communicator lives until the last context sharing it is destroyed; both
 * contexts must be driven from the same host thread (their collectives are issued in program order). */
int gv_comm_share(gv_ctx* ctx, const gv_ctx* owner);
/* Overlap of the N-vector exchange of data::Ax with the decode (kernel mode 1, sharded jobs): tiles > 1 cuts the product into
 * that many chunks of individuals and all-reduces each slice on a side HIP stream while the next chunk decodes; 0 / 1 = one
 * message after the whole pass (default; also set by the environment variable GV_OVERLAP).  Bit-identical results. */
int gv_set_overlap(gv_ctx* ctx, int tiles);
/* TEST HOOK, never part of a production job.  A context of a ONE-rank job takes every branch a sharded job takes -- the N-vector
 * exchange inside data::Ax (data.cpp:928/:995), the packed scalar all-reduces (utilities.cpp:203), k_finalize + all-reduce instead of
 * the one-rank shortcuts of the device-resident CG, the side-stream exchange of gv_set_overlap -- with an exchange that is
 * asynchronous and in-stream like RCCL's (no host synchronisation, unlike the gv_comm_init_local / _callback transports):
 *   transport 1: loop-back kernels -- the message moves to scratch, the buffer is filled with NaNs, (delay_us later) it moves back;
 *                a consumer that is not stream-ordered behind its exchange reads NaNs;
 *   transport 2: ncclAllReduce on a 1-rank RCCL communicator (created here if the context holds none);   3: both, RCCL first;
 *   transport 0: back to the plain one-rank paths;   + 4: fault injection -- the overlapped exchange drops its closing event edge
 *                (the harness must then SEE wrong results: tests/test_gpu_forced_multi.py checks that it has teeth).
 * Sums over one rank are the identity, so every result must equal the plain one-rank run BIT FOR BIT (tests/test_gpu_forced_multi.py).
 * Also: environment variable GVAMP_FORCE_MULTI=<transport>[:<delay_us>] at gv_create (drivers, bench.py). */
int gv_debug_force_multi(gv_ctx* ctx, int transport, int delay_us);
int gv_comm_rank(const gv_ctx* ctx);
/* One process per GPU on a multi-socket node: restricts the calling thread -- and every thread it starts afterwards -- to the
 * CPUs of the NUMA node `device` hangs off (sysfs: the PCI device's numa_node, the node's cpulist), intersected with the CPUs it
 * may already use.  *numa_node_out (may be NULL) = that node, or -1 when nothing was changed (single-node host, topology hidden,
 * fewer than 8 CPUs in common, GVAMP_NUMA_BIND=0).  Call it before the first gv_create of the process; the drivers and bench.py do. */
int gv_bind_host_numa(int device, int* numa_node_out);
int gv_comm_size(const gv_ctx* ctx);

/* ---- instrumentation ---------------------------------------------------------------------------------- */
typedef struct {
    int64_t n_ax, n_atx;          /* kernel launches of each matvec since the last reset */
    double ms_ax, ms_atx;         /* HIP-event time spent in them (on the context's stream) */
    double ms_allreduce;          /* HIP-event time of the N-vector all-reduces */
    int64_t n_ax_kernel, n_atx_kernel;   /* timing == 2: launches of the dominant matvec kernel measured ... */
    double ms_ax_kernel, ms_atx_kernel;  /* ... and their summed HIP-event durations */
    int64_t n_ax_pass, n_atx_pass;       /* passes over the genotype shard (a two-vector product counts 2 in n_ax / n_atx
                                          * and 1 here) */
    int64_t n_allreduce;                 /* N-vector exchange steps timed into ms_allreduce (timing 1 or 2; sharded jobs) */
} gv_counters;
/* timing: 0 off; 1 brackets every whole matvec (prep + kernel + epilogue) with HIP events and synchronises per
 * call (development); 2 records event pairs around the dominant matvec kernel only, WITHOUT synchronising --
 * they are resolved in gv_get_counters (live roofline measurement inside a timed region). */
int gv_set_timing(gv_ctx* ctx, int timing);
int gv_get_counters(gv_ctx* ctx, gv_counters* out);
int gv_reset_counters(gv_ctx* ctx);
/* The work decomposition of the four streaming-kernel classes (0 ATx, 1 two-vector ATx, 2 Ax, 3 two-vector Ax) as
 * picked by the on-device tuning (or its cache / an override): what a bench line reports per rank. */
typedef struct {
    int ks;                  /* uniform K-split: segments per group of four row groups */
    int64_t balanced_cells;  /* > 0: balanced decomposition, cells per workgroup (ks unused) */
    int prio;                /* progress-based wave priority on */
    float taper;             /* uniform split: taper of the segment lengths */
    int tuned;               /* 1 once the pick has been made (measured or read from the cache) */
    int64_t whole_quads;     /* balanced only, > 0: hybrid -- that many groups of four row groups go one per workgroup over the
                              * whole K range, only the remaining ones are cut into ranges of balanced_cells cells */
    float geo;               /* uniform split, 0 < geo < 1: segment j is geo^j times segment 0 (big first; replaces taper) */
    int wgs_per_cu;          /* ABI 4: workgroups of the streaming kernel a CU holds: 3 (what the registers allow; 0 on input means
                              * this) or 2 (the launch reserves LDS to that end: faster on some 12.5 GB shapes, slower at 100 GB) */
    float xcd_skew;          /* ABI 4, uniform split with ks >= 2: segments whose workgroup lands on one of the four faster XCDs of the
                              * part (odd block index) are 1 + xcd_skew, the others 1 - xcd_skew times their nominal length (equal
                              * shares finish 4-6 % apart on an MI355X); 0 = equal, range -0.2 .. 0.2 */
} gv_decomp_info;
/* Wall time of the last ingest (gv_upload_bed / gv_upload_bed_file / gv_synth_bed), split into allocating the resident
 * layouts (hipMalloc of 100+ GB: the driver maps and wipes the pages; 0 when the buffers were reused) and filling them. */
int gv_ingest_info(gv_ctx* ctx, double* alloc_seconds, double* fill_seconds);
/* The same and more.  alloc_seconds = wall time until the resident layout was allocated AND the source prepared (the allocation
 * runs on a helper thread beside the pinned staging buffers, the chunk buffer and, for gv_upload_bed_file, the first two chunks
 * read from the file); overlap_seconds = how much of the allocation that preparation hid (wall = max, not sum). */
typedef struct {
    double alloc_seconds, fill_seconds, overlap_seconds;
    double resident_bytes;       /* genotype bytes resident after the ingest (one or two re-encoded layouts, raw rows if asked for) */
    int layout;                  /* gv_get_layout */
    int64_t expected_passes;     /* what gv_set_expected_passes said (0 = unknown) */
} gv_ingest_stats;
int gv_ingest_info2(gv_ctx* ctx, gv_ingest_stats* out);
/* Hint for gv_set_layout(.., 3) (auto), to be given before the ingest: how many ATx passes over the shard the caller expects to
 * make (a VAMP run: iterations x CG steps; 0 = unknown).  The second stripe set of layout 1 costs its bytes once more at ingest
 * (0.5-4.3 s per 100 GB, depending on whether the driver is still wiping freed memory) and makes every ATx pass ~3 % faster: it
 * pays for itself after 500-9 000 passes, whatever the shard size.  Auto builds the two sets only with a hint >= 1 000 (and room in
 * HBM); below that, and with no hint at all (ABI 3: a run that says nothing about its length is not assumed to be long), it takes
 * the one tile layout.  The drivers pass iterations x 12, bench.py the passes it is about to make. */
int gv_set_expected_passes(gv_ctx* ctx, int64_t passes);
int gv_get_decomp(gv_ctx* ctx, gv_decomp_info* out4);
/* Pins the decomposition of class cls (0 ATx, 1 two-vector ATx, 2 Ax, 3 two-vector Ax), e.g. one a deployment measured itself:
 * `tuned` is ignored, balanced_cells > 0 selects a balanced / hybrid grid (ks unused), else a uniform split of ks segments with
 * taper or geo.  Call it after the ingest; a first matvec that tunes (gv_tune_info source 1 / 2 / 4) overwrites it, one that has
 * already happened does not.  Refused when the decomposition needs more partial-sum pieces than the context holds room for.
 * Never changes a bit of output (exact integer accumulation). */
int gv_set_decomp(gv_ctx* ctx, int cls, const gv_decomp_info* in);
/* How the picks were made: *source = -1 not yet (the first matvec in kernel mode 1 makes them), 0 the cost model's first
 * candidate (tuning impossible), 1 measured on the device now (*seconds of wall time), 2 read from the cache an earlier run on
 * the same (device, N, M, kernel sources) left ($GV_TUNE_CACHE_DIR, $XDG_CACHE_HOME/gvamp_amd or ~/.cache/gvamp_amd;
 * GV_TUNE_CACHE=0 disables it), 3 fixed by an override, 4 taken from the table shipped with the library for the shapes of
 * BASELINE.json and the per-GPU shards of the headline job on an MI355X (gv_tune_builtin.h; GV_TUNE_BUILTIN=0 ignores it).
 * Picks never change results (exact integer accumulation). */
int gv_tune_info(gv_ctx* ctx, double* seconds, int* source);
/* device copy bandwidth probe: copies nbytes device->device `reps` times, returns GB/s (read+write bytes) */
int gv_copy_bandwidth(gv_ctx* ctx, size_t nbytes, int reps, double* gbps);
/* read-only stream probe with the access shape of the matvec kernels (every wave walks a contiguous run of 4 KiB blocks,
 * non-temporal 16-byte loads, nothing else): GB/s over the resident marker-major stripes when there are any, else over a
 * scratch buffer of nbytes.  The ceiling a streaming kernel can be held against on this part. */
int gv_read_bandwidth(gv_ctx* ctx, size_t nbytes, int reps, double* gbps);

#ifdef __cplusplus
}
#endif
#endif /* GVAMP_H */
