/* gvamp_host.h -- C entry points of libgvamp_host.so: the host-side C++ mirror of the reference's `vamp` / `data`
 * classes (gvamp_amd/csrc/host) driven on a shard that is already resident in a gv_ctx.  Used by bench.py (VAMP
 * iterations/s on synthetic UKB-scale shards, which have no .bed file) and by the full-run parity tests; the
 * executables gvamp_sim / gvamp_main_real are the reference-style drivers (sim.cpp, main_real.cpp). */
#ifndef GVAMP_HOST_H
#define GVAMP_HOST_H
#include "gvamp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* libgvamp_host.so has its own version: gvh_opts / gvh_iter grow with the host classes (2: reanchor_every, probe_product). */
#define GVH_ABI_VERSION 2
int gvh_abi_version(void);

typedef struct {                 /* the knobs vamp reads from Options (options.hpp:91-142) and its constructor */
    int iterations, CG_max_iter, EM_max_iter;
    double EM_err_thr, stop_criteria_thr, rho;
    int learn_vars;
    unsigned long seed;
    int use_lmmse_damp;
    double gam1, gamw;            /* sim.cpp:242-251: 1e-8, 2 ; main_real.cpp:64-70: 1e-6, 2 or 1/(1-h2) */
    int L;
    const double* probs;          /* unscaled, as on the command line; L == 0 -> default 23-component prior */
    const double* vars;
    const char* out_prefix;       /* out_dir + out_name (vamp.cpp:435); NULL or "" = no per-iteration files */
    int verbose;
    int diagnostics;              /* 1: also run the 3 print-only Ax of vamp.cpp:646-681 */
    double alpha_scale;
    int use_XXT_denoiser;         /* --use-XXT-denoiser (options.cpp:208-216) */
    int bin_class;                /* 1: --model bin_class (vamp_probit.cpp), y in {0,1}; x_est is then the UNSCALED x1_hat */
    double probit_var;            /* --probit-var (options.hpp:124) */
    int fuse_solves;              /* 0 ... 4: see --fuse-solves (docs/history/rounds1-3.md section 5) */
    int C;                        /* probit covariates (--C); 0 = none */
    const double* covs;           /* N x C row-major (the rows of --cov-file), or NULL */
    double* cov_eff_out;          /* C fitted covariate effects (vamp::get_cov_eff), or NULL */
    const char* freeze_index_file; /* --use-freeze 1 --freeze-index-file <text file, one value per global marker>, or NULL */
    int reanchor_every;           /* --reanchor-every K (levels 3 / 4); < 0 = the drivers' default (10), 0 = never */
} gvh_opts;

typedef struct {
    double gam1_denoise, alpha1, eta1, gam2, alpha2, eta2, gam2_reest, gam1_next, gamw, rho, R2_denoise, R2_lmmse;
    int cg_iters, onsager_iters, revar_rounds, L_after;
    long n_ax, n_atx;              /* vector products */
    long n_ax_pass, n_atx_pass;    /* passes over the genotype shard */
    double beta1, tau2, tau1_next; /* bin_class only */
    double seconds, seconds_io;   /* compute wall time of the iteration, and time spent writing / copying iterates */
    int probe_product;            /* --fuse-solves 4: A^T A u of the Onsager probe this iteration: 0 not in play, 1 captured and kept,
                                   * 2 captured but dropped by the cancellation rule, 3 the kept product used */
} gvh_iter;

/* sim.cpp data recipe on a resident shard: beta_out[M] (this rank's slice), y_out[N] (identical on every rank) */
/* message of the last failed gvh_* call of this thread (a C++ exception of the host classes, caught at the boundary) */
const char* gvh_last_error(void);

int gvh_sim_phen(gv_ctx* ctx, int N, int M, int Mt, int S, int rank, double h2, int CV, unsigned long seed,
                 double* beta_out, double* y_out);

/* vamp::infere (vamp.cpp:149) for --model linear on the shard resident in ctx.  y[N]; mask4 NULL = all present.
 * x_est[M] = x1_hat / sqrt(N) of the last iteration.  x1_hist / x2_hist / r1_hist (optional): iters_cap * M doubles. */
int gvh_infere_linear(gv_ctx* ctx, const gvh_opts* o, int N, int M, int Mt, int S, int rank, const double* y,
                      const unsigned char* mask4, int nonas, const double* true_signal, double* x_est,
                      gvh_iter* iters, int iters_cap, int* n_iters, double* x1_hist, double* x2_hist, double* r1_hist,
                      double* probs_out, double* vars_out, int* L_out);

/* ---- host transport: the ranks of one node as processes meeting in POSIX shared memory (gvamp_amd/csrc/host/shm_comm.hpp) -----
 * What GVAMP_COMM=host selects in the drivers; exported so that tests and a reference-side binding can plug the same sums
 * into gv_comm_init_callback (gvh_shm_allreduce has the signature of gv_allreduce_fn, user = the communicator).
 * name: "/something" shared by the ranks of one job (NULL: derived from $GVAMP_RENDEZVOUS or MASTER_PORT + the launcher's pid). */
typedef struct gvh_shm_comm gvh_shm_comm;
int gvh_shm_open(const char* name, int nranks, int rank, size_t cap_doubles, gvh_shm_comm** out);
int gvh_shm_allreduce(void* comm, double* buf, size_t n);
void gvh_shm_close(gvh_shm_comm* comm);
/* The rendezvous of the RCCL drivers (host/data.cpp: gv_host_world), exposed so that it can be exercised without RCCL: the key the
 * ranks of one job derive from their environment ($GVAMP_RENDEZVOUS, else the launcher's job id -- TORCHELASTIC_RUN_ID,
 * SLURM_JOB_ID.SLURM_STEP_ID, PMIX_NAMESPACE, PMI_JOBID, OpenMPI's jobid -- with MASTER_ADDR:MASTER_PORT, else the parent's pid), and
 * the exchange of the 128-byte communicator id through a file (path NULL: the job's default): rank 0 writes id128, the others
 * receive it; 0 = ok, else gvh_last_error names the path and the key. */
const char* gvh_job_key_c(void);
int gvh_exchange_id(const char* path_or_null, int rank, unsigned char* id128, double timeout_s);

#ifdef __cplusplus
}
#endif
#endif
