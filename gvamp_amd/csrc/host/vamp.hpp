// vamp.hpp -- host side of the reference's `class vamp` (vamp.hpp:7-151) for `--model linear`.
// Same constructors and entry point (infere(data*)); every M- or N-length vector of vamp.cpp lives in HBM as a
// gv_vec and every loop over markers / individuals is a HIP kernel behind include/gvamp.h.  The host keeps only the
// scalar recursions (gam1, gam2, eta, alpha, rho, gamw), the prior (probs / vars) and the Hutchinson RNG.
#pragma once
#include <string>
#include <vector>

#include "data.hpp"
#include "options.hpp"

struct vamp_iter_stats {     // one row per VAMP iteration (what the reference prints on rank 0)
    double gam1_denoise, alpha1, eta1, gam2, alpha2, eta2, gam2_reest, gam1_next, gamw, rho, R2_denoise, R2_lmmse;
    int cg_iters, onsager_iters, revar_rounds, L_after;
    long n_ax, n_atx;                 // vector products
    long n_ax_pass, n_atx_pass;       // passes over the genotype shard (two-vector kernels share a pass)
    double beta1, tau2, tau1_next;   // bin_class only
    double seconds, seconds_io;
    // --fuse-solves 4: what happened to A^T A u of the Onsager probe in this iteration: 0 not in play, 1 captured from the solve's
    // first application and kept, 2 captured but not kept (cancellation, vamp::probe_product_is_usable), 3 the kept product used
    int probe_product;
};

class vamp {
private:
    int N, M, Mt, S = 0, C = 0, max_iter, rank, nranks;
    double gam1, gam2 = 0, gam_before = 0, eta1 = 0, eta2 = 0;
    std::vector<double> gam1s, gam2s, R2trains;
    double alpha1 = 0, alpha2 = 0, rho, gamw;
    std::vector<double> true_signal;
    std::vector<double> probs, probs_before, vars, vars_before;
    double gamma_min = 1e-11, gamma_max = 1e11;          // vamp.hpp:31-32
    int EM_max_iter, CG_max_iter, auto_var_max_iter = 5; // vamp.hpp:34-37
    double EM_err_thr, stop_criteria_thr;
    int learn_vars, init_est = 0;
    long unsigned int seed;
    std::string model, out_dir, out_name;
    int store_pvals = 0, use_lmmse_damp = 0, reverse = 0;
    double gam1_init = -1, gamw_init = 0;
    std::string r1_init_file, estimate_file;
    int diagnostics = 0, store_iterates = 1, verbose = 1;
    int fuse_solves = 4;        // = Options' default (options.hpp); every constructor overwrites it from the Options it is given
    int reanchor_every = 10;    // levels 3 / 4: iterations between explicit re-anchors of the chained products (0 = never)
    int cur_it = 0;             // iteration infere_linear is in (read by fused_solves)
    int probe_product_state = 0;   // vamp_iter_stats::probe_product of the solve just made
    double linearity_max = 100.0; // XXT level 4: A r2 = c1 z1 - c2 A r1 only while both combinations amplify rounding by <= this (development override: GV_LINEARITY_MAX)
    bool reanchor_now() const { return reanchor_every > 0 && cur_it > 1 && cur_it % reanchor_every == 0; }
    double probit_var = 1;   // options.hpp:124
    // covariates of the probit model (--C > 0, --cov-file): effects fitted once in iteration 1 (vamp_probit.cpp:110-126)
    std::vector<double> cov_eff;                                                          // vamp.hpp:24
    std::vector<double> grad_cov(const std::vector<double>& y, const std::vector<double>& gg, double probit_var,
                                 const std::vector<std::vector<double>>& Z, const std::vector<double>& eta);    // :813-838
    double mlogL_probit(const std::vector<double>& y, const std::vector<double>& gg, double probit_var,
                        const std::vector<std::vector<double>>& Z, const std::vector<double>& eta);             // :840-857
    std::vector<double> Newton_method_cov(const std::vector<double>& y, const std::vector<double>& gg,
                                          const std::vector<std::vector<double>>& Z, std::vector<double> eta);  // :936-1062

    // device state (allocated in infere_linear)
    gv_ctx* ctx = nullptr;
    gv_vec *x1_hat = nullptr, *x1_hat_prev = nullptr, *x2_hat = nullptr, *r1 = nullptr, *r2 = nullptr, *r2_prev = nullptr,
           *z1 = nullptr, *y = nullptr, *mu_CG_last = nullptr, *bern_vec = nullptr, *invQ_bern_vec = nullptr,
           *vM = nullptr, *tM = nullptr, *tN = nullptr, *tN2 = nullptr, *mu_CG_last_N = nullptr;
    bool have_mu_CG_last = false, have_aty = false;
    gv_vec* aty = nullptr;   // A^T y, constant over the iterations (computed once instead of at vamp.cpp:588 every time)
    std::vector<vamp_iter_stats> stats;
    std::vector<std::vector<double>> x1_hist, x2_hist, r1_hist;   // per iteration, already / sqrt(N) (if keep_history)
    int keep_history = 0;

    void ck(int rc, const char* what);
    double dotM(gv_vec* a, gv_vec* b);   // inner_prod(a, b, 1): all-reduced
    double dotN(gv_vec* a, gv_vec* b);   // inner_prod(a, b, 0): N-vectors are replicated
    void store_scaled(const std::string& path, gv_vec* v, std::vector<std::vector<double>>* hist);
    int cg(gv_vec* v, gv_vec* mu_start, double tau, int denoiser, gv_vec* mu_out, int* iters);
    void common_init(const Options& opt);

public:
    std::vector<double> get_cov_eff() const { return cov_eff; }                 // vamp.hpp:149
    vamp(int N, int M, int Mt, double gam1, double gamw, int max_iter, double rho, std::vector<double> vars,
         std::vector<double> probs, std::vector<double> true_signal, int rank, std::string out_dir,
         std::string out_name, std::string model, Options opt = Options());                    // vamp.cpp:32-82
    vamp(int M, double gam1, double gamw, std::vector<double> true_signal, int rank, Options opt);   // vamp.cpp:89-139
    ~vamp();

    std::vector<double> infere(data* dataset);           // vamp.cpp:149-183
    std::vector<double> infere_linear(data* dataset);    // vamp.cpp:190-803
    std::vector<double> infere_bin_class(data* dataset); // vamp_probit.cpp:20-658
    double g2d_onsager(double gam2, double tau, data* dataset, int* iters);   // vamp.cpp:871-889
    void draw_onsager_probe(data* dataset);
    bool probe_product_is_usable(double tau, double gam2);                    // level 4: is the captured A^T A u accurate enough to keep?                                   // vamp.cpp:875-882 (host RNG)
    // LMMSE solve (:593-596) and Onsager probe solve (:884) in lock-step on the shared operator; returns alpha2
    double fused_solves(gv_vec* v, gv_vec* mu_start, double tau, data* dataset, int* cg_iters, int* onsager_iters,
                        gv_vec* ride_x = nullptr, gv_vec* ride_out = nullptr, bool defer_dot = false);
    gv_vec *aat_der = nullptr;        // --use-XXT-denoiser 1: A A^T u from the residual of the N-space solve
    gv_vec *ax2_der = nullptr, *ata_der = nullptr;   // --fuse-solves 2: A x2_hat and A^T A invQ u as by-products of the solves
    bool have_derived = false;
    // --fuse-solves 3: A^T A x2_hat from the final residual of the LMMSE solve; the next iteration's warm start (mu_CG_last =
    // this x2_hat) forms its initial residual from it instead of applying the operator (gvamp.h: gv_cg_warm)
    gv_vec* ata_x2 = nullptr;
    gv_vec* ata_u = nullptr;        // level 3: A^T A u of the Onsager probe (the same u every iteration): its solve starts one step in
    bool have_ata_u = false;
    gv_vec* at_u = nullptr;         // --use-XXT-denoiser 1 at level 3: A^T u, accumulated inside the N-space solve
    // --use-XXT-denoiser 1 at level 4: A r1 and A r2 carried from iteration to iteration -- r2 = (eta1 x1_hat - gam1 r1) / gam2
    // (vamp.cpp:485-486) and r1 = (eta2 x2_hat - gam2 r2) / gam1 (:706-707) are linear in vectors whose products are at hand, so
    // A r2 costs no pass: the first pass of the joint solve carries z1 = A x1_hat instead, and z1 needs no pass of its own
    gv_vec *ar1 = nullptr, *ar2 = nullptr;
    bool have_ar1 = false;
    double r2_c1 = 0, r2_c2 = 0;    // r2 = r2_c1 x1_hat - r2_c2 r1 of the current iteration
    bool have_ata_x2 = false, warm_chain = false, have_aat_prev = false;
    // --use-freeze 1 (vamp.cpp:205-209,:308,:353): markers whose g1d does not enter alpha1 and that are not damped
    int use_freeze = 0;
    std::string freeze_index_file;
    gv_vec *unfrozen = nullptr, *frozen = nullptr, *dvec = nullptr;   // 0/1 masks and the g1d values (M-space)
    bool have_probe = false;          // draw_onsager_probe: bern_vec is resident for (seed + S) == probe_key
    long unsigned int probe_key = 0;
    void updatePrior(int verbose);                       // vamp.cpp:929-1072

    void set_verbose(int v) { verbose = v; }
    void set_keep_history(int k) { keep_history = k; }
    const std::vector<vamp_iter_stats>& get_stats() const { return stats; }
    const std::vector<std::vector<double>>& get_x1_hist() const { return x1_hist; }
    const std::vector<std::vector<double>>& get_x2_hist() const { return x2_hist; }
    const std::vector<std::vector<double>>& get_r1_hist() const { return r1_hist; }
    std::vector<double> get_probs() const { return probs; }
    std::vector<double> get_vars() const { return vars; }
    std::vector<double> get_gam1s() const { return gam1s; }
    std::vector<double> get_gam2s() const { return gam2s; }
    std::vector<double> get_R2trains() const { return R2trains; }
};
