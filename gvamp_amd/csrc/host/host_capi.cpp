// host_capi.cpp -- flat entry points over the host-side `data` / `vamp` classes, so that bench.py and tests/ can run
// vamp::infere() on a shard that is already resident in a gv_ctx (synthetic shards at UKB scale have no .bed file).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <stdexcept>
#include <string>

#include "data.hpp"
#include "gvamp_host.h"
#include "utilities.hpp"
#include "shm_comm.hpp"
#include "vamp.hpp"

// Exceptions of the host classes (the reference lets them terminate the program, e.g. initialize_prior's
// "No probabilities or variances were specified and Mt < 50,000", utilities.cpp:91-140) stop at the C boundary: the entry
// points return 1 and gvh_last_error() holds the message.
static thread_local std::string g_host_err;
#define GVH_TRY try {
#define GVH_CATCH } catch (const std::exception& e) { g_host_err = e.what(); return 1; } \
                    catch (...) { g_host_err = "unknown C++ exception"; return 1; }

extern "C" {

const char* gvh_last_error(void) { return g_host_err.c_str(); }
int gvh_abi_version(void) { return GVH_ABI_VERSION; }

// sim.cpp:78-79,153,183-218 on an already-resident shard: beta (local M), y (N).  Every rank draws the same beta /
// noise streams; data::Ax carries the cross-rank all-reduce when a communicator is attached.
int gvh_sim_phen(gv_ctx* ctx, int N, int M, int Mt, int S, int rank, double h2, int CV, unsigned long seed,
                 double* beta_out, double* y_out) {
    GVH_TRY
    gv_host_set_quiet(true);   // library use: nothing on stdout (bench.py prints exactly one JSON line)
    std::vector<double> vars_true{0, h2 / CV};
    std::vector<double> probs_true{1 - (double)CV / Mt, (double)CV / Mt};
    const double gamw = 1 / (1 - h2);
    std::vector<double> beta_all = simulate(Mt, vars_true, probs_true, seed);
    std::mt19937 generator{seed};
    std::normal_distribution<double> gauss(0, 1 / sqrt(gamw));
    std::vector<double> noise(N);
    for (int i = 0; i < N; i++) noise[i] = gauss(generator);
    data ds(ctx, std::vector<double>(N, 0.0), N, M, Mt, S, rank);
    std::vector<double> scaled(beta_all.begin() + S, beta_all.begin() + S + M);
    for (double& b : scaled) b *= sqrt(N);
    std::vector<double> y = ds.Ax(scaled.data());
    for (int i = 0; i < N; i++) y[i] += noise[i];
    memcpy(beta_out, beta_all.data() + S, sizeof(double) * M);
    memcpy(y_out, y.data(), sizeof(double) * N);
    return 0;
    GVH_CATCH
}

int gvh_infere_linear(gv_ctx* ctx, const gvh_opts* o, int N, int M, int Mt, int S, int rank, const double* y,
                      const unsigned char* mask4, int nonas, const double* true_signal, double* x_est,
                      gvh_iter* iters, int iters_cap, int* n_iters, double* x1_hist, double* x2_hist, double* r1_hist,
                      double* probs_out, double* vars_out, int* L_out) {
    GVH_TRY
    std::vector<double> yv(y, y + N);
    std::vector<unsigned char> m4;
    if (mask4) m4.assign(mask4, mask4 + (N + 3) / 4);
    data ds(ctx, yv, N, M, Mt, S, rank, mask4 ? &m4 : nullptr, nonas, o->alpha_scale);
    std::vector<double> ts(M > 0 ? M : 0, 0.0);
    if (true_signal) ts.assign(true_signal, true_signal + M);
    std::vector<double> probs, vars;
    if (o->L > 0) {
        probs.assign(o->probs, o->probs + o->L);
        vars.assign(o->vars, o->vars + o->L);
    }
    Options opt;   // carries the knobs the reference reads from the command line
    opt.set_solver(o->EM_max_iter, o->CG_max_iter, o->EM_err_thr, o->stop_criteria_thr, o->learn_vars,
                   (unsigned int)o->seed, o->use_lmmse_damp, o->diagnostics, (o->out_prefix && o->out_prefix[0]) ? 1 : 0);
    opt.set_use_XXT_denoiser(o->use_XXT_denoiser);
    opt.set_probit_var(o->probit_var);
    opt.set_fuse_solves(o->fuse_solves);
    if (o->reanchor_every >= 0) opt.set_reanchor_every(o->reanchor_every);
    if (o->freeze_index_file && o->freeze_index_file[0]) opt.set_freeze(o->freeze_index_file);
    if (o->C > 0 && o->covs) {
        opt.set_C((unsigned int)o->C);
        std::vector<std::vector<double>> z(N, std::vector<double>(o->C));
        for (int i = 0; i < N; i++)
            for (int j = 0; j < o->C; j++) z[i][j] = o->covs[(size_t)i * o->C + j];
        ds.set_covs(std::move(z));
    }
    gv_host_set_quiet(!o->verbose);
    vamp vm(N, M, Mt, o->gam1, o->gamw, o->iterations, o->rho, vars, probs, ts, rank, "",
            o->out_prefix ? o->out_prefix : "", o->bin_class ? "bin_class" : "linear", opt);
    vm.set_verbose(o->verbose);
    vm.set_keep_history((x1_hist || x2_hist || r1_hist) ? 1 : 0);
    std::vector<double> x = vm.infere(&ds);
    if (x_est && M > 0) memcpy(x_est, x.data(), sizeof(double) * M);
    if (o->cov_eff_out) {
        const std::vector<double> ce = vm.get_cov_eff();
        for (size_t j = 0; j < ce.size() && (int)j < o->C; j++) o->cov_eff_out[j] = ce[j];
    }
    const std::vector<vamp_iter_stats>& st = vm.get_stats();
    int n = (int)st.size();
    if (n_iters) *n_iters = n;
    for (int i = 0; i < n && i < iters_cap; i++) {
        gvh_iter& t = iters[i];
        const vamp_iter_stats& s = st[i];
        t.gam1_denoise = s.gam1_denoise; t.alpha1 = s.alpha1; t.eta1 = s.eta1; t.gam2 = s.gam2; t.alpha2 = s.alpha2;
        t.eta2 = s.eta2; t.gam2_reest = s.gam2_reest; t.gam1_next = s.gam1_next; t.gamw = s.gamw; t.rho = s.rho;
        t.R2_denoise = s.R2_denoise; t.R2_lmmse = s.R2_lmmse;
        t.cg_iters = s.cg_iters; t.onsager_iters = s.onsager_iters; t.revar_rounds = s.revar_rounds; t.L_after = s.L_after;
        t.beta1 = s.beta1; t.tau2 = s.tau2; t.tau1_next = s.tau1_next;
        t.n_ax = s.n_ax; t.n_atx = s.n_atx; t.n_ax_pass = s.n_ax_pass; t.n_atx_pass = s.n_atx_pass; t.seconds = s.seconds; t.seconds_io = s.seconds_io;
        t.probe_product = s.probe_product;
    }
    auto dump = [&](const std::vector<std::vector<double>>& h, double* dst) {
        if (!dst) return;
        for (size_t i = 0; i < h.size() && (int)i < iters_cap; i++) memcpy(dst + i * (size_t)M, h[i].data(), sizeof(double) * M);
    };
    dump(vm.get_x1_hist(), x1_hist);
    dump(vm.get_x2_hist(), x2_hist);
    dump(vm.get_r1_hist(), r1_hist);
    std::vector<double> p = vm.get_probs(), v = vm.get_vars();
    if (L_out) *L_out = (int)p.size();
    for (size_t i = 0; i < p.size(); i++) {
        if (probs_out) probs_out[i] = p[i];
        if (vars_out) vars_out[i] = v[i];
    }
    return 0;
    GVH_CATCH
}

int gvh_shm_open(const char* name, int nranks, int rank, size_t cap_doubles, gvh_shm_comm** out) {
    GVH_TRY
    std::string err;
    *out = gvh_shm_open_impl(name ? std::string(name) : gvh_shm_default_name(), nranks, rank, cap_doubles, err);
    if (!*out) throw std::runtime_error(err);
    return 0;
    GVH_CATCH
}
void gvh_shm_close(gvh_shm_comm* comm) { gvh_shm_close_impl(comm); }

// the rendezvous of the RCCL drivers without RCCL (tests): the job key the ranks derive, and the id file exchange
const char* gvh_job_key_c(void) {
    static thread_local std::string k;
    k = gvh_job_key();
    return k.c_str();
}
int gvh_exchange_id(const char* path_or_null, int rank, unsigned char* id128, double timeout_s) {
    std::string err;
    const int rc = gvh_exchange_id_impl(path_or_null ? std::string(path_or_null) : gvh_id_file_default(), rank, id128, timeout_s, err);
    if (rc) g_host_err = err;
    return rc;
}

}  // extern "C"
