// gv_oracle_capi.cpp -- flat C entry points over gv_oracle.{hpp,cpp} for ctypes (tests, smoke(),
// bench.py cpu_baseline).  TEST INFRASTRUCTURE ONLY -- see gv_oracle.hpp.
#include <condition_variable>
#include <cmath>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>

#include "gv_oracle.hpp"

using namespace gvo;

namespace {

// In-process stand-in for MPI ranks: one thread per shard, SUM in rank order (deterministic).
struct ThreadGroup {
    int n;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    long gen = 0;
    std::vector<const double*> slots;
    explicit ThreadGroup(int n_) : n(n_), slots(n_, nullptr) {}
    void barrier() {
        std::unique_lock<std::mutex> lk(mu);
        long g = gen;
        if (++arrived == n) {
            arrived = 0;
            gen++;
            cv.notify_all();
        } else
            cv.wait(lk, [&] { return gen != g; });
    }
};
struct ThreadComm : Comm {
    ThreadGroup* g;
    ThreadComm(ThreadGroup* g_, int r) : g(g_) { rank = r; nranks = g_->n; }
    void allreduce_sum(double* buf, size_t n) override {
        std::vector<double> mine(buf, buf + n);
        g->slots[rank] = mine.data();
        g->barrier();
        for (size_t i = 0; i < n; i++) {
            double s = 0;
            for (int r = 0; r < nranks; r++) s += g->slots[r][i];
            buf[i] = s;
        }
        g->barrier();
    }
};
typedef void (*allreduce_cb_t)(double* buf, long n, void* user);
struct CallbackComm : Comm {
    allreduce_cb_t cb;
    void* user;
    void allreduce_sum(double* buf, size_t n) override { cb(buf, (long)n, user); }
};

}  // namespace

extern "C" {

struct gvo_params {
    int N, Mt;
    int nshards;       // >= 1
    int shard_rank;    // -1: run all shards in threads; else run only this shard with the callback comm
    int iterations, CG_max_iter, EM_max_iter;
    double EM_err_thr, stop_criteria_thr, rho;
    int learn_vars;
    unsigned long seed;
    int use_lmmse_damp;
    double gam1, gamw;
    int L;
    const double* probs;
    const double* vars;
    const double* true_signal;  // Mt or NULL
    const char* out_prefix;     // NULL or "" = no files
    int verbose, nthreads;
    double alpha_scale;
    int phen_mode;              // 0: vector ctor (data.cpp:69-113, full mask, y as given); 1: file semantics
    const unsigned char* is_na; // N flags, phen_mode 1 only
    allreduce_cb_t cb;          // shard_rank >= 0 && nshards > 1
    void* cb_user;
    int use_XXT_denoiser;       // vamp.cpp:169-170, :599-606
    const double* r1_init;      // Mt or NULL: restart (vamp.cpp:226-233)
    const double* x_init;       // Mt or NULL: --init-est 1 (vamp.cpp:244-258)
    int bin_class;              // --model bin_class (vamp_probit.cpp)
    double probit_var;
    int C;                      // covariates of the probit model (--C)
    const double* covs;         // N x C row-major, or NULL
    const double* freeze_ind;   // Mt or NULL: --use-freeze 1 (vamp.cpp:205-209)
};

struct gvo_run {
    int N, Mt, nshards;
    std::vector<double> x_est;                 // Mt
    std::vector<IterTrace> trace;              // of shard 0 (scalars agree across shards)
    std::vector<std::vector<double>> x1_hist, x2_hist, r1_hist;  // [it][Mt]
    std::vector<double> gam1s, gam2s, R2trains, probs, vars;
    std::vector<double> cov_eff;               // probit covariate effects (vamp.hpp:24)
    std::vector<std::vector<double>> cg_relres;
    std::vector<double> mave, msig;            // Mt
};

int gvo_mbytes(int N) { return (N % 4) ? N / 4 + 1 : N / 4; }

void gvo_divide_work(int Mt, int nranks, int rank, int* M, int* S) { divide_work(Mt, nranks, rank, M, S, nullptr); }

static void fill_data(Data& d, const unsigned char* bed, int N, int M, const unsigned char* mask4, int nonas,
                      double alpha_scale, int nthreads) {
    d.N = N; d.M = M; d.Mt = M; d.S = 0; d.bed = bed; d.alpha_scale = alpha_scale; d.nthreads = nthreads;
    d.init_full_mask();
    if (mask4) {
        d.mask4.assign(mask4, mask4 + d.mbytes);
        d.nonas = nonas;
    }
}

int gvo_marker_stats(const unsigned char* bed, int N, int M, const unsigned char* mask4, int nonas,
                     double alpha_scale, int nthreads, double* mave, double* msig) {
    Data d;
    fill_data(d, bed, N, M, mask4, nonas, alpha_scale, nthreads);
    d.compute_markers_statistics();
    memcpy(mave, d.mave.data(), sizeof(double) * M);
    memcpy(msig, d.msig.data(), sizeof(double) * M);
    return 0;
}

int gvo_ax(const unsigned char* bed, int N, int M, const unsigned char* mask4, const double* mave,
           const double* msig, const double* x, int nthreads, double* out) {
    Data d;
    fill_data(d, bed, N, M, mask4, N, 1.0, nthreads);
    d.mave.assign(mave, mave + M);
    d.msig.assign(msig, msig + M);
    std::vector<double> r = d.Ax(x);
    memcpy(out, r.data(), sizeof(double) * r.size());
    return 0;
}

int gvo_atx(const unsigned char* bed, int N, int M, const double* mave, const double* msig, const double* p,
            int nthreads, double* out) {
    Data d;
    fill_data(d, bed, N, M, nullptr, N, 1.0, nthreads);
    d.mave.assign(mave, mave + M);
    d.msig.assign(msig, msig + M);
    std::vector<double> r = d.ATx(p);
    memcpy(out, r.data(), sizeof(double) * r.size());
    return 0;
}

// data::pvals_calc / pvals_calc_LOCO (data.cpp:1108-1353), single full-mask or masked shard, one estimator.
// chrom == NULL: leave-one-out; else leave-one-chromosome-out.
int gvo_pvals(const unsigned char* bed, int N, int M, const unsigned char* mask4, int nonas, const double* z1,
              const double* y, const double* x1_hat, const int* chrom, int nthreads, double* pvals) {
    Data d;
    fill_data(d, bed, N, M, mask4, nonas, 1.0, nthreads);
    d.compute_markers_statistics();
    std::vector<double> zz(z1, z1 + N), yy(y, y + N), xx(x1_hat, x1_hat + M);
    std::vector<double> r;
    if (chrom) r = d.pvals_calc_LOCO(zz, yy, xx, std::vector<int>(chrom, chrom + M));
    else r = d.pvals_calc(zz, yy, xx);
    memcpy(pvals, r.data(), sizeof(double) * M);
    return 0;
}
double gvo_student_t_two_sided(double t, double nu) { return student_t_two_sided(t, nu); }

// data::compute_people_statistics (data.cpp:558-716) on a single shard: three vectors of 4*mbytes doubles
int gvo_people_stats(const unsigned char* bed, int N, int M, const unsigned char* mask4, int nonas, double* mave_p,
                     double* msig_p, double* numb_p) {
    Data d;
    fill_data(d, bed, N, M, mask4, nonas, 1.0, 1);
    d.compute_markers_statistics();
    d.compute_people_statistics();
    memcpy(mave_p, d.mave_people.data(), sizeof(double) * 4 * d.mbytes);
    memcpy(msig_p, d.msig_people.data(), sizeof(double) * 4 * d.mbytes);
    memcpy(numb_p, d.numb_people.data(), sizeof(double) * 4 * d.mbytes);
    return 0;
}

// g1 / g1d on a grid; vars already multiplied by N (as inside infere).
void gvo_g1_g1d(const double* r, long n, double gam1, const double* probs, const double* vars, int L, double* g1,
                double* g1d) {
    Data d;
    d.N = 4; d.M = 1; d.Mt = 1;
    VampOpts o;
    o.probs.assign(probs, probs + L);
    o.vars.assign(vars, vars + L);
    Vamp v(&d, o);
    for (long i = 0; i < n; i++) {
        g1[i] = v.g1(r[i], gam1);
        g1d[i] = v.g1d(r[i], gam1);
    }
}

// one updatePrior call on a single shard (vars already scaled by N).  Returns the new L.
int gvo_update_prior(const double* r1, int M, int Mt, double gam1, double* probs, double* vars, int L,
                     int EM_max_iter, double EM_err_thr, int learn_vars) {
    Data d;
    d.N = 4; d.M = M; d.Mt = Mt;
    VampOpts o;
    o.probs.assign(probs, probs + L);
    o.vars.assign(vars, vars + L);
    o.EM_max_iter = EM_max_iter;
    o.EM_err_thr = EM_err_thr;
    o.learn_vars = learn_vars;
    Vamp v(&d, o);
    v.gam1 = gam1;
    v.r1.assign(r1, r1 + M);
    v.updatePrior(0);
    for (size_t i = 0; i < v.probs.size(); i++) { probs[i] = v.probs[i]; vars[i] = v.vars[i]; }
    return (int)v.probs.size();
}

// one precondCG_solver call on a single full-mask shard.  relres must hold CG_max_iter doubles.
int gvo_cg_solve(const unsigned char* bed, int N, int M, const double* v, const double* mu_start, double tau,
                 double gam2, int denoiser, int CG_max_iter, int nthreads, double* mu_out, double* relres) {
    Data d;
    fill_data(d, bed, N, M, nullptr, N, 1.0, nthreads);
    d.compute_markers_statistics();
    VampOpts o;
    o.CG_max_iter = CG_max_iter;
    o.probs = {1.0};
    o.vars = {0.0};
    Vamp vm(&d, o);
    vm.gam2 = gam2;
    std::vector<double> vv(v, v + M), ms(M, 0.0);
    if (mu_start) ms.assign(mu_start, mu_start + M);
    std::vector<double> mu = vm.precondCG_solver(vv, ms, tau, denoiser);
    memcpy(mu_out, mu.data(), sizeof(double) * M);
    const std::vector<double>& rr = vm.cg_relres.back();
    for (size_t i = 0; i < rr.size(); i++) relres[i] = rr[i];
    return (int)rr.size();
}

// sim.cpp:78-79,153,183-218 -- beta (Mt), y (N) from a full-mask dataset, single rank.
int gvo_sim_phen(const unsigned char* bed, int N, int Mt, double h2, int CV, unsigned long seed, int nthreads,
                 double* beta_out, double* y_out) {
    Data d;
    fill_data(d, bed, N, Mt, nullptr, N, 1.0, nthreads);
    d.compute_markers_statistics();
    std::vector<double> vars_true{0, h2 / CV};
    std::vector<double> probs_true{1 - (double)CV / Mt, (double)CV / Mt};
    double gamw = 1 / (1 - h2);
    std::vector<double> beta_true = simulate(Mt, vars_true, probs_true, seed);
    std::mt19937 generator{seed};
    std::normal_distribution<double> gauss_beta_gen(0, 1 / sqrt(gamw));
    std::vector<double> noise(N, 0.0);
    for (int i = 0; i < N; i++) noise[i] = gauss_beta_gen(generator);
    std::vector<double> beta_true_scaled = beta_true;
    for (int i0 = 0; i0 < Mt; i0++) beta_true_scaled[i0] *= sqrt(N);
    std::vector<double> y = d.Ax(beta_true_scaled.data());
    for (int i = 0; i < N; i++) y[i] += noise[i];
    memcpy(beta_out, beta_true.data(), sizeof(double) * Mt);
    memcpy(y_out, y.data(), sizeof(double) * N);
    return 0;
}

// vamp.cpp:875-882 -- the Hutchinson probe of a shard (for host-side parity of the product's RNG).
void gvo_bern_vec(unsigned long seed, int S, int M, int Mt, double* out) {
    std::mt19937 rd{seed + (unsigned long)S};
    std::bernoulli_distribution bern(0.5);
    for (int i = 0; i < M; i++) out[i] = (2 * bern(rd) - 1) / sqrt(Mt);
}

static void run_shard(const gvo_params* p, const unsigned char* bed_full, const double* y, int rank, Comm* comm,
                      gvo_run* out, std::mutex* mu) {
    int M, S;
    divide_work(p->Mt, p->nshards, rank, &M, &S, nullptr);
    Data d;
    d.N = p->N; d.M = M; d.Mt = p->Mt; d.S = S;
    d.alpha_scale = p->alpha_scale;
    d.nthreads = p->nthreads;
    d.comm = comm;
    d.init_full_mask();
    d.bed = bed_full + size_t(S) * d.mbytes;
    if (p->phen_mode == 1) {
        std::vector<double> raw(y, y + p->N);
        std::vector<uint8_t> na(p->is_na, p->is_na + p->N);
        d.set_phen_file_semantics(raw, na);
    } else
        d.phen.assign(y, y + p->N);
    d.compute_markers_statistics();
    VampOpts o;
    o.iterations = p->iterations; o.CG_max_iter = p->CG_max_iter; o.EM_max_iter = p->EM_max_iter;
    o.EM_err_thr = p->EM_err_thr; o.stop_criteria_thr = p->stop_criteria_thr; o.rho = p->rho;
    o.learn_vars = p->learn_vars; o.seed = p->seed; o.use_lmmse_damp = p->use_lmmse_damp;
    o.gam1 = p->gam1; o.gamw = p->gamw;
    if (p->L > 0) { o.probs.assign(p->probs, p->probs + p->L); o.vars.assign(p->vars, p->vars + p->L); }
    if (p->true_signal) o.true_signal.assign(p->true_signal + S, p->true_signal + S + M);
    if (p->out_prefix) o.out_prefix = p->out_prefix;
    o.verbose = p->verbose;
    o.use_XXT_denoiser = p->use_XXT_denoiser;
    if (p->r1_init) o.r1_init.assign(p->r1_init + S, p->r1_init + S + M);
    if (p->x_init) o.x_init.assign(p->x_init + S, p->x_init + S + M);
    o.bin_class = p->bin_class;
    o.probit_var = p->probit_var;
    if (p->freeze_ind) o.freeze_ind.assign(p->freeze_ind + S, p->freeze_ind + S + M);
    if (p->C > 0 && p->covs) {
        o.C = p->C;
        o.covs.assign(p->covs, p->covs + (size_t)p->N * p->C);
    }
    Vamp v(&d, o);
    std::vector<double> x = v.infere();
    std::lock_guard<std::mutex> lk(*mu);
    if (rank == 0 || p->shard_rank >= 0) out->cov_eff = v.cov_eff;
    size_t nit = v.x1_hist.size();
    if (out->x1_hist.size() < nit) {
        out->x1_hist.resize(nit, std::vector<double>(p->Mt, 0.0));
        out->x2_hist.resize(nit, std::vector<double>(p->Mt, 0.0));
        out->r1_hist.resize(nit, std::vector<double>(p->Mt, 0.0));
    }
    for (size_t k = 0; k < nit; k++) {
        std::copy(v.x1_hist[k].begin(), v.x1_hist[k].end(), out->x1_hist[k].begin() + S);
        std::copy(v.x2_hist[k].begin(), v.x2_hist[k].end(), out->x2_hist[k].begin() + S);
        std::copy(v.r1_hist[k].begin(), v.r1_hist[k].end(), out->r1_hist[k].begin() + S);
    }
    std::copy(x.begin(), x.end(), out->x_est.begin() + S);
    std::copy(d.mave.begin(), d.mave.end(), out->mave.begin() + S);
    std::copy(d.msig.begin(), d.msig.end(), out->msig.begin() + S);
    if (rank == 0 || p->shard_rank >= 0) {
        out->trace = v.trace; out->gam1s = v.gam1s; out->gam2s = v.gam2s; out->R2trains = v.R2trains;
        out->probs = v.probs; out->vars = v.vars; out->cg_relres = v.cg_relres;
    }
}

// vamp::infere on nshards marker shards.  bed_full: Mt*mbytes marker-major bytes (no magic), y: N.
gvo_run* gvo_infere(const gvo_params* p, const unsigned char* bed_full, const double* y) {
    gvo_run* out = new gvo_run();
    out->N = p->N; out->Mt = p->Mt; out->nshards = p->nshards;
    out->x_est.assign(p->Mt, 0.0);
    out->mave.assign(p->Mt, 0.0);
    out->msig.assign(p->Mt, 0.0);
    std::mutex mu;
    if (p->shard_rank >= 0) {
        CallbackComm cc;
        cc.rank = p->shard_rank; cc.nranks = p->nshards; cc.cb = p->cb; cc.user = p->cb_user;
        run_shard(p, bed_full, y, p->shard_rank, p->nshards > 1 ? &cc : nullptr, out, &mu);
    } else if (p->nshards == 1) {
        run_shard(p, bed_full, y, 0, nullptr, out, &mu);
    } else {
        ThreadGroup g(p->nshards);
        std::vector<std::thread> th;
        std::vector<ThreadComm*> comms;
        for (int r = 0; r < p->nshards; r++) comms.push_back(new ThreadComm(&g, r));
        for (int r = 0; r < p->nshards; r++) th.emplace_back(run_shard, p, bed_full, y, r, comms[r], out, &mu);
        for (auto& t : th) t.join();
        for (auto c : comms) delete c;
    }
    return out;
}

void gvo_run_free(gvo_run* r) { delete r; }
int gvo_run_niter(const gvo_run* r) { return (int)r->trace.size(); }
const double* gvo_run_x_est(const gvo_run* r) { return r->x_est.data(); }
const double* gvo_run_mave(const gvo_run* r) { return r->mave.data(); }
const double* gvo_run_msig(const gvo_run* r) { return r->msig.data(); }
const double* gvo_run_x1(const gvo_run* r, int it) { return r->x1_hist[it].data(); }
const double* gvo_run_x2(const gvo_run* r, int it) { return r->x2_hist[it].data(); }
const double* gvo_run_r1(const gvo_run* r, int it) { return r->r1_hist[it].data(); }
int gvo_run_L(const gvo_run* r) { return (int)r->probs.size(); }
const double* gvo_run_probs(const gvo_run* r) { return r->probs.data(); }
const double* gvo_run_vars(const gvo_run* r) { return r->vars.data(); }
int gvo_run_nsolves(const gvo_run* r) { return (int)r->cg_relres.size(); }
int gvo_run_relres(const gvo_run* r, int solve, double* out, int cap) {
    const std::vector<double>& v = r->cg_relres[solve];
    for (int i = 0; i < (int)v.size() && i < cap; i++) out[i] = v[i];
    return (int)v.size();
}
// 16 doubles per iteration: gam1_denoise alpha1 eta1 gam2 alpha2 eta2 gam2_reest gam1_next gamw rho
//                           cg_iters onsager_iters revar_rounds L_after n_ax n_atx   (+ seconds via _seconds)
void gvo_run_trace(const gvo_run* r, int it, double* out16) {
    const IterTrace& t = r->trace[it];
    double v[16] = {t.gam1_denoise, t.alpha1, t.eta1, t.gam2, t.alpha2, t.eta2, t.gam2_reest, t.gam1_next,
                    t.gamw, t.rho, (double)t.cg_iters, (double)t.onsager_iters, (double)t.revar_rounds,
                    (double)t.L_after, (double)t.n_ax, (double)t.n_atx};
    memcpy(out16, v, sizeof(v));
}
double gvo_run_seconds(const gvo_run* r, int it) { return r->trace[it].seconds; }
// bin_class extras: beta1, tau2 (after re-estimation), tau1 entering the next iteration
void gvo_run_trace_probit(const gvo_run* r, int it, double* out3) {
    out3[0] = r->trace[it].beta1;
    out3[1] = r->trace[it].tau2;
    out3[2] = r->trace[it].tau1_next;
}
// vamp::g1_bin_class / g1d_bin_class on a grid (y in {0,1}), and erfcx
int gvo_run_cov_eff(const gvo_run* r, double* out, int cap) {
    for (int i = 0; i < (int)r->cov_eff.size() && i < cap; i++) out[i] = r->cov_eff[i];
    return (int)r->cov_eff.size();
}
// vamp::Newton_method_cov (vamp_probit.cpp:936-1062) on its own: y, gg of length N, covs N x C row-major, eta0 / out of
// length C; also the value of mlogL_probit and grad_cov at the returned point (stats[0], grad[C])
void gvo_newton_cov(int N, int C, const double* covs, const double* y, const double* gg, double probit_var,
                    const double* eta0, double* eta_out, double* mlogL_out, double* grad_out) {
    Data d;
    d.N = N; d.M = 1; d.Mt = 1;
    VampOpts o;
    o.probs = {1.0};
    o.vars = {0.0};
    o.probit_var = probit_var;
    o.C = C;
    o.covs.assign(covs, covs + (size_t)N * C);
    Vamp v(&d, o);
    std::vector<double> yy(y, y + N), g(gg, gg + N), e0(eta0, eta0 + C);
    std::vector<double> e = v.Newton_method_cov(yy, g, e0);
    for (int j = 0; j < C; j++) eta_out[j] = e[j];
    if (mlogL_out) *mlogL_out = v.mlogL_probit(yy, g, e);
    if (grad_out) {
        std::vector<double> gr = v.grad_cov(yy, g, e);
        for (int j = 0; j < C; j++) grad_out[j] = gr[j];
    }
}
int gvo_lu_solve(const double* A, double* b, int n) {
    std::vector<double> a(A, A + (size_t)n * n), bb(b, b + n);
    const bool ok = lu_solve(a, bb, n);
    for (int i = 0; i < n; i++) b[i] = bb[i];
    return ok ? 0 : 1;
}
void gvo_probit_g(const double* p, const double* y, long n, double tau1, double probit_var, double* g, double* gd) {
    Data d;
    d.N = 4; d.M = 1; d.Mt = 1;
    VampOpts o;
    o.probs = {1.0};
    o.vars = {0.0};
    o.probit_var = probit_var;
    Vamp v(&d, o);
    for (long i = 0; i < n; i++) {
        g[i] = v.g1_bin_class(p[i], tau1, y[i], 0);
        gd[i] = v.g1d_bin_class(p[i], tau1, y[i], 0);
    }
}
void gvo_probit_g_cov(const double* p, const double* y, const double* m_cov, long n, double tau1, double probit_var,
                      double* g, double* gd) {
    Data d;
    d.N = 4; d.M = 1; d.Mt = 1;
    VampOpts o;
    o.probs = {1.0};
    o.vars = {0.0};
    o.probit_var = probit_var;
    Vamp v(&d, o);
    for (long i = 0; i < n; i++) {
        g[i] = v.g1_bin_class(p[i], tau1, y[i], m_cov[i]);
        gd[i] = v.g1d_bin_class(p[i], tau1, y[i], m_cov[i]);
    }
}
double gvo_erfcx(double x) { return erfcx(x); }
int gvo_run_R2trains(const gvo_run* r, double* out, int cap) {
    for (int i = 0; i < (int)r->R2trains.size() && i < cap; i++) out[i] = r->R2trains[i];
    return (int)r->R2trains.size();
}

}  // extern "C"
