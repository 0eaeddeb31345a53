// vamp.cpp -- see vamp.hpp.  Statement order of infere_linear follows vamp.cpp:261-759 line by line (SURVEY App. A:
// "statement order matters"); deviations are listed in DESIGN.md (all result-neutral).
#include "vamp.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <random>
#include <stdexcept>

#include "utilities.hpp"

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}

void vamp::ck(int rc, const char* what) {
    if (rc) {
        std::cout << "FATAL: " << what << ": " << gv_last_error(ctx) << std::endl;
        exit(EXIT_FAILURE);
    }
}

void vamp::common_init(const Options& opt) {
    C = opt.get_C();
    init_est = opt.get_init_est();
    seed = opt.get_seed();
    estimate_file = opt.get_estimate_file();
    learn_vars = opt.get_learn_vars();
    EM_max_iter = opt.get_EM_max_iter();
    EM_err_thr = opt.get_EM_err_thr();
    CG_max_iter = opt.get_CG_max_iter();
    reverse = opt.get_use_XXT_denoiser();
    use_lmmse_damp = opt.get_use_lmmse_damp();
    stop_criteria_thr = opt.get_stop_criteria_thr();
    gam1_init = opt.get_gam1_init();
    gamw_init = opt.get_gamw_init();
    r1_init_file = opt.get_estimate_file();
    store_pvals = opt.get_store_pvals();   // the reference's ctor 1 leaves 1 here (vamp.hpp:53); --store-pvals is honoured
    diagnostics = opt.get_diagnostics();
    store_iterates = opt.get_store_iterates();
    fuse_solves = opt.get_fuse_solves();
    reanchor_every = opt.get_reanchor_every();
    if (const char* lm = getenv("GV_LINEARITY_MAX")) linearity_max = atof(lm);
    if (opt.get_redglob() != 0) {   // vamp.cpp:57,:594: CG on a sub-range of individuals (cross-validation variants)
        std::cout << "FATAL: --red " << opt.get_redglob() << " (sub-range CG / cross-validation) is not built" << std::endl;
        exit(EXIT_FAILURE);
    }
    use_freeze = (int)opt.get_use_freeze();
    freeze_index_file = opt.get_freeze_index_file();
    probit_var = opt.get_probit_var();
    nranks = gv_env_nranks();
    initialize_prior(this->probs, this->vars, N, Mt, rank);
}

vamp::vamp(int N, int M, int Mt, double gam1, double gamw, int max_iter, double rho, std::vector<double> vars,
           std::vector<double> probs, std::vector<double> true_signal, int rank, std::string out_dir,
           std::string out_name, std::string model, Options opt)
    : N(N), M(M), Mt(Mt), max_iter(max_iter), rank(rank), gam1(gam1), rho(rho), gamw(gamw), true_signal(true_signal),
      probs(probs), vars(vars), model(model), out_dir(out_dir), out_name(out_name) {
    common_init(opt);
}

vamp::vamp(int M, double gam1, double gamw, std::vector<double> true_signal, int rank, Options opt)
    : N(opt.get_N()), M(M), Mt(opt.get_Mt()), max_iter(opt.get_iterations()), rank(rank), gam1(gam1), rho(opt.get_rho()),
      gamw(gamw), true_signal(true_signal), probs(opt.get_probs()), vars(opt.get_vars()), model(opt.get_model()),
      out_dir(opt.get_out_dir()), out_name(opt.get_out_name()) {
    common_init(opt);
}

vamp::~vamp() {
    if (!ctx) return;
    for (gv_vec* v : {x1_hat, x1_hat_prev, x2_hat, r1, r2, r2_prev, z1, y, mu_CG_last, bern_vec, invQ_bern_vec, vM, tM,
                      tN, tN2, mu_CG_last_N, aty, ax2_der, ata_der, aat_der, ata_x2, at_u, ata_u, unfrozen, frozen, dvec, ar1, ar2})
        if (v) gv_vec_free(ctx, v);
}

double vamp::dotM(gv_vec* a, gv_vec* b) {
    double out = 0;
    ck(gv_vec_dot(ctx, a, b, 1, &out), "gv_vec_dot");
    return out;
}
double vamp::dotN(gv_vec* a, gv_vec* b) {
    double out = 0;
    ck(gv_vec_dot(ctx, a, b, 0, &out), "gv_vec_dot");
    return out;
}

// vamp.cpp:447-462 / :538-542 / :608-612: vec / sqrt(N) at byte offset S*8 of a shared .bin
void vamp::store_scaled(const std::string& path, gv_vec* v, std::vector<std::vector<double>>* hist) {
    if (!store_iterates && !(hist && keep_history)) return;
    std::vector<double> h(M > 0 ? M : 0);
    ck(gv_vec_download(ctx, v, h.data()), "gv_vec_download");
    const double scale = sqrt((double)N);
    for (double& x : h) x /= scale;
    if (store_iterates && !path.empty()) mpi_store_vec_to_file(path, h, S, M);
    if (hist && keep_history) hist->push_back(h);
}

// vamp::precondCG_solver (vamp.cpp:1130-1229) = gv_cg_solve; mu_start == nullptr is the zero start of :1120-1128
int vamp::cg(gv_vec* v, gv_vec* mu_start, double tau, int denoiser, gv_vec* mu_out, int* iters) {
    gv_cg_stats st;
    std::vector<double> relres(CG_max_iter > 0 ? CG_max_iter : 1);
    ck(gv_cg_solve(ctx, v, mu_start, tau, gam2, denoiser, CG_max_iter, mu_out, &st, relres.data()), "gv_cg_solve");
    if (verbose && rank == 0)
        for (int i = 0; i < st.n_relres; i++)
            printf("[CG] it = %d: ||r_it|| / ||RHS|| = %.10g\n", i, relres[i]);
    if (iters) *iters = st.iters;
    return 0;
}

std::vector<double> vamp::infere(data* dataset) {
    // products cached from one iteration to the next belong to this run's design matrix and probe
    have_ata_u = have_ata_x2 = have_aat_prev = false;
    for (size_t i = 0; i < vars.size(); i++) vars[i] *= N;   // design matrix is scaled by 1/sqrt(N) (vamp.cpp:154-155)
    if (reverse == 1) {                                       // vamp.cpp:169-170
        ctx = dataset->get_ctx();
        ck(gv_people_stats(ctx, nullptr, nullptr, nullptr), "gv_people_stats (--use-XXT-denoiser needs --kernel-mode 0 or both layouts)");
    }
    if (!strcmp(model.c_str(), "linear")) return infere_linear(dataset);
    if (!strcmp(model.c_str(), "bin_class")) {
        return infere_bin_class(dataset);
    }
    throw "invalid model specification!";   // vamp.cpp:180 (bin_class / robust: SURVEY 8f)
}

// vamp.cpp:871-889: u in {+-1/sqrt(Mt)}^M from mt19937{seed + S} + bernoulli(0.5) on the host (bit-comparable with the
// reference at matched shard boundaries), CG on the device.
void vamp::draw_onsager_probe(data* dataset) {
    // the generator is re-seeded with the same seed + S on every call (:875): the probe is the same vector every
    // iteration, so it is drawn and uploaded once per (seed, S) and stays resident
    const long unsigned int key = seed + (long unsigned int)dataset->get_S();
    if (have_probe && probe_key == key) return;
    have_probe = true;
    probe_key = key;
    have_ata_u = false;         // A^T A u belongs to the probe
    std::mt19937 rd{key};
    std::bernoulli_distribution bern(0.5);
    std::vector<double> u(M > 0 ? M : 0);
    for (int i = 0; i < M; i++) u[i] = (2 * bern(rd) - 1) / sqrt(Mt);
    ck(gv_vec_upload(ctx, bern_vec, u.data()), "gv_vec_upload");
}

// A^T A u was captured as (diag / tau) d - (gam2 / tau) u from the first operator application of an Onsager solve: a difference
// that loses log10(gam2 / (tau |A^T A u|)) digits.  It is kept for the following iterations only if that loss is below three
// digits; otherwise the next solve captures it again.  (At the usual start, gam1 = 1e-8, gam2 / tau of the first iteration is in the
// thousands: a product kept from there left x_hat 4e-13 from the reference sequence instead of 5e-15.)
bool vamp::probe_product_is_usable(double tau, double gam2_) {
    const double nrm = sqrt(dotM(ata_u, ata_u));         // |u| = 1
    return std::isfinite(nrm) && gam2_ <= 1e3 * tau * nrm;
}

double vamp::g2d_onsager(double gam2_, double tau, data* dataset, int* iters) {
    draw_onsager_probe(dataset);
    cg(bern_vec, nullptr, tau, 0, invQ_bern_vec, iters);
    return gam2_ * dotM(bern_vec, invQ_bern_vec);
}

// The two CG solves of one iteration share the operator (tau A^T A + gam2 I): gv_cg_solve2 applies it to both pending
// directions in one two-vector Ax + ATx pair per round.  Iterates, stopping rules and results per solve are unchanged.
double vamp::fused_solves(gv_vec* v, gv_vec* mu_start, double tau, data* dataset, int* cg_iters, int* onsager_iters,
                          gv_vec* ride_x, gv_vec* ride_out, bool defer_dot) {
    draw_onsager_probe(dataset);
    gv_cg_stats sa, sb;
    std::vector<double> ra(CG_max_iter > 0 ? CG_max_iter : 1), rb(CG_max_iter > 0 ? CG_max_iter : 1);
    gv_cg_extras ex{};
    have_derived = false;
    if (fuse_solves >= 2) {
        // --fuse-solves 2: products the solves make available without a further pass over the shard (gvamp.h, gv_cg_extras)
        if (!ax2_der) ck(gv_vec_alloc(ctx, GV_SPACE_N, &ax2_der), "gv_vec_alloc");
        if (!ata_der) ck(gv_vec_alloc(ctx, GV_SPACE_M, &ata_der), "gv_vec_alloc");
        ex.ride_x = ride_x;
        ex.ride_out = ride_out;
        ex.a_mu_a = ax2_der;        // A x2_hat          (vamp.cpp:897, vamp_probit.cpp:555)
        ex.ata_mu_b = ata_der;      // A^T A invQ u      (vamp.cpp:913-914)
        have_derived = true;
    }
    gv_cg_warm wm{};
    if (fuse_solves >= 3 && warm_chain) {
        // --fuse-solves 3: the warm start of the next iteration is this solve's x2_hat (mu_CG_last, vamp.cpp:1226), and
        // Q x2_hat = v - r holds for the final residual -- so the opening r = v - Q mu_start of precondCG_solver
        // (vamp.cpp:1142-1145, one Ax + one ATx) is formed from the product the previous solve left, and A x2_hat keeps
        // accumulating in place in ax2_der.  8 passes per iteration instead of 10 when the CG runs 4 steps.
        if (!ata_x2) ck(gv_vec_alloc(ctx, GV_SPACE_M, &ata_x2), "gv_vec_alloc");
        wm.ata_mu_a = ata_x2;
        // every reanchor_every-th iteration the opening residual is formed by an explicit operator application (as at level 2):
        // A^T A x2_hat and A x2_hat restart from exact products, so their rounding never chains over more than that many solves
        if (mu_start && have_ata_x2 && !reanchor_now()) {
            wm.ata_mu_start_a = ata_x2;
            wm.a_mu_start_a = ax2_der;
        }
    }
    if (fuse_solves >= 4 && CG_max_iter > 0) {
        // --fuse-solves 4: the Onsager solve starts from zero on the same probe u every iteration: its first operator application is
        // (tau / diag) A^T A u + gam2 u / diag, with A^T A u captured the first time round -- the solve is one pass pair shorter
        if (!ata_u) ck(gv_vec_alloc(ctx, GV_SPACE_M, &ata_u), "gv_vec_alloc");
        if (reanchor_now()) have_ata_u = false;     // captured afresh from this iteration's explicit first application
        wm.ata_v_b = ata_u;
        wm.have_ata_v_b = have_ata_u ? 1 : 0;
    }
    ck(gv_cg_solve2w(ctx, v, mu_start, bern_vec, tau, gam2, CG_max_iter, x2_hat, invQ_bern_vec, &sa, &sb, ra.data(), rb.data(),
                     &ex, &wm), "gv_cg_solve2w");
    have_ata_x2 = wm.ata_mu_a != nullptr;
    probe_product_state = !wm.ata_v_b ? 0 : 3;
    if (wm.ata_v_b && !have_ata_u) {
        have_ata_u = probe_product_is_usable(tau, gam2);
        probe_product_state = have_ata_u ? 1 : 2;
    }
    if (verbose && rank == 0) {
        for (int i = 0; i < sa.n_relres; i++) printf("[CG] it = %d: ||r_it|| / ||RHS|| = %.10g\n", i, ra[i]);
        for (int i = 0; i < sb.n_relres; i++) printf("[CG onsager] it = %d: ||r_it|| / ||RHS|| = %.10g\n", i, rb[i]);
    }
    if (cg_iters) *cg_iters = sa.iters;
    if (onsager_iters) *onsager_iters = sb.iters;
    if (defer_dot) return 0.0;          // <u, invQ u> travels with the other scalars of the iteration (infere_linear)
    return gam2 * dotM(bern_vec, invQ_bern_vec);
}

// vamp.cpp:929-1072: EM on (probs, vars).  The loop over markers (:953-1013) is gv_prior_estep; its 1 + 2(L-1) sums
// travel in ONE all-reduce instead of 1 + 2(L-1) scalar ones.
void vamp::updatePrior(int verbose_) {
    double lambda = 1 - probs[0];
    std::vector<double> omegas = probs;
    for (size_t j = 1; j < omegas.size(); j++) omegas[j] /= lambda;
    int it;
    for (it = 0; it < EM_max_iter; it++) {
        const int L = (int)probs.size();
        std::vector<double> probs_prev = probs, vars_prev = vars;
        if (L >= 2) {
            std::vector<double> sums(1 + 2 * (L - 1));
            // (:990, :1012-1013: the sums over the ranks, all-reduced on the device buffer before the one read-back)
            ck(gv_prior_estep_global(ctx, r1, gam1, lambda, omegas.data(), vars.data(), L, sums.data()), "gv_prior_estep_global");
            const double sum_of_pin = sums[0];
            lambda = sum_of_pin / Mt;
            for (int j = 0; j < L - 1; j++) {
                const double res_total = sums[1 + 2 * j], res_gammas_total = sums[2 + 2 * j];
                if (learn_vars == 1) vars[j + 1] = res_gammas_total / res_total;
                omegas[j + 1] = res_total / sum_of_pin;
                probs[j + 1] = lambda * omegas[j + 1];
            }
            probs[0] = 1 - lambda;
        }
        double distance_probs = 0, norm_probs = 0, distance_vars = 0, norm_vars = 0;
        for (size_t j = 0; j < probs.size(); j++) {
            distance_probs += (probs[j] - probs_prev[j]) * (probs[j] - probs_prev[j]);
            norm_probs += probs[j] * probs[j];
            distance_vars += (vars[j] - vars_prev[j]) * (vars[j] - vars_prev[j]);
            norm_vars += vars[j] * vars[j];
        }
        const double dist_probs = sqrt(distance_probs / norm_probs), dist_vars = sqrt(distance_vars / norm_vars);
        if (verbose_ == 1 && verbose && rank == 0)
            std::cout << "it = " << it << ": dist_probs = " << dist_probs << " & dist_vars = " << dist_vars << std::endl;
        if (dist_probs < EM_err_thr && dist_vars < EM_err_thr) break;
    }
    if (verbose_ == 1 && verbose && rank == 0)
        std::cout << "Final number of prior EM iterations = " << std::min(it + 1, EM_max_iter) << " / " << EM_max_iter
                  << std::endl;
    // merging close variances (vamp.cpp:1054-1071)
    for (size_t j = 0; j < vars.size(); j++)
        for (size_t k = j + 1; k < vars.size(); k++) {
            const double denom = (vars[j] != 0) ? std::min(vars[j], vars[k]) : 1e-7;
            if (std::abs(vars[j] - vars[k]) / denom < 5e-1) {
                probs[j] += probs[k];
                vars.erase(vars.begin() + k);
                probs.erase(probs.begin() + k);
                k--;
            }
        }
}

std::vector<double> vamp::infere_linear(data* dataset) {
    ctx = dataset->get_ctx();
    S = dataset->get_S();
    auto newM = [&](gv_vec** v) { ck(gv_vec_alloc(ctx, GV_SPACE_M, v), "gv_vec_alloc"); };
    auto newN = [&](gv_vec** v) { ck(gv_vec_alloc(ctx, GV_SPACE_N, v), "gv_vec_alloc"); };
    for (gv_vec** v : {&x1_hat, &x1_hat_prev, &x2_hat, &r1, &r2, &r2_prev, &mu_CG_last, &bern_vec, &invQ_bern_vec, &vM, &tM})
        newM(v);
    for (gv_vec** v : {&z1, &y, &tN, &tN2}) newN(v);
    {   // y = filter_pheno() (vamp.cpp:219): NA / pad slots zeroed on the device
        std::vector<double> yh = dataset->get_phen();
        yh.resize(N, 0.0);
        ck(gv_set_phen(ctx, y, yh.data()), "gv_set_phen");
    }
    alpha1 = 0;
    alpha2 = 0;   // read at vamp.cpp:501 before its first assignment (:631); harmless there, 0 here
    have_ata_x2 = have_aat_prev = false;
    // A r1 is known at the start of a fresh run: r1 = 0 (a restart / an initial estimate load r1: its product is not at hand)
    have_ar1 = gam1_init == -1 && init_est != 1;
    if (ar1) ck(gv_vec_fill(ctx, ar1, 0.0), "gv_vec_fill");       // (a second run on this object)
    warm_chain = true;      // the LMMSE solves of this loop warm-start one another (infere_bin_class starts every solve from zero)
    const double sqrtN = sqrt((double)N);

    if (gam1_init != -1) {   // restart (vamp.cpp:226-233): r1 file stores r1 / sqrt(N)... and is divided once more
        gam1 = gam1_init;
        gamw = gamw_init;
        std::vector<double> r1_init = mpi_read_vec_from_file(r1_init_file, M, S);
        for (double& v : r1_init) v /= sqrtN;
        ck(gv_vec_upload(ctx, r1, r1_init.data()), "gv_vec_upload");
    }
    if (init_est == 1) {     // vamp.cpp:244-258
        const size_t pos_dot = estimate_file.find(".");
        std::vector<double> x_est = (estimate_file.substr(pos_dot + 1) == "bin") ? mpi_read_vec_from_file(estimate_file, M, S)
                                                                                  : read_vec_from_file(estimate_file, M, S);
        x_est.resize(M, 0.0);
        for (double& v : x_est) v *= sqrtN;
        ck(gv_vec_upload(ctx, x1_hat, x_est.data()), "gv_vec_upload");
        ck(gv_vec_upload(ctx, r1, x_est.data()), "gv_vec_upload");
    }
    const std::string pre = out_dir + out_name;
    std::vector<double> x1_hat_stored(M > 0 ? M : 0, 0.0);
    bool ran_any = false;
    if (use_freeze == 1) {                                                  // vamp.cpp:205-209
        std::vector<double> freeze_ind = read_vec_from_file(freeze_index_file, M, S);
        freeze_ind.resize(M > 0 ? M : 0, 0.0);
        std::vector<double> u(freeze_ind.size()), f(freeze_ind.size());
        for (size_t i = 0; i < freeze_ind.size(); i++) {
            u[i] = freeze_ind[i] == 0 ? 1.0 : 0.0;
            f[i] = 1.0 - u[i];
        }
        for (gv_vec** v : {&unfrozen, &frozen, &dvec})
            if (!*v) newM(v);
        ck(gv_vec_upload(ctx, unfrozen, u.data()), "gv_vec_upload");
        ck(gv_vec_upload(ctx, frozen, f.data()), "gv_vec_upload");
    }

    // the reference's named MPI_Wtime brackets (vamp.cpp:425-433,:515-530,:557,:617-637,:733-736), printed by rank 0 when
    // verbose: the device is drained before each reading so that a phase is charged with its own kernels
    auto tick = [&]() -> double {
        if (!(verbose && rank == 0)) return 0.0;
        ck(gv_synchronize(ctx), "gv_synchronize");
        return now_s();
    };
    for (int it = 1; it <= max_iter; it++) {
        cur_it = it;
        probe_product_state = 0;
        const double t_start = now_s();
        double t_io = 0;
        const double t_denoising = tick();
        gv_counters c0;
        gv_get_counters(ctx, &c0);
        vamp_iter_stats st;
        memset(&st, 0, sizeof(st));
        if (verbose && rank == 0)
            std::cout << std::endl << "********************" << std::endl << "iteration = " << it << std::endl
                      << "********************" << std::endl << "->DENOISING" << std::endl;

        ck(gv_vec_copy(ctx, x1_hat_prev, x1_hat), "gv_vec_copy");
        probs_before = probs;
        vars_before = vars;
        const double alpha1_prev = alpha1;
        int it_revar = 1;
        for (; it_revar <= auto_var_max_iter; it_revar++) {               // vamp.cpp:289-338
            double sums[2];
            gv_vec* dout = use_freeze == 1 ? dvec : nullptr;
            // (:313 + the l2_norm2 of :326 in one message: summed over the ranks on the device before the one read-back -- except
            // with a freeze mask, where the first sum is replaced by a local one before the exchange)
            auto denoise = use_freeze == 1 ? gv_denoise : gv_denoise_global;
            if (it == 1 && init_est == 1) {                               // :295-296: x1_hat = r1, g1d still evaluated
                ck(denoise(ctx, r1, gam1, probs.data(), vars.data(), (int)probs.size(), tM, dout, sums), "gv_denoise");
                ck(gv_vec_copy(ctx, x1_hat, r1), "gv_vec_copy");
                sums[1] = 0.0;
            } else
                ck(denoise(ctx, r1, gam1, probs.data(), vars.data(), (int)probs.size(), x1_hat, dout, sums), "gv_denoise");
            if (use_freeze == 1) {
                ck(gv_vec_dot(ctx, dvec, unfrozen, 0, &sums[0]), "gv_vec_dot");   // :308: unfrozen markers only
                ck(gv_allreduce_host(ctx, sums, 2), "gv_allreduce_host");
            }
            alpha1 = sums[0] / Mt;
            eta1 = gam1 / alpha1;
            if (it <= 1) break;
            const double gam1_reEst_prev = gam1;
            gam1 = std::min(std::max(1.0 / (1.0 / eta1 + sums[1] / Mt), gamma_min), gamma_max);   // :326
            updatePrior(0);
            if (verbose && rank == 0) std::cout << "[old] it_revar = " << it_revar << ": gam1 = " << gam1 << std::endl;
            if (std::abs(gam1 - gam1_reEst_prev) < 1e-3) break;
        }
        gam1s.push_back(gam1);
        st.gam1_denoise = gam1;
        st.revar_rounds = std::max(it_revar - 1, 1);
        if (verbose && rank == 0)
            std::cout << "A total of " << st.revar_rounds << " variance and prior tuning iterations were performed" << std::endl;

        if (it > 1 && use_freeze == 1) {                                  // damping of the unfrozen markers only (:353)
            ck(gv_vec_axpby(ctx, tM, rho, x1_hat, 1 - rho, x1_hat_prev), "gv_vec_axpby");
            ck(gv_vec_mul(ctx, tM, tM, unfrozen), "gv_vec_mul");          // exact: the masks are 0 / 1
            ck(gv_vec_mul(ctx, x1_hat, x1_hat, frozen), "gv_vec_mul");
            ck(gv_vec_axpby(ctx, x1_hat, 1.0, x1_hat, 1.0, tM), "gv_vec_axpby");
            alpha1 = rho * alpha1 + (1 - rho) * alpha1_prev;
        } else if (it > 1) {                                              // damping (:348-414)
            ck(gv_vec_axpby(ctx, x1_hat, rho, x1_hat, 1 - rho, x1_hat_prev), "gv_vec_axpby");
            alpha1 = rho * alpha1 + (1 - rho) * alpha1_prev;
        }
        // z1 = A x1_hat (:429).  --fuse-solves 2: it is only printed / stored, so it rides in a free slot of the CG passes
        // below and its outputs follow the solve.
        // --use-XXT-denoiser 1 with --fuse-solves >= 1: it shares the pass of A r2 (denoiserXXT.cpp:41) instead.
        const bool z1_rides = (fuse_solves >= 2 && reverse == 0) || (fuse_solves >= 1 && reverse == 1);
        if (!z1_rides) {
            const double tz1 = tick();
            ck(gv_ax_dev(ctx, x1_hat, z1), "gv_ax_dev");
            if (verbose && rank == 0) std::cout << "time needed to calculate z1 = " << tick() - tz1 << " seconds" << std::endl;
        }

        double t0 = now_s();
        auto z1_outputs_a = [&]() {
            if (store_iterates && rank == 0) {                            // :435-436 (every rank in the reference)
                std::vector<double> zh(4 * dataset->get_mbytes());
                ck(gv_vec_download(ctx, z1, zh.data()), "gv_vec_download");
                store_vec_to_file(pre + "_z1_it_" + std::to_string(it) + ".csv", zh);
            }
        };
        if (!z1_rides) z1_outputs_a();
        if (verbose && rank == 0) std::cout << "rho = " << rho << std::endl;
        if (store_iterates || keep_history) {   // otherwise only the last iterate is wanted: fetched after the loop
            ck(gv_vec_download(ctx, x1_hat, x1_hat_stored.data()), "gv_vec_download");
            for (double& v : x1_hat_stored) v /= sqrtN;
            if (store_iterates) mpi_store_vec_to_file(pre + "_it_" + std::to_string(it) + ".bin", x1_hat_stored, S, M);
            if (keep_history) x1_hist.push_back(x1_hat_stored);
        }
        ran_any = true;
        store_scaled(pre + "_r1_it_" + std::to_string(it) + ".bin", r1, &r1_hist);
        t_io += now_s() - t0;

        gam_before = gam2;
        gam2 = std::min(std::max(eta1 - gam1, gamma_min), gamma_max);    // :472
        if (verbose && rank == 0) std::cout << "eta1 = " << eta1 << std::endl << "gam2 = " << gam2 << std::endl;
        ck(gv_vec_copy(ctx, r2_prev, r2), "gv_vec_copy");                 // :483
        ck(gv_vec_axpby(ctx, r2, eta1 / gam2, x1_hat, -gam1 / gam2, r1), "gv_vec_axpby");   // :485-486
        r2_c1 = eta1 / gam2;
        r2_c2 = gam1 / gam2;
        if (use_lmmse_damp == 1 && it > 1) {                              // :488-498
            const double xi = std::min(2 * rho, 1.0);
            gam2 = 1.0 / pow(xi / sqrt(gam2) + (1 - xi) / sqrt(gam_before), 2);
        }
        rho = std::max(rho, std::min(2 * std::min(alpha1, alpha2), 1.0));  // :501-502
        st.alpha1 = alpha1; st.eta1 = eta1; st.gam2 = gam2; st.rho = rho;

        if (auto_var_max_iter == 0 || it <= 1) updatePrior(1);            // :518-519
        // ---- the scalars of the iteration's second half -- R2 of z1 (:1295-1317), <u, invQ u> (:631), |x2 - r2|^2 (:686-693), the two
        // sums of updateNoisePrec (:892-927) and the stopping rule (:741-749) -- do not depend on one another: they are gathered and
        // travel in ONE launch and ONE read-back (gv_vec_dots_ex) instead of six; each is bit-identical to its own axpby + dot
        struct { gv_dot_spec spec[8]; int* slot[8]; int n = 0; } tail;
        auto want = [&](int* slot, const gv_vec* xa, const gv_vec* xb, const gv_vec* ya, const gv_vec* yb, int sync) {
            if (tail.n >= 8) throw std::runtime_error("infere_linear: more than eight gathered inner products (gv_vec_dots_ex takes 8)");
            tail.spec[tail.n] = gv_dot_spec{xa, xb, ya, yb, sync};
            tail.slot[tail.n++] = slot;
        };
        int k_z1 = -1, k_yy = -1, k_alpha2 = -1, k_gam2 = -1, k_temp = -1, k_trace = -1, k_stop = -1, k_stop_n = -1;
        auto z1_r2 = [&](double num, double den) {   // err_measures(1) (:1295-1317): R2 of the cached z1
            const double e = sqrt(num / den);
            st.R2_denoise = 1 - e * e;
            R2trains.push_back(st.R2_denoise);
            if (verbose && rank == 0) std::cout << "R2 = " << st.R2_denoise << std::endl;
        };
        auto z1_outputs_b = [&]() {
            if (z1_rides) {                   // z1 comes out of the solve: with the tail
                want(&k_z1, y, z1, y, z1, 0);
                want(&k_yy, y, nullptr, y, nullptr, 0);
                return;
            }
            ck(gv_vec_axpby(ctx, tN, 1.0, y, -1.0, z1), "gv_vec_axpby");
            const gv_vec* xs[2] = {tN, y};
            const gv_vec* ys[2] = {tN, y};
            double d2[2];
            ck(gv_vec_dots(ctx, 2, xs, ys, 0, d2), "gv_vec_dots");
            z1_r2(d2[0], d2[1]);
        };
        if (!z1_rides) z1_outputs_b();
        if (verbose && rank == 0) {
            std::cout << "prior variances = ";
            for (double v : vars) std::cout << v << ' ';
            std::cout << std::endl << "prior probabilities = ";
            for (double p : probs) std::cout << p << ' ';
            std::cout << std::endl << "gamw = " << gamw << std::endl;
        }
        t0 = now_s();
        store_scaled(pre + "_r2_it_" + std::to_string(it) + ".bin", r2, nullptr);
        t_io += now_s() - t0;

        if (verbose && rank == 0) std::cout << "denoising step took " << tick() - t_denoising << " seconds." << std::endl;

        // ---- LMMSE step (:547-620)
        if (verbose && rank == 0) std::cout << "______________________" << std::endl << "->LMMSE" << std::endl;
        const double t_lmmse = tick(), t_cg = t_lmmse;
        if (reverse == 0) {
            if (!have_aty) {                                               // A^T y (:588) does not change: once
                if (!aty) ck(gv_vec_alloc(ctx, GV_SPACE_M, &aty), "gv_vec_alloc");
                ck(gv_atx_dev(ctx, y, aty), "gv_atx_dev");
                have_aty = true;
            }
            ck(gv_vec_axpby(ctx, vM, gamw, aty, gam2, r2), "gv_vec_axpby");   // v = gamw A^T y + gam2 r2 (:590-591)
            gv_vec* warm = (it == 1 || !have_mu_CG_last) ? nullptr : mu_CG_last;
            if (fuse_solves) {
                fused_solves(vM, warm, gamw, dataset, &st.cg_iters, &st.onsager_iters, z1_rides ? x1_hat : nullptr,
                             z1_rides ? z1 : nullptr, true);
                want(&k_alpha2, bern_vec, nullptr, invQ_bern_vec, nullptr, 1);
                if (z1_rides) {
                    double tz = now_s();
                    z1_outputs_a();
                    t_io += now_s() - tz;
                    z1_outputs_b();
                }
            } else
                cg(vM, warm, gamw, 1, x2_hat, &st.cg_iters);               // :593-596
            ck(gv_vec_copy(ctx, mu_CG_last, x2_hat), "gv_vec_copy");       // :1225-1226
            have_mu_CG_last = true;
        } else if (fuse_solves) {
            // lmmse_denoiserAAT (denoiserXXT.cpp:37-50) and g2d_onsager (:631) on shared passes (gv_cg_solve_aat2):
            // (gamw A A^T + gam2 I) u = y - A r2 ; x2 = r2 + gamw A^T u ; alpha2 from (gamw A^T A + gam2 I) w = probe
            if (!mu_CG_last_N) ck(gv_vec_alloc(ctx, GV_SPACE_N, &mu_CG_last_N), "gv_vec_alloc");
            if (!ax2_der) ck(gv_vec_alloc(ctx, GV_SPACE_N, &ax2_der), "gv_vec_alloc");
            // level 4: A r2 and z1 are taken inside the joint solve (gv_aat_warm.pre_x / ride_x) -- A r2 in the pass that carries
            // the first half-application of the Onsager solve, so that the two solves run in phase, z1 in a free slot
            const bool inside = fuse_solves >= 4;
            // ... and with A r1 at hand (carried from the previous iteration, 0 at the start) that pass carries z1 = A x1_hat, not A r2:
            // A r2 = c1 z1 - c2 A r1 by linearity.  One pass fewer per iteration whenever the two solves finish in the same step
            // (z1 then found no free slot and took a pass of its own).  Every reanchor_every-th iteration takes the explicit products.
            for (gv_vec** v : {&ar1, &ar2})
                if (inside && !*v) ck(gv_vec_alloc(ctx, GV_SPACE_N, v), "gv_vec_alloc");       // (zero-filled: A r1 of r1 = 0)
            // The combination cancels: c1 - c2 = 1 while c1 + c2 = (eta1 + gam1) / (eta1 - gam1) = (1 + alpha1) / (1 - alpha1), and
            // the rounding of z1 and A r1 is amplified by that factor in y - A r2; the A r1 carried from the previous iteration is
            // (eta2 A x2_hat - gam2 A r2) / gam1 with gam1 = eta2 - gam2: (1 + alpha2) / (1 - alpha2).  Measured (scripts/drift_check.py,
            // profiles/r6_fuse_level_drift.json): the first stays at 1.06-1.8; the second is ~1.5e4 after iteration 1 (gam1 starts at
            // 1e-8: alpha2 ~ 1) -- that one product put 1.5e-13 into iteration 2 of round 5's LD run where the explicit form has 2e-15 --
            // and 5-90 afterwards.  Beyond linearity_max (100: two digits) the iteration takes A r2 explicitly (pre_x = r2, z1
            // riding), as on a re-anchoring iteration; have_ar1 carries the test for the combination that formed A r1.
            const bool by_linearity = inside && have_ar1 && !reanchor_now() && std::fabs(r2_c1) + std::fabs(r2_c2) <= linearity_max;
            auto z1_outputs = [&]() {
                double tz = now_s();
                z1_outputs_a();
                t_io += now_s() - tz;
                z1_outputs_b();
            };
            if (by_linearity)
                ck(gv_vec_axpby(ctx, tN, 1.0, y, r2_c2, ar1), "gv_vec_axpby");   // the solver turns y + c2 A r1 into y - A r2 (- c1 z1)
            else if (inside)
                ck(gv_vec_copy(ctx, tN, y), "gv_vec_copy");                // the solver turns it into y - A r2
            else {
                ck(gv_ax2_dev(ctx, x1_hat, r2, z1, ax2_der), "gv_ax2_dev");   // z1 = A x1_hat (:429) and A r2, one pass
                z1_outputs();
                ck(gv_vec_axpby(ctx, tN, 1.0, y, -1.0, ax2_der), "gv_vec_axpby");
            }
            draw_onsager_probe(dataset);
            have_derived = fuse_solves >= 2;
            if (have_derived) {
                if (!aat_der) ck(gv_vec_alloc(ctx, GV_SPACE_N, &aat_der), "gv_vec_alloc");
                if (!ata_der) ck(gv_vec_alloc(ctx, GV_SPACE_M, &ata_der), "gv_vec_alloc");
            }
            gv_cg_stats sa, sb;
            std::vector<double> ra(CG_max_iter > 0 ? CG_max_iter : 1), rb(CG_max_iter > 0 ? CG_max_iter : 1);
            // --fuse-solves 3: A A^T u of the previous iteration (aat_der, from that solve's final residual) gives the initial
            // residual of this warm start without the ATx + Ax pair of denoiserXXT.cpp:76-78 (gvamp.h: gv_cg_solve_aat2w)
            // and A^T u (for x2_hat = r2 + gamw A^T u, denoiserXXT.cpp:46-48) is accumulated from the A^T p_k of the solve
            // instead of taking a closing ATx pass: it lives in at_u across the iterations
            const bool known_start = fuse_solves >= 3 && it > 1 && have_aat_prev && !reanchor_now();
            gv_aat_warm wm{};
            gv_vec* at_out = tM;
            if (fuse_solves >= 3) {
                if (!at_u) ck(gv_vec_alloc(ctx, GV_SPACE_M, &at_u), "gv_vec_alloc");
                at_out = at_u;
                wm.accumulate_at_mu_a = 1;
                if (known_start) { wm.aat_mu_start_a = aat_der; wm.at_mu_start_a = at_u; }
                if (by_linearity) { wm.pre_x = x1_hat; wm.pre_out = z1; wm.pre_scale = r2_c1; }
                else if (inside) { wm.pre_x = r2; wm.pre_out = ax2_der; wm.ride_x = x1_hat; wm.ride_out = z1; }
                if (fuse_solves >= 4 && CG_max_iter > 0) {      // A^T A u of the probe, as in fused_solves
                    if (!ata_u) ck(gv_vec_alloc(ctx, GV_SPACE_M, &ata_u), "gv_vec_alloc");
                    if (reanchor_now()) have_ata_u = false;
                    wm.ata_v_b = ata_u;
                    wm.have_ata_v_b = have_ata_u ? 1 : 0;
                }
            }
            ck(gv_cg_solve_aat2w(ctx, tN, it == 1 ? nullptr : mu_CG_last_N, bern_vec, gamw, gam2, CG_max_iter, tN2, at_out,
                                 invQ_bern_vec, &sa, &sb, ra.data(), rb.data(), have_derived ? aat_der : nullptr,
                                 have_derived ? ata_der : nullptr, &wm), "gv_cg_solve_aat2w");
            have_aat_prev = have_derived;
            probe_product_state = !wm.ata_v_b ? 0 : 3;
            if (wm.ata_v_b && !have_ata_u) {
                have_ata_u = probe_product_is_usable(gamw, gam2);
                probe_product_state = have_ata_u ? 1 : 2;
            }
            if (by_linearity) ck(gv_vec_axpby(ctx, ax2_der, r2_c1, z1, -r2_c2, ar1), "gv_vec_axpby");   // A r2
            if (inside) {
                ck(gv_vec_copy(ctx, ar2, ax2_der), "gv_vec_copy");         // (kept: ax2_der becomes A x2_hat below)
                z1_outputs();
            }
            st.cg_iters = sa.iters;
            st.onsager_iters = sb.iters;
            if (verbose && rank == 0) {
                for (int i = 0; i < sa.n_relres; i++) printf("[CG] it = %d: ||r_it|| / ||RHS|| = %.10g\n", i, ra[i]);
                for (int i = 0; i < sb.n_relres; i++) printf("[CG onsager] it = %d: ||r_it|| / ||RHS|| = %.10g\n", i, rb[i]);
            }
            ck(gv_vec_copy(ctx, mu_CG_last_N, tN2), "gv_vec_copy");
            ck(gv_vec_axpby(ctx, x2_hat, gamw, at_out, 1.0, r2), "gv_vec_axpby");
            want(&k_alpha2, bern_vec, nullptr, invQ_bern_vec, nullptr, 1);
            // A x2_hat = A r2 + gamw A A^T u, the latter from the residual of the N-space solve (no pass)
            if (have_derived) ck(gv_vec_axpby(ctx, ax2_der, 1.0, ax2_der, gamw, aat_der), "gv_vec_axpby");
        } else {
            // lmmse_denoiserAAT (denoiserXXT.cpp:37-50): (gamw A A^T + gam2 I) u = y - A r2 ; x2 = r2 + gamw A^T u
            if (!mu_CG_last_N) ck(gv_vec_alloc(ctx, GV_SPACE_N, &mu_CG_last_N), "gv_vec_alloc");
            ck(gv_ax_dev(ctx, r2, tN), "gv_ax_dev");
            ck(gv_vec_axpby(ctx, tN, 1.0, y, -1.0, tN), "gv_vec_axpby");
            gv_cg_stats cst;
            std::vector<double> relres(CG_max_iter > 0 ? CG_max_iter : 1);
            ck(gv_cg_solve_aat(ctx, tN, it == 1 ? nullptr : mu_CG_last_N, gamw, gam2, CG_max_iter, tN2, &cst, relres.data()),
               "gv_cg_solve_aat");
            st.cg_iters = cst.iters;
            if (verbose && rank == 0)
                for (int i = 0; i < cst.n_relres; i++) printf("[CG] it = %d: ||r_it|| / ||RHS|| = %.10g\n", i, relres[i]);
            ck(gv_vec_copy(ctx, mu_CG_last_N, tN2), "gv_vec_copy");
            ck(gv_atx_dev(ctx, tN2, x2_hat), "gv_atx_dev");
            ck(gv_vec_axpby(ctx, x2_hat, gamw, x2_hat, 1.0, r2), "gv_vec_axpby");
        }
        t0 = now_s();
        store_scaled(pre + "_it_" + std::to_string(it) + "_x2_hat.bin", x2_hat, &x2_hist);
        t_io += now_s() - t0;

        if (verbose && rank == 0)
            std::cout << (fuse_solves ? "CG + onsager (shared passes) took " : "CG took ") << tick() - t_cg << " seconds."
                      << std::endl;
        const double t_ons = tick();
        const double gam2_solve = gam2;                                    // alpha2 = gam2 <u, invQ u> (:631) with the gam2 of the solves
        if (!fuse_solves) {
            alpha2 = g2d_onsager(gam2, gamw, dataset, &st.onsager_iters);
            if (verbose && rank == 0) std::cout << "onsager took " << tick() - t_ons << " seconds." << std::endl;
        }
        double onsager_approx = 0;
        if (it > 1 && diagnostics) {
            // "polynomial onsager" diagnostics (:646-681): printed only; off by default (they cost 3 Ax per iteration)
            ck(gv_vec_axpby(ctx, tM, 1.0, r2, -1.0, r2_prev), "gv_vec_axpby");
            onsager_approx = dotM(x2_hat, tM) / dotM(r2, tM);
            ck(gv_ax_dev(ctx, r2, tN), "gv_ax_dev");
            ck(gv_ax_dev(ctx, x2_hat, tN2), "gv_ax_dev");
            ck(gv_ax_dev(ctx, x2_hat, tN2), "gv_ax_dev");
        }
        // updateNoisePrec (:892-927): A x2_hat (:897) and A^T A invQ u (:913-914) -- by-products of the solves, or one Ax pass for
        // both and one ATx pass.  temp = A x2_hat - y is also what err_measures(2) recomputes (:1301-1314): its R2 is taken here.
        const gv_vec *ax2 = ax2_der, *ata = ata_der;
        if (!have_derived) {
            ck(gv_ax2_dev(ctx, x2_hat, invQ_bern_vec, tN, tN2), "gv_ax2_dev");
            ck(gv_atx_dev(ctx, tN2, tM), "gv_atx_dev");
            ax2 = tN;
            ata = tM;
        }
        const bool reest = auto_var_max_iter >= 1 && it > 2;               // :686-693
        if (reest) want(&k_gam2, x2_hat, r2, x2_hat, r2, 1);
        want(&k_temp, ax2, y, ax2, y, 0);
        {
            bool have_yy = false;                                          // (already asked for with the R2 of z1)
            for (int k = 0; k < tail.n; k++) have_yy = have_yy || tail.slot[k] == &k_yy;
            if (!have_yy) want(&k_yy, y, nullptr, y, nullptr, 0);
        }
        want(&k_trace, bern_vec, nullptr, ata, nullptr, 1);
        want(&k_stop, x1_hat_prev, x1_hat, x1_hat_prev, x1_hat, 1);        // stopping criterion (:741-749)
        want(&k_stop_n, x1_hat_prev, nullptr, x1_hat_prev, nullptr, 1);
        double tv[8];
        {   // the sums every rank holds whole (N-space) first, then the ones summed over the ranks: one all-reduce
            gv_dot_spec ordered[8];
            int no = 0;
            for (int pass = 0; pass < 2; pass++)
                for (int k = 0; k < tail.n; k++)
                    if ((tail.spec[k].sync != 0) == (pass == 1)) {
                        ordered[no] = tail.spec[k];
                        *tail.slot[k] = no++;
                    }
            ck(gv_vec_dots_ex(ctx, no, ordered, tv), "gv_vec_dots_ex");
        }
        if (k_z1 >= 0) z1_r2(tv[k_z1], tv[k_yy]);
        if (k_alpha2 >= 0) alpha2 = gam2_solve * tv[k_alpha2];             // :631, solved together with x2_hat above
        st.alpha2 = alpha2;
        if (verbose && rank == 0) std::cout << "alpha2 = " << alpha2 << std::endl;
        if (it > 1 && diagnostics && rank == 0) std::cout << "onsager approx = " << onsager_approx << std::endl;
        eta2 = gam2 / alpha2;                                              // :682
        if (reest) gam2 = std::min(std::max(1 / (1 / eta2 + tv[k_gam2] / Mt), gamma_min), gamma_max);
        gam2s.push_back(gam2);
        st.eta2 = eta2; st.gam2_reest = gam2;
        gam1 = std::min(std::max(eta2 - gam2, gamma_min), gamma_max);     // :702
        ck(gv_vec_axpby(ctx, r1, eta2 / gam1, x2_hat, -gam2 / gam1, r2), "gv_vec_axpby");   // :706-707
        if (reverse == 1 && fuse_solves >= 4 && ar1 && have_derived) {   // A r1 of the next iteration, from A x2_hat and A r2
            ck(gv_vec_axpby(ctx, ar1, eta2 / gam1, ax2_der, -gam2 / gam1, ar2), "gv_vec_axpby");
            // ... which cancels the same way: coefficients (eta2, gam2) / gam1 with gam1 = eta2 - gam2, amplification
            // (1 + alpha2) / (1 - alpha2).  Beyond linearity_max the carried A r1 is not used: the next iteration takes A r2
            // explicitly, and the chain starts afresh from that product.
            have_ar1 = (std::fabs(eta2) + std::fabs(gam2)) <= linearity_max * std::fabs(gam1);
        }
        st.gam1_next = gam1;
        if (verbose && rank == 0) std::cout << "gam2 re-est = " << gam2 << std::endl << "gam1 = " << gam1 << std::endl;

        {   // :726 (+ err_measures(2), :731)
            const double temp_norm2 = tv[k_temp], trace_corr = tv[k_trace] * Mt;
            if (verbose && rank == 0) {
                std::cout << "l2_norm2(temp) / N = " << temp_norm2 / N << std::endl;
                std::cout << "trace_correction / N = " << trace_corr / N << std::endl;
            }
            gamw = (double)N / (temp_norm2 + trace_corr);
            const double l2_pred_err = sqrt(temp_norm2 / tv[k_yy]);
            st.R2_lmmse = 1 - l2_pred_err * l2_pred_err;
        }
        R2trains.push_back(st.R2_lmmse);
        st.gamw = gamw;
        if (verbose && rank == 0) std::cout << "R2 = " << st.R2_lmmse << std::endl << "gamw = " << gamw << std::endl;
        if (verbose && rank == 0) std::cout << "lmmse step took " << tick() - t_lmmse << " seconds." << std::endl;

        st.L_after = (int)probs.size();
        st.probe_product = probe_product_state;
        gv_counters c1;
        gv_get_counters(ctx, &c1);
        st.n_ax = (long)(c1.n_ax - c0.n_ax);
        st.n_atx = (long)(c1.n_atx - c0.n_atx);
        st.n_ax_pass = (long)(c1.n_ax_pass - c0.n_ax_pass);
        st.n_atx_pass = (long)(c1.n_atx_pass - c0.n_atx_pass);

        st.seconds_io = t_io;
        st.seconds = now_s() - t_start - t_io;
        stats.push_back(st);
        if (it > 1 && sqrt(tv[k_stop] / tv[k_stop_n]) < stop_criteria_thr) {
            if (verbose && rank == 0)
                std::cout << "VAMP stopping criteria fulfilled with threshold = " << stop_criteria_thr << "." << std::endl;
            break;
        }
        if (verbose && rank == 0)
            std::cout << "total iteration time = " << st.seconds << " (Ax " << st.n_ax << ", ATx " << st.n_atx << ")"
                      << std::endl << std::endl;
    }
    if (ran_any && !(store_iterates || keep_history)) {   // x1_hat still holds the last iteration's (damped) estimate
        ck(gv_vec_download(ctx, x1_hat, x1_hat_stored.data()), "gv_vec_download");
        for (double& v : x1_hat_stored) v /= sqrtN;
    }
    if (store_pvals == 1) {                                                 // vamp.cpp:761-776
        // z1 / x1_hat of the last completed iteration, y = filtered phenotype (all still resident on the device)
        const double t0 = now_s();
        std::vector<double> pv = dataset->pvals_calc_dev(z1, y, x1_hat, false);
        mpi_store_vec_to_file(pre + "_pvals.bin", pv, S, M);
        if (verbose && rank == 0) std::cout << "filepath_out_pvals = " << pre + "_pvals.bin" << std::endl;
        if (dataset->get_bimfp() != "") {
            std::vector<double> pl = dataset->pvals_calc_dev(z1, y, x1_hat, true, pre);   // + <pre>_LOCO_chr_<ch>.csv (data.cpp:1276-1281)
            mpi_store_vec_to_file(pre + "_pvals_LOCO.bin", pl, S, M);
            if (verbose && rank == 0) std::cout << "filepath_out_pvals_LOCO = " << pre << std::endl;
        }
        if (verbose && rank == 0) std::cout << "p-values took " << now_s() - t0 << " seconds." << std::endl;
    }
    if (store_iterates && rank == 0) {                                      // :779-794 (rank 0 only, App. B)
        store_vec_to_file(pre + "_gam1s.csv", gam1s);
        store_vec_to_file(pre + "_gam2s.csv", gam2s);
        store_vec_to_file(pre + "_R2trains.csv", R2trains);
    }
    return x1_hat_stored;   // x1_hat / sqrt(N) of the last iteration (:802)
}

// vamp::infere_bin_class (vamp_probit.cpp:20-658): generalised VAMP for y in {0, 1}, with or without covariates.
// Signal side shared with the linear model (gv_denoise, updatePrior, gv_cg_solve, g2d_onsager); the z side is
// gv_probit_denoise over the N individuals.  probit_err_measures and the "true ..." prints of the reference are
// diagnostics against a known signal and are not reproduced.
std::vector<double> vamp::infere_bin_class(data* dataset) {
    ctx = dataset->get_ctx();
    S = dataset->get_S();
    warm_chain = false;     // every LMMSE solve of this loop starts from zero (vamp_probit.cpp:497)
    auto newM = [&](gv_vec** v) { ck(gv_vec_alloc(ctx, GV_SPACE_M, v), "gv_vec_alloc"); };
    auto newN = [&](gv_vec** v) { ck(gv_vec_alloc(ctx, GV_SPACE_N, v), "gv_vec_alloc"); };
    for (gv_vec** v : {&x1_hat, &x1_hat_prev, &x2_hat, &r1, &r2, &bern_vec, &invQ_bern_vec, &vM, &tM}) newM(v);
    gv_vec *p1 = nullptr, *p2 = nullptr, *z1_hat = nullptr;
    for (gv_vec** v : {&y, &tN, &tN2, &p1, &p2, &z1_hat}) newN(v);
    {
        std::vector<double> yh = dataset->get_phen();
        yh.resize(N, 0.0);
        ck(gv_set_phen(ctx, y, yh.data()), "gv_set_phen");
        // p1 = simulate(N, {1}, {1}) with the default seed 1 (vamp_probit.cpp:52, utilities.hpp:23)
        std::vector<double> p1h = simulate(N, std::vector<double>{1.0 / 1}, std::vector<double>{1}, 1);
        p1h.resize(4 * dataset->get_mbytes(), 0.0);
        ck(gv_vec_upload(ctx, p1, p1h.data()), "gv_vec_upload");
    }
    double tau1 = gam1, tau2 = 0;
    alpha1 = 0;
    const double sqrtN = sqrt((double)N);
    const std::string pre = out_dir + out_name;
    std::vector<double> x1_host(M > 0 ? M : 0, 0.0);
    // covariates (:84-87): effects start at 0, Z = dataset->get_covs(); m_cov[i] = <Z[i], cov_eff> lives in HBM
    gv_vec* m_cov = nullptr;
    std::vector<std::vector<double>> Z;
    if (C > 0) {
        cov_eff = std::vector<double>(C, 0.0);
        Z = dataset->get_covs();
        if ((int)Z.size() < N) {
            std::cout << "FATAL: --C " << C << " but only " << Z.size() << " covariate rows for N = " << N
                      << " (--cov-file)" << std::endl;
            exit(EXIT_FAILURE);
        }
    }

    for (int it = 1; it <= max_iter; it++) {
        const double t_start = now_s();
        double t_io = 0;
        gv_counters c0;
        gv_get_counters(ctx, &c0);
        vamp_iter_stats st;
        memset(&st, 0, sizeof(st));
        if (verbose && rank == 0)
            std::cout << std::endl << "********************" << std::endl << "iteration = " << it << std::endl
                      << "********************" << std::endl;
        if (it == 1 && C > 0) {                                           // :110-131: once, with gg = z1_hat = 0
            const double tc = now_s();
            // y: the filtered phenotype (NA -> 0), i.e. what the z-side denoiser sees (:320).  The reference hands over
            // get_phen() (vamp.cpp:151), in which an NA is DBL_MAX and the fit degenerates; equal when nothing is missing.
            std::vector<double> yh = dataset->filter_pheno();
            yh.resize(N, 0.0);
            cov_eff = Newton_method_cov(yh, std::vector<double>(N, 0.0), Z, cov_eff);
            if (rank == 0 && verbose) {
                for (int i0 = 0; i0 < C; i0++) {
                    std::cout << "cov_eff[" << i0 << "] = " << cov_eff[i0] << ", ";
                    if (i0 % 4 == 3) std::cout << std::endl;
                }
                std::cout << std::endl << "time for covariates effects update = " << now_s() - tc << " seconds." << std::endl;
            }
            std::vector<double> mh = dataset->Zx(cov_eff);
            newN(&m_cov);
            ck(gv_vec_upload(ctx, m_cov, mh.data()), "gv_vec_upload");
        }
        if (verbose && rank == 0) std::cout << "->DENOISING" << std::endl;
        const double rho_it = 1;                                          // :71
        ck(gv_vec_copy(ctx, x1_hat_prev, x1_hat), "gv_vec_copy");
        const double alpha1_prev = alpha1;
        int it_revar = 1;
        for (; it_revar <= 50; it_revar++) {                              // :117-160
            double sums[2];
            ck(gv_denoise_global(ctx, r1, gam1, probs.data(), vars.data(), (int)probs.size(), x1_hat, nullptr, sums), "gv_denoise_global");
            alpha1 = sums[0] / Mt;
            eta1 = gam1 / alpha1;
            if (it <= 1) break;
            const double prev = gam1;
            gam1 = std::min(std::max(1 / (1 / eta1 + sums[1] / Mt), gamma_min), gamma_max);
            if (verbose && rank == 0) std::cout << "it_revar = " << it_revar << ": gam1 = " << gam1 << std::endl;
            updatePrior(0);
            if (std::abs(gam1 - prev) < 1e-3) break;
        }
        st.gam1_denoise = gam1;
        st.revar_rounds = std::max(it_revar - 1, 1);
        if (it > 1) {                                                     // :197-203
            ck(gv_vec_axpby(ctx, x1_hat, rho, x1_hat, 1 - rho, x1_hat_prev), "gv_vec_axpby");
            alpha1 = rho * alpha1 + (1 - rho) * alpha1_prev;
        }
        double t0 = now_s();
        {   // :205-224
            ck(gv_vec_download(ctx, x1_hat, x1_host.data()), "gv_vec_download");
            std::vector<double> stored = x1_host;
            for (double& v : stored) v /= sqrtN;
            if (store_iterates) mpi_store_vec_to_file(pre + "_probit_it_" + std::to_string(it) + ".bin", stored, S, M);
            if (keep_history) x1_hist.push_back(stored);
        }
        store_scaled(pre + "_probit_r1_it_" + std::to_string(it) + ".bin", r1, &r1_hist);
        t_io += now_s() - t0;
        if (verbose && rank == 0) std::cout << "alpha1 = " << alpha1 << std::endl;

        gam_before = gam2;
        gam2 = std::min(std::max(eta1 - gam1, gamma_min), gamma_max);    // :283
        if (verbose && rank == 0) std::cout << "eta1 = " << eta1 << std::endl << "gam2 = " << gam2 << std::endl;
        ck(gv_vec_axpby(ctx, r2, eta1 / gam2, x1_hat, -gam1 / gam2, r1), "gv_vec_axpby");   // :291-292
        st.alpha1 = alpha1; st.eta1 = eta1; st.gam2 = gam2; st.rho = rho;

        double zs[2];                                                     // :335-379 (one round)
        ck(gv_probit_denoise_cov(ctx, p1, y, m_cov, tau1, probit_var, z1_hat, zs), "gv_probit_denoise_cov");
        const double beta1 = zs[0] / N;
        if (verbose && rank == 0) std::cout << "beta1 = " << beta1 << std::endl;
        const double zeta1 = tau1 / beta1;
        if (it > 1) tau1 = std::min(std::max(1 / (1 / zeta1 + zs[1] / N), gamma_min), gamma_max);
        ck(gv_vec_axpby(ctx, p2, 1.0 / (1 - beta1), z1_hat, -beta1 / (1 - beta1), p1), "gv_vec_axpby");   // :447-448
        tau2 = tau1 * (1 - beta1) / beta1;                                // :460
        st.beta1 = beta1;
        if (verbose && rank == 0) std::cout << "tau2 = " << tau2 << std::endl << "probit_var = " << probit_var << std::endl;

        if (verbose && rank == 0) std::cout << std::endl << "->LMMMSE" << std::endl;
        ck(gv_atx_dev(ctx, p2, vM), "gv_atx_dev");                         // :492-495
        ck(gv_vec_axpby(ctx, vM, tau2, vM, gam2, r2), "gv_vec_axpby");
        if (fuse_solves)                                                   // :497 (always from zero) + :512 in lock-step
            alpha2 = fused_solves(vM, nullptr, tau2, dataset, &st.cg_iters, &st.onsager_iters);
        else {
            cg(vM, nullptr, tau2, 1, x2_hat, &st.cg_iters);
            alpha2 = g2d_onsager(gam2, tau2, dataset, &st.onsager_iters);
        }
        store_scaled("", x2_hat, &x2_hist);
        st.alpha2 = alpha2;
        if (verbose && rank == 0) std::cout << "alpha2 = " << alpha2 << std::endl;
        eta2 = gam2 / alpha2;
        if (it > 1) {                                                     // :520-526
            ck(gv_vec_axpby(ctx, tM, 1.0, x2_hat, -1.0, r2), "gv_vec_axpby");
            gam2 = std::min(std::max(1 / (1 / eta2 + dotM(tM, tM) / Mt), gamma_min), gamma_max);
        }
        st.eta2 = eta2; st.gam2_reest = gam2;
        if (verbose && rank == 0) std::cout << "gam2 after reest = " << gam2 << std::endl;
        // r1 = rho_it (x2 - alpha2 r2) / (1 - alpha2) + (1 - rho_it) r1_prev, rho_it = 1 (:533-535)
        ck(gv_vec_axpby(ctx, r1, rho_it / (1 - alpha2), x2_hat, -rho_it * alpha2 / (1 - alpha2), r2), "gv_vec_axpby");
        gam1 = gam2 * (1 - alpha2) / alpha2;                              // :548-549
        st.gam1_next = gam1;
        if (verbose && rank == 0) std::cout << "gam1 = " << gam1 << std::endl;
        if (fuse_solves && have_derived)                                   // z2_hat = A x2_hat (:555): a by-product of the solve
            ck(gv_vec_copy(ctx, tN, ax2_der), "gv_vec_copy");
        else
            ck(gv_ax_dev(ctx, x2_hat, tN), "gv_ax_dev");
        const double beta2 = (double)Mt / N * (1 - alpha2);               // :561
        const double zeta2 = tau2 / beta2;
        if (it > 1) {                                                     // :565-574
            ck(gv_vec_axpby(ctx, tN2, 1.0, tN, -1.0, p2), "gv_vec_axpby");
            tau2 = 1.0 / (1.0 / zeta2 + dotN(tN2, tN2) / N);
        }
        if (verbose && rank == 0) std::cout << "beta2 = " << beta2 << std::endl << "tau2 after reest = " << tau2 << std::endl;
        ck(gv_vec_axpby(ctx, p1, rho_it / (1 - beta2), tN, -rho_it * beta2 / (1 - beta2), p2), "gv_vec_axpby");   // :579-581
        tau1 = tau2 * (1 - beta2) / beta2;                                // :594-595
        st.tau2 = tau2; st.tau1_next = tau1;
        if (verbose && rank == 0) std::cout << "tau1 = " << tau1 << std::endl;

        st.L_after = (int)probs.size();
        gv_counters c1;
        gv_get_counters(ctx, &c1);
        st.n_ax = (long)(c1.n_ax - c0.n_ax);
        st.n_atx = (long)(c1.n_atx - c0.n_atx);
        st.n_ax_pass = (long)(c1.n_ax_pass - c0.n_ax_pass);
        st.n_atx_pass = (long)(c1.n_atx_pass - c0.n_atx_pass);
        ck(gv_vec_axpby(ctx, tM, 1.0, x1_hat_prev, -1.0, x1_hat), "gv_vec_axpby");   // :624-640
        const gv_vec* xs[2] = {tM, x1_hat_prev};
        const gv_vec* ys[2] = {tM, x1_hat_prev};
        double d2[2];
        ck(gv_vec_dots(ctx, 2, xs, ys, 1, d2), "gv_vec_dots");
        st.seconds_io = t_io;
        st.seconds = now_s() - t_start - t_io;
        stats.push_back(st);
        if (it > 1 && sqrt(d2[0] / d2[1]) < stop_criteria_thr) {
            if (verbose && rank == 0)
                std::cout << "probitVAMP stopping criteria fulfilled with threshold = " << stop_criteria_thr << "." << std::endl;
            break;
        }
    }
    for (gv_vec* v : {p1, p2, z1_hat, m_cov})
        if (v) gv_vec_free(ctx, v);
    return x1_host;   // unscaled x1_hat (:657)
}

// ---- covariates of the probit model ----------------------------------------------------------------------------------
// Every rank holds all N individuals and all C covariates, so these run redundantly on the host of every rank, as in the
// reference (no collective); N x C is tiny next to the genotype passes.

// vamp_probit.cpp:813-838 -- gradient of the mean negative log-likelihood of y ~ probit(gg + Z eta)
std::vector<double> vamp::grad_cov(const std::vector<double>& y, const std::vector<double>& gg, double probit_var,
                                   const std::vector<std::vector<double>>& Z, const std::vector<double>& eta) {
    std::vector<double> grad(C, 0.0);
    const double isd = 1.0 / sqrt(probit_var);
    for (int i = 0; i < N; i++) {
        const double s = 2 * y[i] - 1;
        const double arg = s * isd * (gg[i] + inner_prod(Z[i], eta, 0));
        const double ratio = 2.0 / sqrt(2 * M_PI) / erfcx(-arg / sqrt(2.0));          // pdf / cdf at arg
        for (int j = 0; j < C; j++) grad[j] -= ratio * s * isd * Z[i][j];
    }
    for (int j = 0; j < C; j++) grad[j] /= N;
    return grad;
}

// vamp_probit.cpp:840-857
double vamp::mlogL_probit(const std::vector<double>& y, const std::vector<double>& gg, double probit_var,
                          const std::vector<std::vector<double>>& Z, const std::vector<double>& eta) {
    double acc = 0;
    const double isd = 1.0 / sqrt(probit_var);
    for (int i = 0; i < N; i++) acc -= log(normal_cdf((2 * y[i] - 1) * isd * (gg[i] + inner_prod(Z[i], eta, 0))));
    return acc / N;
}

// vamp_probit.cpp:936-1062 -- Newton-Raphson with backtracking (step scaled by 0.9 until the Armijo-type test of
// :1023 holds).  Reference behaviour kept on purpose: the Newton weights use (2y-1) g without 1/sqrt(probit_var)
// (:951-957) while the line search uses mlogL_probit with it; the relative-step exit returns the point BEFORE the
// last step (:1044-1049); a singular Hessian yields a zero step (:991-994); at most 501 outer iterations.
std::vector<double> vamp::Newton_method_cov(const std::vector<double>& y, const std::vector<double>& gg,
                                            const std::vector<std::vector<double>>& Z, std::vector<double> eta) {
    std::vector<double> eta_new = eta;
    for (int it = 0; it <= 500; it++) {
        std::vector<double> hess((size_t)C * C, 0.0), step(C, 0.0);
        for (int i = 0; i < N; i++) {
            const double g_i = gg[i] + inner_prod(Z[i], eta, 0);
            const double s = 2 * y[i] - 1;
            const double lam = s * (2.0 / sqrt(2 * M_PI) / erfcx(-s * g_i / sqrt(2.0)));
            const double w = lam * (lam + g_i);
            for (int j = 0; j < C; j++) {
                step[j] += Z[i][j] * lam;                                              // Xtm * lambda (:976)
                for (int k = 0; k < C; k++) hess[(size_t)j * C + k] += Z[i][j] * (Z[i][k] * w);   // Xtm * WXm (:975)
            }
        }
        if (!lu_solve(hess, step, C)) step.assign(C, 0.0);
        const std::vector<double> grad = grad_cov(y, gg, probit_var, Z, eta);
        double fac = 1;
        const double f0 = mlogL_probit(y, gg, probit_var, Z, eta);
        std::vector<double> displ(C, 0.0);
        for (int i = 1; i < 300; i++) {                                                // 0.9^300 = 1.8e-14
            for (int j = 0; j < C; j++) {
                displ[j] = fac * step[j];
                eta_new[j] = eta[j] + displ[j];
            }
            if (mlogL_probit(y, gg, probit_var, Z, eta_new) <= f0 + inner_prod(displ, grad, 0) / 2) {
                if (rank == 0 && verbose) std::cout << "scale = " << fac << std::endl;
                break;
            }
            fac *= 0.9;
        }
        std::vector<double> diff = eta;
        for (int j = 0; j < C; j++) diff[j] -= eta_new[j];
        const double norm_eta = sqrt(l2_norm2(eta, 0));
        const double rel_err = norm_eta == 0 ? 1 : sqrt(l2_norm2(diff, 0)) / norm_eta;
        if (rank == 0 && verbose) std::cout << "[Newton_cov] it = " << it << ", relative err = " << rel_err << std::endl;
        if (rel_err < 1e-4) {
            if (rank == 0 && verbose)
                std::cout << "[Newton_cov] relative error <= 1e-4 - stoping criteria satisfied" << std::endl;
            break;
        }
        const double before = mlogL_probit(y, gg, probit_var, Z, eta);
        eta = eta_new;
        const double after = mlogL_probit(y, gg, probit_var, Z, eta);
        if (after > before) {
            if (rank == 0 && verbose) {
                std::cout << "previous mlogL = " << before << ", current mlogL = " << after << std::endl;
                std::cout << "likelihood value is not increasing -> terminating Newton-Raphson mehod" << std::endl;
            }
            break;
        }
    }
    return eta;
}
