// options.hpp -- `class Options`: the command line of the reference (options.hpp:5-155, options.cpp:18-429),
// same flag spellings, defaults, getter names and fatal-error behaviour.  One table (GV_OPTION_FIELDS) declares every
// scalar option -- type, member, default -- and generates the member and its `get_<member>()`; the parser in options.cpp
// is table-driven too (the reference has a strcmp chain and hand-written getters).  Extensions of this build: [ext].
#pragma once
#include <string>
#include <vector>

// X(type, member, default)
#define GV_OPTION_FIELDS(X)                                                                                          \
    /* files and run selection */                                                                                    \
    X(std::string, bed_file, "") X(std::string, bed_file_test, "") X(std::string, bim_file, "")                     \
    X(std::string, estimate_file, "") X(std::string, cov_estimate_file, "") X(std::string, cov_file, "")           \
    X(std::string, freeze_index_file, "") X(std::string, out_dir, "") X(std::string, out_name, "")                 \
    X(std::string, model, "linear") X(std::string, run_mode, "")                                                    \
    /* problem sizes (uninitialised in the reference) */                                                             \
    X(unsigned int, N, 0) X(unsigned int, Mt, 0) X(unsigned int, N_test, 0) X(unsigned int, Mt_test, 0)              \
    X(unsigned int, num_mix_comp, 0) X(unsigned int, C, 0) X(unsigned int, CV, 0)                                     \
    /* solver (options.hpp:107-...) */                                                                                \
    X(unsigned int, iterations, 1) X(unsigned int, EM_max_iter, 2) X(unsigned int, CG_max_iter, 60)                   \
    X(double, stop_criteria_thr, 1e-4) X(double, EM_err_thr, 1e-2) X(double, rho, 0.15) X(double, h2, -1)            \
    X(double, probit_var, 1) X(double, alpha_scale, 1.0) X(double, gamw_init, 0) X(double, gam1_init, -1)            \
    X(double, gamma_damp, 1)                                                                                          \
    X(unsigned int, learn_vars, 1) X(unsigned int, seed, 1) X(unsigned int, init_est, 0) X(unsigned int, redglob, 0) \
    X(unsigned int, use_lmmse_damp, 0) X(unsigned int, use_XXT_denoiser, 0) X(unsigned int, use_freeze, 0)            \
    X(unsigned int, store_pvals, 0)                                                                                   \
    /* [ext] --device (default: LOCAL_RANK or 0), --kernel-mode 0|1|2 (default 1: i8 MFMA; 2: the same in two-level fixed point), --synth-seed S (on-device   \
       synthetic .bed when there is no --bed-file), --synth-miss-ppm, --diagnostics 1 (the 3 print-only Ax of         \
       vamp.cpp:646-681), --store-iterates 0 (skip the per-iteration .bin/.csv dumps), --fuse-solves (default 4): 0 the reference\
       sequence; 1 the LMMSE and Onsager CG solves share their passes (bit-identical); 2 also z1 = A x1_hat rides in  \
       a free slot and A x2_hat, A^T A invQ u come out of the CG recurrences (rounding-level); 3 also the warm start's \
       initial residual comes from the previous solve's final residual (no pass); 4 also the Onsager solve's first  \
       step comes from A^T A u of the probe, computed once (rounding-level, like 2 and 3); --resident-layout 1|2|3:    \
       kernel mode 1 keeps two stripe sets (2 x M N / 4 bytes) or one tile layout (M N / 4 bytes) in HBM, same bits; \
       3 (default) = two stripe sets if they fit the free HBM, else the tile layout; --reanchor-every K (default 10, 0 =  \
       never): at --fuse-solves 3 / 4 every K-th iteration applies the operator to the warm start explicitly and captures \
       A^T A u afresh, so the products chained from CG residuals never run unanchored for more than K iterations */       \
    X(int, device, -1) X(int, kernel_mode, 1) X(long, synth_seed, -1) X(unsigned int, synth_miss_ppm, 5000)          \
    X(int, diagnostics, 0) X(int, store_iterates, 1) X(int, fuse_solves, 4) X(int, resident_layout, 3)              \
    X(int, reanchor_every, 10)

class Options {
public:
    Options() = default;
    Options(int argc, char** argv) {
        read_command_line_options(argc, argv);
        check_options();
    }
    void read_command_line_options(int argc, char** argv);

#define GV_X(T, n, d) T get_##n() const { return n; }
    GV_OPTION_FIELDS(GV_X)
#undef GV_X
    // list-valued options
    std::vector<double> get_vars() const { return vars; }
    std::vector<double> get_probs() const { return probs; }
    std::vector<int> get_test_iter_range() const { return test_iter_range; }
    const std::vector<std::string>& get_phen_files() const { return phen_files; }
    const std::vector<std::string>& get_phen_files_test() const { return phen_files_test; }
    const std::vector<std::string>& get_true_signal_files() const { return true_signal_files; }
    int count_phen_files() const { return (int)phen_files.size(); }
    int count_phen_files_test() const { return (int)phen_files_test.size(); }
    void list_phen_files() const;

    void set_probit_var(double v) { probit_var = v; }
    // [ext] programmatic construction (host_capi.cpp): the solver knobs that `vamp` reads through the getters above
    void set_solver(unsigned int EM_max_iter_, unsigned int CG_max_iter_, double EM_err_thr_, double stop_criteria_thr_,
                    unsigned int learn_vars_, unsigned int seed_, unsigned int use_lmmse_damp_, int diagnostics_,
                    int store_iterates_) {
        EM_max_iter = EM_max_iter_; CG_max_iter = CG_max_iter_; EM_err_thr = EM_err_thr_;
        stop_criteria_thr = stop_criteria_thr_; learn_vars = learn_vars_; seed = seed_;
        use_lmmse_damp = use_lmmse_damp_; diagnostics = diagnostics_; store_iterates = store_iterates_;
    }
    void set_use_XXT_denoiser(unsigned int v) { use_XXT_denoiser = v; }
    void set_fuse_solves(int v) { fuse_solves = v; }
    void set_reanchor_every(int v) { reanchor_every = v; }
    void set_C(unsigned int v) { C = v; }
    void set_freeze(const std::string& file) { use_freeze = 1; freeze_index_file = file; }

private:
#define GV_X(T, n, d) T n = d;
    GV_OPTION_FIELDS(GV_X)
#undef GV_X
    std::vector<double> vars, probs;
    std::vector<int> test_iter_range = std::vector<int>(2, -1);
    std::vector<std::string> phen_files, phen_files_test, true_signal_files;

    void fail_if_last(char** argv, const int i);
    void check_options();
};
