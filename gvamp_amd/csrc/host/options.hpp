// options.hpp -- `class Options`: the command line of the reference (options.hpp:5-155, options.cpp:18-429),
// same flag spellings, defaults, getters and fatal-error behaviour; a table-driven parser instead of the
// reference's strcmp chain.  Extensions of this build are marked [ext].
#pragma once
#include <string>
#include <vector>

class Options {
public:
    Options() = default;
    Options(int argc, char** argv) {
        read_command_line_options(argc, argv);
        check_options();
    }
    void read_command_line_options(int argc, char** argv);

    std::string get_bed_file() const { return bed_file; }
    std::string get_bed_file_test() const { return bed_file_test; }
    std::string get_bim_file() const { return bim_file; }
    std::string get_estimate_file() const { return estimate_file; }
    std::string get_cov_estimate_file() const { return cov_estimate_file; }
    std::string get_cov_file() const { return cov_file; }
    std::string get_freeze_index_file() const { return freeze_index_file; }
    std::string get_out_dir() const { return out_dir; }
    std::string get_out_name() const { return out_name; }
    std::string get_model() const { return model; }
    std::string get_run_mode() const { return run_mode; }
    double get_stop_criteria_thr() const { return stop_criteria_thr; }
    double get_EM_err_thr() const { return EM_err_thr; }
    double get_rho() const { return rho; }
    double get_probit_var() const { return probit_var; }
    unsigned int get_EM_max_iter() const { return EM_max_iter; }
    unsigned int get_CG_max_iter() const { return CG_max_iter; }
    unsigned int get_Mt() const { return Mt; }
    unsigned int get_Mt_test() const { return Mt_test; }
    unsigned int get_N() const { return N; }
    unsigned int get_N_test() const { return N_test; }
    unsigned int get_num_mix_comp() const { return num_mix_comp; }
    unsigned int get_use_lmmse_damp() const { return use_lmmse_damp; }
    unsigned int get_use_XXT_denoiser() const { return use_XXT_denoiser; }
    unsigned int get_store_pvals() const { return store_pvals; }
    unsigned int get_CV() const { return CV; }
    unsigned int get_C() const { return C; }
    unsigned int get_seed() const { return seed; }
    unsigned int get_redglob() const { return redglob; }
    unsigned int get_learn_vars() const { return learn_vars; }
    unsigned int get_init_est() const { return init_est; }
    unsigned int get_use_freeze() const { return use_freeze; }
    unsigned int get_iterations() const { return iterations; }
    double get_h2() const { return h2; }
    double get_alpha_scale() const { return alpha_scale; }
    double get_gamw_init() const { return gamw_init; }
    double get_gam1_init() const { return gam1_init; }
    double get_gamma_damp() const { return gamma_damp; }
    std::vector<double> get_vars() const { return vars; }
    std::vector<double> get_probs() const { return probs; }
    std::vector<int> get_test_iter_range() const { return test_iter_range; }
    const std::vector<std::string>& get_phen_files() const { return phen_files; }
    const std::vector<std::string>& get_phen_files_test() const { return phen_files_test; }
    const std::vector<std::string>& get_true_signal_files() const { return true_signal_files; }
    void list_phen_files() const;
    int count_phen_files() const { return (int)phen_files.size(); }
    int count_phen_files_test() const { return (int)phen_files_test.size(); }
    void set_probit_var(double v) { probit_var = v; }

    // [ext] device / kernel selection and synthetic input of this build
    int get_device() const { return device; }                  // --device (default: LOCAL_RANK or 0)
    int get_kernel_mode() const { return kernel_mode; }        // --kernel-mode 0|1 (default 1: i8 MFMA)
    long get_synth_seed() const { return synth_seed; }         // --synth-seed S: on-device synthetic .bed (no --bed-file)
    unsigned int get_synth_miss_ppm() const { return synth_miss_ppm; }   // --synth-miss-ppm (default 5000)
    int get_diagnostics() const { return diagnostics; }        // --diagnostics 1: the 3 print-only Ax of vamp.cpp:646-681
    // --fuse-solves 0: the reference sequence; 1: the LMMSE and Onsager CG solves share their passes (bit-identical);
    // 2: also z1 = A x1_hat rides in a free slot and A x2_hat, A^T A invQ u come out of the CG recurrences (rounding-level)
    int get_fuse_solves() const { return fuse_solves; }
    int get_store_iterates() const { return store_iterates; }  // --store-iterates 0: skip the per-iteration .bin/.csv dumps
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
    void set_C(unsigned int v) { C = v; }
    void set_freeze(const std::string& file) { use_freeze = 1; freeze_index_file = file; }

private:
    std::string bed_file = "", bed_file_test = "", estimate_file = "", freeze_index_file = "", cov_estimate_file = "",
                cov_file = "", run_mode = "", bim_file = "", out_dir = "", out_name = "", model = "linear";
    double stop_criteria_thr = 1e-4;   // options.hpp:107
    double EM_err_thr = 1e-2;
    unsigned int EM_max_iter = 2;
    unsigned int CG_max_iter = 60;
    unsigned int Mt = 0, N = 0, N_test = 0, Mt_test = 0, num_mix_comp = 0;   // uninitialised in the reference
    unsigned int store_pvals = 0, use_lmmse_damp = 0, use_XXT_denoiser = 0, use_freeze = 0, learn_vars = 1, seed = 1;
    double alpha_scale = 1.0;
    unsigned int CV = 0, redglob = 0, C = 0, init_est = 0;
    double probit_var = 1, gamw_init = 0, gam1_init = -1, gamma_damp = 1;
    std::vector<double> vars, probs;
    std::vector<int> test_iter_range = std::vector<int>(2, -1);
    double rho = 0.15, h2 = -1;
    unsigned int iterations = 1;
    std::vector<std::string> phen_files, phen_files_test, true_signal_files;
    int device = -1, kernel_mode = 1, diagnostics = 0, store_iterates = 1, fuse_solves = 1;
    long synth_seed = -1;
    unsigned int synth_miss_ppm = 5000;

    void fail_if_last(char** argv, const int i);
    void check_options();
};
