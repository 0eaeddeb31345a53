// options.cpp -- table-driven parser for the reference's command line (options.cpp:18-429).
#include "options.hpp"

#include <sys/stat.h>

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <map>
#include <sstream>

namespace {
int rank_for_printing() {
    const char* r = getenv("RANK");
    return r ? atoi(r) : 0;
}
[[noreturn]] void fatal(const std::string& msg) {
    std::cout << msg << std::endl;
    exit(EXIT_FAILURE);
}
std::vector<std::string> split_commas(const std::string& s) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string item;
    while (getline(ss, item, ',')) out.push_back(item);
    return out;
}
void existing_files(const std::string& cslist, std::vector<std::string>& dst) {   // options.cpp:73-123
    for (const std::string& fp : split_commas(cslist)) {
        std::ifstream f(fp);
        if (!f.is_open()) fatal("FATAL: file " + fp + " not found");
        dst.push_back(fp);
    }
}
enum Check { ANY, NONNEG, POS };
unsigned int checked_uint(const char* flag, const char* arg, Check chk, const char* what) {
    int v = atoi(arg);
    if ((chk == NONNEG && v < 0) || (chk == POS && v < 1))
        fatal(std::string("FATAL  : option ") + flag + " has to be " + what + "! (" + arg + " was passed)");
    return (unsigned int)v;
}
}  // namespace

void Options::fail_if_last(char** argv, const int i) {
    std::cout << "FATAL  : missing argument for last option \"" << argv[i]
              << "\". Please check your input and relaunch." << std::endl;
    exit(EXIT_FAILURE);
}

void Options::read_command_line_options(int argc, char** argv) {
    using Handler = std::function<void(const char*)>;
    std::map<std::string, Handler> H;
    auto str = [&](const char* f, std::string& dst) { H[f] = [&dst](const char* a) { dst = a; }; };
    auto dbl = [&](const char* f, double& dst) { H[f] = [&dst](const char* a) { dst = atof(a); }; };
    auto uns = [&](const char* f, unsigned int& dst, Check chk, const char* what) {
        std::string flag = f;
        H[f] = [&dst, chk, what, flag](const char* a) { dst = checked_uint(flag.c_str(), a, chk, what); };
    };
    str("--bed-file", bed_file);
    str("--cov-file", cov_file);
    str("--bed-file-test", bed_file_test);
    str("--estimate-file", estimate_file);
    str("--freeze-index-file", freeze_index_file);
    str("--cov-estimate-file", cov_estimate_file);
    str("--run-mode", run_mode);
    str("--out-name", out_name);
    str("--model", model);
    str("--bim-file", bim_file);
    H["--phen-files"] = [&](const char* a) { existing_files(a, phen_files); };
    H["--true-signal-files"] = [&](const char* a) { existing_files(a, true_signal_files); };
    H["--phen-files-test"] = [&](const char* a) { existing_files(a, phen_files_test); };
    H["--vars"] = [&](const char* a) { for (auto& v : split_commas(a)) vars.push_back(atof(v.c_str())); };
    H["--probs"] = [&](const char* a) { for (auto& v : split_commas(a)) probs.push_back(atof(v.c_str())); };
    H["--test-iter-range"] = [&](const char* a) {
        int nit = 0;
        for (auto& v : split_commas(a)) if (nit < 2) test_iter_range[nit++] = atoi(v.c_str());
    };
    H["--out-dir"] = [&](const char* a) {   // created on parse (options.cpp:271-279)
        out_dir = a;
        struct stat st;
        if (stat(a, &st) != 0) mkdir(a, 0777);
    };
    uns("--use-lmmse-damp", use_lmmse_damp, NONNEG, "a non-negative integer");
    uns("--use-freeze", use_freeze, NONNEG, "a non-negative integer");
    uns("--seed", seed, NONNEG, "a non-negative integer");
    uns("--learn-vars", learn_vars, NONNEG, "a non-negative integer");
    uns("--use-XXT-denoiser", use_XXT_denoiser, NONNEG, "a non-negative integer");
    uns("--iterations", iterations, POS, "a strictly positive integer");
    uns("--num-mix-comp", num_mix_comp, POS, "a strictly positive integer");
    uns("--store-pvals", store_pvals, NONNEG, "an integer");
    uns("--red", redglob, NONNEG, "an integer");
    uns("--init-est", init_est, NONNEG, "an integer");
    uns("--EM-max-iter", EM_max_iter, POS, "a strictly positive integer");
    uns("--Mt", Mt, POS, "a strictly positive integer");
    uns("--CV", CV, NONNEG, "a positive integer");
    uns("--C", C, NONNEG, "a non-negative integer");
    uns("--N", N, POS, "a strictly positive integer");
    uns("--N-test", N_test, POS, "a strictly positive integer");
    uns("--Mt-test", Mt_test, POS, "a strictly positive integer");
    uns("--CG-max-iter", CG_max_iter, POS, "a strictly positive integer");
    dbl("--stop-criteria-thr", stop_criteria_thr);
    dbl("--EM-err-thr", EM_err_thr);
    dbl("--alpha-scale", alpha_scale);
    dbl("--rho", rho);
    dbl("--gamma-damp", gamma_damp);
    dbl("--gam1-init", gam1_init);
    dbl("--gamw-init", gamw_init);
    dbl("--probit-var", probit_var);
    dbl("--h2", h2);
    // [ext]
    H["--device"] = [&](const char* a) { device = atoi(a); };
    H["--kernel-mode"] = [&](const char* a) { kernel_mode = atoi(a); };
    H["--synth-seed"] = [&](const char* a) { synth_seed = atol(a); };
    uns("--synth-miss-ppm", synth_miss_ppm, NONNEG, "a non-negative integer");
    H["--diagnostics"] = [&](const char* a) { diagnostics = atoi(a); };
    H["--store-iterates"] = [&](const char* a) { store_iterates = atoi(a); };
    H["--fuse-solves"] = [&](const char* a) { fuse_solves = atoi(a); };
    H["--reanchor-every"] = [&](const char* a) { reanchor_every = atoi(a) < 0 ? 0 : atoi(a); };
    H["--resident-layout"] = [&](const char* a) {      // read by data::open_device (every data object of the run)
        resident_layout = atoi(a);
        setenv("GVAMP_RESIDENT_LAYOUT", a, 1);
    };

    std::stringstream ss;
    ss << "\nardyh command line options:\n";
    for (int i = 1; i < argc; ++i) {
        auto it = H.find(argv[i]);
        if (it == H.end()) {   // options.cpp:421-424
            std::cout << "FATAL: option \"" << argv[i] << "\" unknown\n";
            exit(EXIT_FAILURE);
        }
        if (i == argc - 1) fail_if_last(argv, i);
        const char* arg = argv[++i];
        it->second(arg);
        ss << argv[i - 1] << " " << arg << "\n";
    }
    if (rank_for_printing() == 0) std::cout << ss.str() << std::endl;
    // [ext] what --resident-layout 3 (auto) may assume about the length of the run: every data object of the process tells the
    // library to expect about 12 ATx passes per VAMP iteration (4-40 in the runs measured), gv_set_expected_passes -- below 1 000 passes a
    // second stripe set does not earn its allocation back and auto takes the one tile layout (as it does for a caller that says
    // nothing: bench.py and a bare binding get the same layout as a driver run of fewer than 84 iterations).  $GVAMP_EXPECTED_PASSES overrides.
    if (!getenv("GVAMP_EXPECTED_PASSES")) setenv("GVAMP_EXPECTED_PASSES", std::to_string((long long)iterations * 12).c_str(), 1);
}

void Options::list_phen_files() const {
    for (const auto& p : phen_files) std::cout << " phen file: " << p << std::endl;
}

void Options::check_options() {   // options.cpp:447-493: a bed file is the only hard requirement
    if (get_bed_file() == "" && get_bed_file_test() == "" && synth_seed < 0) {
        std::cout << "FATAL  : no bed file provided! Please use the --bed-file option." << std::endl;
        exit(EXIT_FAILURE);
    }
}
