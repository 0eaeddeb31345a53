// main_real.cpp -- real-data driver, `--run-mode infere` of the reference's main_real.cpp:34-128: phenotype file,
// .bed slab of this rank, vamp(M, gam1 = 1e-6, gamw, ...), infere().  The post-processing run modes (test, both,
// pvals-calc, restart, predict, predict_single; main_real.cpp:129-594) are SURVEY 8f "next" rows.
#include <iostream>

#include "data.hpp"
#include "options.hpp"
#include "utilities.hpp"
#include "vamp.hpp"

int main(int argc, char** argv) {
    const Options opt(argc, argv);
    const int rank = gv_env_rank();
    if (opt.get_run_mode() != "infere") {
        std::cout << "FATAL: --run-mode " << opt.get_run_mode() << " is not built yet (only \"infere\")" << std::endl;
        return EXIT_FAILURE;
    }
    const size_t Mt = opt.get_Mt();
    std::vector<double> MS = divide_work((int)Mt);
    const int M = (int)MS[0], S = (int)MS[1];
    if (opt.get_phen_files().empty()) {
        std::cout << "FATAL  : no phen file(s) provided! Please use the --phen-files option." << std::endl;
        return EXIT_FAILURE;
    }
    const std::string phenfp = opt.get_phen_files()[0];
    data dataset(phenfp, opt.get_bed_file(), (int)opt.get_N(), M, (int)opt.get_Mt(), S, rank, "bed", opt.get_alpha_scale(),
                 opt.get_bim_file(), opt.get_device(), opt.get_kernel_mode());
    const double gam1 = 1e-6;                                                     // main_real.cpp:64
    const double gamw = (opt.get_h2() == -1) ? 2 : 1.0 / (1.0 - opt.get_h2());    // :65-69
    std::vector<double> beta_true(M, 0.0);
    vamp emvamp(M, gam1, gamw, beta_true, rank, opt);
    std::vector<double> x_est = emvamp.infere(&dataset);
    return 0;
}
