// main_real_probit.cpp -- probit (case/control) driver, `--run-mode infere` of the reference's main_real_probit.cpp:
// phenotype file, .bed slab, vamp(M, gam1 = 1e-8, gamw = 1, ...) and infere() with --model bin_class.
// Covariates (--cov-file / --C > 0) and the probit test mode are not built yet.
#include <iostream>

#include "data.hpp"
#include "options.hpp"
#include "utilities.hpp"
#include "vamp.hpp"

int main(int argc, char** argv) {
    const Options opt(argc, argv);
    const int rank = gv_env_rank();
    if (opt.get_run_mode() != "infere") {
        std::cout << "FATAL: gvamp_main_real_probit only builds --run-mode infere" << std::endl;
        return EXIT_FAILURE;
    }
    if (opt.get_model() != "bin_class") {
        std::cout << "FATAL: gvamp_main_real_probit needs --model bin_class" << std::endl;
        return EXIT_FAILURE;
    }
    const int Mt = (int)opt.get_Mt(), N = (int)opt.get_N();
    std::vector<double> MS = divide_work(Mt);
    const int M = (int)MS[0], S = (int)MS[1];
    if (opt.get_phen_files().empty()) {
        std::cout << "FATAL  : no phen file(s) provided! Please use the --phen-files option." << std::endl;
        return EXIT_FAILURE;
    }
    data dataset(opt.get_phen_files()[0], opt.get_bed_file(), N, M, Mt, S, rank, "bed", 1.0, "", opt.get_device(),
                 opt.get_kernel_mode());
    const double gam1 = 1e-8;
    vamp emvamp(M, gam1, 1, std::vector<double>(M, 0.0), rank, opt);
    emvamp.infere(&dataset);
    return 0;
}
