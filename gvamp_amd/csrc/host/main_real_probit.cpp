// main_real_probit.cpp -- probit (case/control) driver, the run modes of the reference's main_real_probit.cpp:
//   --run-mode infere (:30-78)  : phenotype file, .bed slab, optional covariates, vamp(M, gam1 = 1e-8, gamw = 1) and infere()
//   --run-mode test   (:79-222) : classify a test set with stored estimates, print P / N / TPR / FPR
//   --run-mode both   (:223-316): infere, then classify the test set with the final estimate
// --model bin_class is implied by this driver (the reference's Options default the model to "linear" and this main never
// checks it; passing anything else here is refused rather than silently running the linear model).
#include <cmath>
#include <iostream>

#include "data.hpp"
#include "options.hpp"
#include "utilities.hpp"
#include "vamp.hpp"

namespace {

std::vector<double> load_estimate(const std::string& file, int M, int S) {
    const size_t dot = file.find(".");
    const std::string ext = dot == std::string::npos ? "" : file.substr(dot + 1);
    std::vector<double> x = (ext == "bin") ? mpi_read_vec_from_file(file, M, S) : read_vec_from_file(file, M, S);
    x.resize(M, 0.0);
    return x;
}

// :112-160 / :170-219 / :262-312 -- z = A (x_est sqrt(N_test)) [+ Z cov_effect]; predicted label = [Phi(z) >= 0.5];
// prints "P = .., N = .., TPR = .., FPR = .., " on rank 0.  Individuals with an NA phenotype (read_phen stores DBL_MAX)
// fall into the "N" class exactly as in the reference, which tests y_test[i] == 1 and nothing else.
void classify(data& ds_test, std::vector<double> x_est, int N_test, const std::vector<double>& y_test,
              const std::vector<double>* zx, int rank) {
    for (double& v : x_est) v *= sqrt((double)N_test);
    std::vector<double> z = ds_test.Ax(x_est.data());
    if (zx)
        for (size_t i = 0; i < z.size() && i < zx->size(); i++) z[i] += (*zx)[i];
    int Ne = 0, P = 0, TP = 0, FP = 0;
    for (int i = 0; i < N_test; i++) {
        const int pred = normal_cdf(z[i]) >= 0.5 ? 1 : 0;
        if (y_test[i] == 1) {
            P++;
            TP += pred;
        } else {
            Ne++;
            FP += pred;
        }
    }
    if (rank == 0)
        std::cout << "P = " << P << ", " << "N = " << Ne << ", TPR = " << (double)TP / P << ", FPR = " << (double)FP / Ne << ", ";
}

void need(const std::vector<std::string>& files, const char* flag) {
    if (files.empty()) {
        std::cout << "FATAL  : no phen file(s) provided! Please use the " << flag << " option." << std::endl;
        exit(EXIT_FAILURE);
    }
}

}  // namespace

int main(int argc, char** argv) {
    const Options opt(argc, argv);
    const int rank = gv_env_rank();
    const std::string mode = opt.get_run_mode();
    if (opt.get_model() != "bin_class") {
        std::cout << "FATAL: gvamp_main_real_probit needs --model bin_class" << std::endl;
        return EXIT_FAILURE;
    }
    const int dev = opt.get_device(), km = opt.get_kernel_mode();
    const int C = (int)opt.get_C();
    const bool with_cov = opt.get_cov_file() != "";

    if (mode == "infere") {
        const int Mt = (int)opt.get_Mt(), N = (int)opt.get_N();
        std::vector<double> MS = divide_work(Mt);
        const int M = (int)MS[0], S = (int)MS[1];
        need(opt.get_phen_files(), "--phen-files");
        data dataset(opt.get_phen_files()[0], opt.get_bed_file(), N, M, Mt, S, rank, "bed", 1.0, "", dev, km);
        dataset.read_covariates(opt.get_cov_file(), C);
        vamp emvamp(M, 1e-8, 1, std::vector<double>(M, 0.0), rank, opt);
        emvamp.infere(&dataset);
    } else if (mode == "test") {
        const int N_test = (int)opt.get_N_test(), Mt_test = (int)opt.get_Mt_test();
        std::vector<double> MS = divide_work(Mt_test);
        const int M_test = (int)MS[0], S_test = (int)MS[1];
        need(opt.get_phen_files_test(), "--phen-files-test");
        data dataset_test(opt.get_phen_files_test()[0], opt.get_bed_file_test(), N_test, M_test, Mt_test, S_test, rank, "bed",
                          1.0, "", dev, km);
        const std::vector<double> y_test = dataset_test.get_phen();
        std::vector<double> zx;
        if (with_cov) {                                                                // :117-129
            std::vector<double> cov_effect = read_vec_from_file(opt.get_cov_estimate_file(), C, 0);
            dataset_test.read_covariates(opt.get_cov_file(), C);
            zx = dataset_test.Zx(cov_effect);
        }
        const std::string est = opt.get_estimate_file();
        const size_t dot = est.find("."), pos_it = est.find("it");
        const std::string ext = dot == std::string::npos ? "" : est.substr(dot + 1);
        if (rank == 0) std::cout << "est_file_name = " << est << std::endl;
        const std::vector<int> range = opt.get_test_iter_range();
        if (rank == 0) std::cout << "iter range = [" << range[0] << ", " << range[1] << "]" << std::endl;
        if (range[0] != -1) {
            for (int it = range[0]; it <= range[1]; it++) {
                const std::string f = est.substr(0, pos_it) + "it_" + std::to_string(it) + "." + ext;
                classify(dataset_test, load_estimate(f, M_test, S_test), N_test, y_test, with_cov ? &zx : nullptr, rank);
            }
        } else
            classify(dataset_test, load_estimate(est, M_test, S_test), N_test, y_test, with_cov ? &zx : nullptr, rank);
        if (rank == 0) std::cout << std::endl;
    } else if (mode == "both") {
        const int Mt = (int)opt.get_Mt(), N = (int)opt.get_N();
        std::vector<double> MS = divide_work(Mt);
        const int M = (int)MS[0], S = (int)MS[1];
        need(opt.get_phen_files(), "--phen-files");
        need(opt.get_phen_files_test(), "--phen-files-test");
        std::vector<double> x_est, zx;
        {
            data dataset(opt.get_phen_files()[0], opt.get_bed_file(), N, M, Mt, S, rank, "bed", 1.0, "", dev, km);
            dataset.read_covariates(opt.get_cov_file(), C);
            vamp emvamp(M, 1e-8, 1, std::vector<double>(M, 0.0), rank, opt);
            x_est = emvamp.infere(&dataset);
            // :270-280: the covariate term of the TEST prediction is formed from the TRAINING covariates (the reference
            // has no test covariates option); it is only meaningful when both sets list the same individuals
            if (with_cov) zx = dataset.Zx(emvamp.get_cov_eff());
        }   // the training shard leaves HBM before the test shard is read
        const int N_test = (int)opt.get_N_test(), Mt_test = (int)opt.get_Mt_test();
        // :255-259: the test slab uses the TRAINING marker range (divide_work(Mt), offset S)
        data dataset_test(opt.get_phen_files_test()[0], opt.get_bed_file_test(), N_test, M, Mt_test, S, rank, "bed", 1.0, "",
                          dev, km);
        classify(dataset_test, x_est, N_test, dataset_test.get_phen(), with_cov ? &zx : nullptr, rank);
        if (rank == 0) std::cout << std::endl;
    } else {
        std::cout << "FATAL: --run-mode " << mode << " is not a mode of main_real_probit (infere | test | both)" << std::endl;
        return EXIT_FAILURE;
    }
    gv_host_finalize();      // the process's communicator, after the last data object (MPI_Finalize of the reference)
    return 0;
}
