// main_real.cpp -- real-data driver mirroring the reference's main_real.cpp:13-599, all run modes:
//   infere (:34-128), test (:129-213), both (:214-283), pvals-calc (:284-368), restart (:369-385),
//   predict (:386-437), predict_single (:438-594).
// Every mode is "load data, maybe run vamp::infere, one or more data::Ax, a reduction, a file": the matvecs are HIP
// kernels behind libgvamp, the rest is host code.  One process per GPU (RANK / WORLD_SIZE from the launcher).
#include <cmath>
#include <iostream>

#include "data.hpp"
#include "options.hpp"
#include "utilities.hpp"
#include "vamp.hpp"

namespace {

std::vector<double> load_estimate(const std::string& file, int M, int S) {
    // ".bin" = raw doubles at offset S*8 (mpi_read_vec_from_file), anything else = text, one value per line
    const size_t dot = file.find(".");
    const std::string ext = dot == std::string::npos ? "" : file.substr(dot + 1);
    std::vector<double> x = (ext == "bin") ? mpi_read_vec_from_file(file, M, S) : read_vec_from_file(file, M, S);
    x.resize(M, 0.0);
    return x;
}

// out-of-sample R2 of an estimate (main_real.cpp:163-176, :191-206): x_est * sqrt(N_test), one Ax, residual norm
double test_r2(data& ds, std::vector<double> x_est, int N_test, const std::vector<double>& y_test, double* err2_out,
               double intercept = 0.0, double scale = 1.0) {
    for (double& v : x_est) v *= sqrt((double)N_test);
    std::vector<double> z = ds.Ax(x_est.data());
    double err2 = 0;
    for (int i = 0; i < N_test; i++) {
        const double zi = intercept + scale * z[i];
        err2 += (y_test[i] - zi) * (y_test[i] - zi);
    }
    if (err2_out) *err2_out = err2;
    const double sd = calc_stdev(y_test);
    return 1 - err2 / (sd * sd * y_test.size());
}

double initial_gamw(const Options& opt) { return (opt.get_h2() == -1) ? 2 : 1.0 / (1.0 - opt.get_h2()); }   // :65-69

}  // namespace

int main(int argc, char** argv) {
    const Options opt(argc, argv);
    const int rank = gv_env_rank();
    const std::string mode = opt.get_run_mode();
    const std::string type_data = "bed";
    const double alpha_scale = opt.get_alpha_scale();
    const std::string bimfp = opt.get_bim_file();
    const int dev = opt.get_device(), km = opt.get_kernel_mode();
    auto need_phen = [&](const std::vector<std::string>& files, const char* flag) {
        if (files.empty()) {
            std::cout << "FATAL  : no phen file(s) provided! Please use the " << flag << " option." << std::endl;
            exit(EXIT_FAILURE);
        }
    };

    if (mode == "infere" || mode == "restart") {
        const int Mt = (int)opt.get_Mt(), N = (int)opt.get_N();
        std::vector<double> MS = divide_work(Mt);
        const int M = (int)MS[0], S = (int)MS[1];
        need_phen(opt.get_phen_files(), "--phen-files");
        data dataset(opt.get_phen_files()[0], opt.get_bed_file(), N, M, Mt, S, rank, type_data, alpha_scale, bimfp, dev, km);
        // infere: gam1 = 1e-6, gamw from --h2 (:64-69); restart: both from --gam1-init / --gamw-init (:381-382), r1 is
        // reloaded from --estimate-file inside infere_linear (vamp.cpp:226-233)
        const double gam1 = (mode == "restart") ? opt.get_gam1_init() : 1e-6;
        const double gamw = (mode == "restart") ? opt.get_gamw_init() : initial_gamw(opt);
        vamp emvamp(M, gam1, gamw, std::vector<double>(M, 0.0), rank, opt);
        emvamp.infere(&dataset);
    } else if (mode == "test") {
        const int N_test = (int)opt.get_N_test(), Mt_test = (int)opt.get_Mt_test();
        std::vector<double> MS = divide_work(Mt_test);
        const int M_test = (int)MS[0], S_test = (int)MS[1];
        need_phen(opt.get_phen_files_test(), "--phen-files-test");
        data dataset_test(opt.get_phen_files_test()[0], opt.get_bed_file_test(), N_test, M_test, Mt_test, S_test, rank,
                          type_data, alpha_scale, bimfp, dev, km);
        std::vector<double> y_test = dataset_test.get_phen();
        const std::string est = opt.get_estimate_file();
        const size_t dot = est.find("."), pos_it = est.rfind("it");
        const std::string ext = est.substr(dot + 1);
        const std::vector<int> range = opt.get_test_iter_range();
        if (rank == 0) std::cout << "iter range = [" << range[0] << ", " << range[1] << "]" << std::endl;
        if (range[0] != -1) {                                                          // :152-181
            double maxR2 = -1;
            int maxind = -1;
            for (int it = range[0]; it <= range[1]; it++) {
                const std::string f = est.substr(0, pos_it) + "it_" + std::to_string(it) + "." + ext;
                const double R2 = test_r2(dataset_test, load_estimate(f, M_test, S_test), N_test, y_test, nullptr);
                if (rank == 0) std::cout << R2 << ", ";
                if (R2 > maxR2) {
                    maxR2 = R2;
                    maxind = it;
                }
            }
            if (rank == 0)
                std::cout << std::endl << "max R2 = " << maxR2 << std::endl << std::endl << "max ind = " << maxind << std::endl;
        } else {                                                                       // :183-211
            if (rank == 0) std::cout << "est_file_name = " << est << std::endl;
            double err2 = 0;
            const double R2 = test_r2(dataset_test, load_estimate(est, M_test, S_test), N_test, y_test, &err2);
            const double sd = calc_stdev(y_test);
            if (rank == 0) {
                std::cout << "y stdev^2 = " << sd * sd << std::endl;
                std::cout << "test l2 pred err^2 = " << err2 << std::endl;
                std::cout << "test R2 = " << R2 << std::endl;
            }
        }
    } else if (mode == "both") {                                                       // :214-283
        const int Mt = (int)opt.get_Mt(), N = (int)opt.get_N();
        std::vector<double> MS = divide_work(Mt);
        const int M = (int)MS[0], S = (int)MS[1];
        need_phen(opt.get_phen_files(), "--phen-files");
        need_phen(opt.get_phen_files_test(), "--phen-files-test");
        std::vector<double> x_est;
        double intercept, scale;
        {
            data dataset(opt.get_phen_files()[0], opt.get_bed_file(), N, M, Mt, S, rank, type_data, alpha_scale, bimfp, dev, km);
            vamp emvamp(M, 1e-6, initial_gamw(opt), std::vector<double>(M, 0.0), rank, opt);
            x_est = emvamp.infere(&dataset);
            intercept = dataset.get_intercept();
            scale = dataset.get_scale();
        }
        if (rank == 0) std::cout << "intercept = " << intercept << std::endl << "scale = " << scale << std::endl;
        const int N_test = (int)opt.get_N_test(), Mt_test = (int)opt.get_Mt_test();
        data dataset_test(opt.get_phen_files_test()[0], opt.get_bed_file_test(), N_test, M, Mt_test, S, rank, type_data,
                          alpha_scale, bimfp, dev, km);
        std::vector<double> y_test = dataset_test.get_phen();
        double err2 = 0;
        const double R2 = test_r2(dataset_test, x_est, N_test, y_test, &err2, intercept, scale);   // :262-272
        const double sd = calc_stdev(y_test);
        if (rank == 0) {
            std::cout << std::endl << "y stdev^2 = " << sd * sd << std::endl;
            std::cout << "test l2 pred err^2 = " << err2 << std::endl;
            std::cout << "test R2 = " << R2 << std::endl;
        }
    } else if (mode == "pvals-calc") {                                                 // :284-368
        const int Mt = (int)opt.get_Mt(), N = (int)opt.get_N();
        std::vector<double> MS = divide_work(Mt);
        const int M = (int)MS[0], S = (int)MS[1];
        need_phen(opt.get_phen_files(), "--phen-files");
        data dataset(opt.get_phen_files()[0], opt.get_bed_file(), N, M, Mt, S, rank, type_data, alpha_scale, bimfp, dev, km);
        const std::string est = opt.get_estimate_file();
        const size_t dot = est.rfind("."), pos_it = est.rfind("it");
        const std::string ext = est.substr(dot + 1);
        const std::vector<int> range = opt.get_test_iter_range();
        if (rank == 0) std::cout << "iter range = [" << range[0] << ", " << range[1] << "]" << std::endl;
        std::vector<std::vector<double>> z1_hats, x1_hats;
        std::vector<std::string> out_loo, out_loco;
        const std::string pre = opt.get_out_dir() + opt.get_out_name();
        auto add = [&](const std::string& file, const std::string& tag) {
            std::vector<double> x = load_estimate(file, M, S);
            for (double& v : x) v *= sqrt((double)N);
            z1_hats.push_back(dataset.Ax(x.data()));
            x1_hats.push_back(x);
            out_loo.push_back(pre + tag + "_pvals.bin");
            out_loco.push_back(pre + tag);          // pvals_calc_LOCO appends "_pvals_LOCO.bin" (data.cpp:1347-1350)
        };
        if (range[0] != -1)
            for (int it = range[0]; it <= range[1]; it++)
                add(est.substr(0, pos_it) + "it_" + std::to_string(it) + "." + ext, "_it_" + std::to_string(it));
        else
            add(est, "");
        std::vector<double> y = dataset.filter_pheno();
        const int sp = (int)opt.get_store_pvals();   // 0 = LOO and LOCO, 1 = only LOO, 2 = only LOCO (:313)
        if (sp == 0 || sp == 1) dataset.pvals_calc(z1_hats, y, x1_hats, out_loo);
        if (dataset.get_bimfp() != "" && (sp == 0 || sp == 2)) dataset.pvals_calc_LOCO(z1_hats, y, x1_hats, out_loco);
    } else if (mode == "predict" || mode == "predict_single") {                        // :386-594
        const int N_test = (int)opt.get_N_test(), Mt_test = (int)opt.get_Mt_test();
        std::vector<double> MS = divide_work(Mt_test);
        const int M_test = (int)MS[0], S_test = (int)MS[1];
        data dataset_test(std::vector<double>(N_test, 0.0), opt.get_bed_file_test(), N_test, M_test, Mt_test, S_test, rank,
                          type_data, alpha_scale, bimfp, dev, km);
        const std::string est = opt.get_estimate_file();
        const std::string ext = est.substr(est.find(".") + 1);
        const std::string pre = opt.get_out_dir() + opt.get_out_name();
        auto predict = [&](const std::string& file) {
            std::vector<double> x = load_estimate(file, M_test, S_test);
            for (double& v : x) v *= sqrt((double)N_test);
            return dataset_test.Ax(x.data());
        };
        if (mode == "predict_single") {
            std::vector<double> z = predict(est);
            if (rank == 0) {
                std::cout << "filepath_out = " << pre + "_predict.csv" << std::endl;
                store_vec_to_file(pre + "_predict.csv", z);
            }
        } else {
            if (rank == 0) std::cout << "est_file_name = " << est << std::endl;
            const size_t pos_it = est.rfind("temp");
            const std::vector<int> range = opt.get_test_iter_range();
            if (rank == 0) std::cout << "iter range = [" << range[0] << ", " << range[1] << "]" << std::endl;
            if (range[0] != -1) {
                std::vector<std::vector<double>> zs;
                for (int it = range[0]; it <= range[1]; it++) {
                    const std::string f = est.substr(0, pos_it) + "temp_" + std::to_string(it) + "_" + std::to_string(it) +
                                          "_gibbs_est." + ext;
                    if (rank == 0) std::cout << "est_file_name_it = " << f << std::endl;
                    zs.push_back(predict(f));
                }
                for (int i = 0; i < N_test && rank == 0; i++) {       // one csv per individual (:424-434)
                    std::vector<double> row;
                    for (auto& z : zs) row.push_back(z[i]);
                    store_vec_to_file(pre + "_predict_" + std::to_string(i) + ".csv", row);
                }
            }
        }
    } else {
        std::cout << "FATAL: unknown --run-mode \"" << mode << "\"" << std::endl;
        return EXIT_FAILURE;
    }
    gv_host_finalize();      // the process's communicator, after the last data object (MPI_Finalize of the reference)
    return 0;
}
