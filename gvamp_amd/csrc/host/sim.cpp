// sim.cpp -- driver of BASELINE config 1, mirroring the reference's sim.cpp:11-273: simulate a spike-and-slab beta,
// y = A (beta sqrt(N)) + noise on the resident genotype shard, then vamp::infere().  One process per GPU; ranks come
// from RANK / WORLD_SIZE (torchrun-style), the N-vector all-reduce is RCCL inside libgvamp.
// [ext] --synth-seed S generates the .bed shard on the device instead of reading --bed-file.
#include <cmath>
#include <iostream>
#include <random>

#include "data.hpp"
#include "options.hpp"
#include "utilities.hpp"
#include "vamp.hpp"

int main(int argc, char** argv) {
    const Options opt(argc, argv);
    const int rank = gv_env_rank(), nranks = gv_env_nranks();
    const size_t Mt = opt.get_Mt(), N = opt.get_N();
    std::vector<double> MS = divide_work((int)Mt);
    const int M = (int)MS[0], S = (int)MS[1];
    const long unsigned int seed = opt.get_seed();

    data* dataset;
    gv_ctx* synth_ctx = nullptr;
    if (opt.get_synth_seed() >= 0) {
        int dev = opt.get_device() >= 0 ? opt.get_device() : gv_env_local_rank();
        if (gv_create(dev, &synth_ctx) || gv_set_dims(synth_ctx, N, M, Mt, S) ||
            gv_set_layout(synth_ctx, opt.get_kernel_mode() == 0,
                          opt.get_kernel_mode() != 0 ? ((opt.get_resident_layout() >= 1 && opt.get_resident_layout() <= 3) ? opt.get_resident_layout() : 3) : 0) ||
            gv_set_kernel_mode(synth_ctx, opt.get_kernel_mode()) ||
            gv_set_expected_passes(synth_ctx, (int64_t)opt.get_iterations() * 12) ||
            gv_synth_bed(synth_ctx, (uint64_t)opt.get_synth_seed(), opt.get_synth_miss_ppm())) {
            std::cout << "FATAL: " << gv_last_error(synth_ctx) << std::endl;
            return EXIT_FAILURE;
        }
        if (nranks > 1) {
            std::cout << "FATAL: --synth-seed with WORLD_SIZE > 1 is driven through bench.py" << std::endl;
            return EXIT_FAILURE;
        }
        dataset = new data(synth_ctx, std::vector<double>(N, 0.0), (int)N, M, (int)Mt, S, rank);
    } else
        dataset = new data(std::vector<double>(N, 0.0), opt.get_bed_file(), (int)N, M, (int)Mt, S, rank, "bed", 1.0,
                           opt.get_bim_file(), opt.get_device(), opt.get_kernel_mode());

    std::vector<double> vars_init = opt.get_vars(), probs_init = opt.get_probs();
    const int CV = opt.get_CV();
    const double h2 = opt.get_h2();
    std::vector<double> vars_true{0, h2 / CV};                                   // sim.cpp:78-79
    std::vector<double> probs_true{1 - (double)CV / Mt, (double)CV / Mt};
    if (rank == 0) {
        std::cout << "true scaled variances = ";
        for (double v : vars_true) std::cout << v * N << ' ';
        std::cout << std::endl << "true probs = ";
        for (double p : probs_true) std::cout << p << ' ';
        std::cout << std::endl;
    }
    const double gamw = 1 / (1 - h2);
    if (rank == 0) std::cout << "true gamw = " << gamw << std::endl;

    std::vector<double> beta_true(M, 0.0), y;
    const std::vector<std::string>& true_signal_files = opt.get_true_signal_files();
    if (!true_signal_files.empty()) {                                             // sim.cpp:134-146
        y = read_vec_from_file(opt.get_phen_files()[0], (int)N, 0);
        dataset->set_phen(y);
        beta_true = mpi_read_vec_from_file(true_signal_files[0], M, S);
    } else {
        // every rank draws the same beta / noise streams (the reference draws on rank 0 and sends them, :151-202)
        std::vector<double> beta_all = simulate((int)Mt, vars_true, probs_true, seed);
        for (int i = 0; i < M; i++) beta_true[i] = beta_all[S + i];
        mpi_store_vec_to_file(opt.get_out_dir() + opt.get_out_name() + "_beta_true.bin", beta_true, S, M);
        std::mt19937 generator{seed};
        std::normal_distribution<double> gauss_beta_gen(0, 1 / sqrt(gamw));
        std::vector<double> noise(N, 0.0);
        for (size_t i = 0; i < N; i++) noise[i] = gauss_beta_gen(generator);
        if (rank == 0) std::cout << "noise prec = " << 1.0 / pow(calc_stdev(noise), 2) << std::endl;
        std::vector<double> beta_true_scaled = beta_true;
        for (double& b : beta_true_scaled) b *= sqrt((double)N);
        y = dataset->Ax(beta_true_scaled.data());
        if (rank == 0) std::cout << "Var(Ax) = " << pow(calc_stdev(y), 2) << std::endl;
        for (size_t i = 0; i < N; i++) y[i] += noise[i];
        y.resize(N);
        dataset->set_phen(y);
        if (rank == 0) {
            store_vec_to_file(opt.get_out_dir() + opt.get_out_name() + "_y.txt", y);
            std::cout << "Var(y) = " << pow(calc_stdev(y), 2) << std::endl;
            const double r = calc_stdev(noise) / calc_stdev(y);
            std::cout << "true R2 = " << 1 - r * r << std::endl;
        }
    }

    const double gamw_init = 2, gam1 = 1e-8;                                     // sim.cpp:242-245
    {   // the vamp object owns device vectors of the dataset's context: it must go first
        vamp emvamp((int)N, M, (int)Mt, gam1, gamw_init, opt.get_iterations(), opt.get_rho(), vars_init, probs_init,
                    beta_true, rank, opt.get_out_dir(), opt.get_out_name(), opt.get_model(), opt);
        std::vector<double> x_est = emvamp.infere(dataset);
    }
    if (rank == 0) {
        std::cout << "var(y) = " << pow(calc_stdev(y), 2) << std::endl;
        std::cout << "true gamw = " << gamw << std::endl;
    }
    delete dataset;
    if (synth_ctx) gv_destroy(synth_ctx);
    gv_host_finalize();      // the process's communicator, after the last data object (MPI_Finalize of the reference)
    return 0;
}
