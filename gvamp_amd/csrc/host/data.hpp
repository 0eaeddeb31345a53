// data.hpp -- host side of the reference's `class data` (data.hpp:18-146) for the linear model: same constructors,
// getters and Ax / ATx signatures; the genotype block lives in HBM behind a gv_ctx and every matvec is a HIP kernel
// (include/gvamp.h).  No CPU path: construction fails (exit, as the reference does on I/O errors) without a GPU.
#pragma once
#include <string>
#include <vector>

#include "gvamp.h"

// the process's one communicator (RCCL, or the shared-memory host transport with GVAMP_COMM=host), created at first use by the
// first data object of a sharded run; gv_host_finalize() is the drivers' MPI_Finalize
gv_ctx* gv_host_world(int device);
void gv_host_finalize();

class data {
private:
    std::string phenfp, bedfp, bimfp, type_data = "bed";
    int N = 0, M = 0, Mt = 0, S = 0, rank = 0;
    int nonas = 0, nas = 0, im4 = 0;
    size_t mbytes = 0;
    std::vector<double> phen_data;
    std::vector<unsigned char> mask4;
    std::vector<double> mave, msig;    // host copies of the device statistics (get_mave / get_msig)
    std::vector<std::vector<double>> covs;
    double intercept = 0, scale = 1, alpha_scale = 1;
    gv_ctx* ctx = nullptr;
    bool owns_ctx = true;

    void open_device(int device, int kernel_mode);
    void push_mask();

public:
    // data.cpp:69-113 -- phenotype given as a vector, every individual present
    data(std::vector<double> y, std::string genofp, const int N, const int M, const int Mt, const int S, const int rank,
         std::string type_data = "bed", double alpha_scale = 1, std::string bimfp = "", int device = -1,
         int kernel_mode = 1);
    // data.cpp:30-61 -- phenotype file (.phen) with NA handling
    data(std::string fp, std::string genofp, const int N, const int M, const int Mt, const int S, const int rank,
         std::string type_data = "bed", double alpha_scale = 1, std::string bimfp = "", int device = -1,
         int kernel_mode = 1);
    // [ext] adopt a context whose genotype shard is already resident (synthetic shards, bench, tests).
    // The context's mask is replaced by the full mask unless mask4 is given.
    data(gv_ctx* resident, std::vector<double> y, const int N, const int M, const int Mt, const int S, const int rank,
         const std::vector<unsigned char>* mask4 = nullptr, int nonas = -1, double alpha_scale = 1);
    ~data();
    data(const data&) = delete;
    data& operator=(const data&) = delete;

    std::vector<double> get_phen() { return phen_data; }
    void set_phen(std::vector<double> new_data) { phen_data = new_data; }
    std::string get_bimfp() { return bimfp; }
    double get_intercept() { return intercept; }
    double get_scale() { return scale; }
    size_t get_mbytes() { return mbytes; }
    double* get_mave() { return mave.data(); }
    double* get_msig() { return msig.data(); }
    std::vector<unsigned char>& get_mask4() { return mask4; }
    int get_im4() const { return im4; }
    int get_nonas() { return nonas; }
    void set_nonas(int num) { nonas = num; }
    int get_S() const { return S; }
    int get_N() const { return N; }
    int get_M() const { return M; }
    int get_Mt() const { return Mt; }
    std::string get_type_data() const { return type_data; }
    gv_ctx* get_ctx() { return ctx; }

    void read_phen();                       // data.cpp:128-192
    void read_genotype_data();              // data.cpp:201-234
    std::vector<int> read_chromosome_info(std::string bim_file);   // data.cpp:346-380
    void compute_markers_statistics();      // data.cpp:392-546
    std::vector<double> Ax(double* __restrict__ phen);    // data.cpp:848 : M doubles -> 4*mbytes, reduced + scaled
    std::vector<double> ATx(double* __restrict__ phen);   // data.cpp:810 : 4*mbytes doubles -> M
    // covariates of the probit model (data.cpp:286-331, :1050-1058): one row of C values per individual
    void read_covariates(std::string covfp, int C = 0);
    std::vector<std::vector<double>> get_covs() { return covs; }
    void set_covs(std::vector<std::vector<double>> z) { covs = std::move(z); }   // [ext] covariates given in memory
    std::vector<double> Zx(std::vector<double> phen);     // 4*mbytes: <covs[i], phen> for i < N, 0 beyond
    std::vector<double> filter_pheno();                   // data.cpp:1065-1079
    std::vector<double> filter_pheno(int* nonnan);        // data.cpp:1081-1097
    // data.cpp:1108-1226 / :1235-1353 -- leave-one-out / leave-one-chromosome-out t-test p-values, one vector per
    // estimator; written to filepath[ie] (LOO) or filepath[ie] + "_pvals_LOCO.bin" (LOCO) at byte offset S*8.
    std::vector<std::vector<double>> pvals_calc(std::vector<std::vector<double>> z1, std::vector<double> y,
                                                std::vector<std::vector<double>> x1_hat, std::vector<std::string> filepath);
    std::vector<std::vector<double>> pvals_calc_LOCO(std::vector<std::vector<double>> z1, std::vector<double> y,
                                                     std::vector<std::vector<double>> x1_hat,
                                                     std::vector<std::string> filepath);
    // the same on device handles (what vamp::infere_linear calls): z1, y N-space; x1_hat M-space
    // pred_prefix (LOCO only, may be empty): also dump the per-chromosome predictors <prefix>_LOCO_chr_<ch>.csv (data.cpp:1276-1281)
    std::vector<double> pvals_calc_dev(gv_vec* z1, gv_vec* y, gv_vec* x1_hat, bool loco, const std::string& pred_prefix = std::string());
};
