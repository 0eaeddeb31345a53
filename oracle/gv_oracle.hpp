// gv_oracle.hpp -- CPU restatement of the gVAMP linear-model hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under gvamp_amd/ may include, link or call this.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker.
//
// PARITY STATUS: "parity unpinned" in the formal sense of the build rules: every translation
// unit of the reference #includes Boost headers (data.cpp:17, utilities.cpp:8, vamp.cpp:19,
// options.cpp:9) and Boost is absent from this image, so the reference cannot be built without
// stand-in headers, which the rules forbid.  The reference ships no tests or golden vectors
// (SURVEY.md section 4).  What IS pinned:
//   * the 2-bit decode tables: oracle/_ref builds a checker straight from the reference's own
//     generated headers dotp_lut.hpp / na_lut.hpp (self-contained C arrays, no Boost) and
//     compares all 4160 entries with gvo::lut_a/lut_b/na bit for bit (oracle/Makefile: lutcheck);
//   * informational cross-check (not claimed as a pin): outputs of survey-phase probe runs of
//     the reference found in the container (tests/golden/survey_probe/README.md).
//
// Every function cites the reference file:line it restates (paths relative to /root/reference).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace gvo {

// ---- decode (dotp_lut.hpp:3,1030 ; na_lut.hpp:3) -------------------------------------------
// PLINK 2-bit code c (LSB-first within the byte): a(c) = {2,0,1,0}, b(c) = {1,0,1,1}.
static inline double lut_a(unsigned byte, int k) {
    static const double A[4] = {2.0, 0.0, 1.0, 0.0};
    return A[(byte >> (2 * k)) & 3u];
}
static inline double lut_b(unsigned byte, int k) {
    static const double B[4] = {1.0, 0.0, 1.0, 1.0};
    return B[(byte >> (2 * k)) & 3u];
}
static inline double lut_na(unsigned nib, int k) { return double((nib >> k) & 1u); }

// ---- communicator: stands in for MPI_COMM_WORLD (SUM all-reduce of doubles) -------------------
struct Comm {
    int rank = 0, nranks = 1;
    virtual void allreduce_sum(double* buf, size_t n) { (void)buf; (void)n; }  // single rank: identity
    virtual ~Comm() {}
};

// ---- utilities.cpp ---------------------------------------------------------------------------
void divide_work(int Mt, int nranks, int rank, int* M, int* S, int* Mm);                      // :259-291
std::vector<double> simulate(int M, const std::vector<double>& eta,
                             const std::vector<double>& pi, unsigned long seed);             // :48-88
void initialize_prior(std::vector<double>& probs, std::vector<double>& vars, int N, int Mt);  // :91-140
double inner_prod(const std::vector<double>& u, const std::vector<double>& v, int sync, Comm* c);  // :190-210
double l2_norm2(const std::vector<double>& u, int sync, Comm* c);                             // :212-214
double calc_stdev(const std::vector<double>& v);                                              // :235-257 (sync=0)
// utilities.cpp:321-334 -- two-sided t-test of a simple regression slope.  The reference takes the Student-t survival
// function from Boost (absent here); restated with the regularised incomplete beta function (published algorithm:
// P(|T_nu| > t) = I_{nu/(nu+t^2)}(nu/2, 1/2), continued fraction of Numerical Recipes 6.4 / DLMF 8.17.22).
double student_t_two_sided(double t, double nu);
// utilities.cpp:345-409 -- scaled complementary error function exp(x^2) erfc(x).  Restated from its definition
// (libm erfc / exp where they are accurate, the asymptotic series beyond, the reflection 2 exp(x^2) - erfcx(-x) for x < 0)
// instead of the reference's polynomial.
double erfcx(double x);
double normal_cdf(double value);                                                             // utilities.cpp:336-339
// dense solve A x = b (n x n, row-major) by LU with partial pivoting: what Newton_method_cov takes from Boost uBLAS
// (lu_factorize / lu_substitute, vamp_probit.cpp:983-996).  Returns false when a pivot is exactly zero ("singular").
bool lu_solve(std::vector<double> A, std::vector<double>& b, int n);
double linear_reg1d_pvals(double sumx, double sumsqx, double sumxy, double sumy, double sumsqy, int n);

// ---- class data (data.hpp:93-140) --------------------------------------------------------------
struct Data {
    int N = 0, M = 0, Mt = 0, S = 0;
    size_t mbytes = 0;
    int nonas = 0;
    double alpha_scale = 1.0;
    const uint8_t* bed = nullptr;      // M * mbytes, marker-major (data.cpp:201-234 minus the 3 magic bytes)
    std::vector<uint8_t> mask4;        // mbytes nibbles
    std::vector<double> phen;          // N
    std::vector<double> mave, msig;    // M
    Comm* comm = nullptr;
    int nthreads = 1;                  // OpenMP threads for Ax/ATx (cpu_baseline); results do not depend on it
    long n_ax = 0, n_atx = 0;          // call counters (equal work must be shown)

    void init_full_mask();                                        // data.cpp:86-100
    void set_phen_file_semantics(const std::vector<double>& raw,  // data.cpp:128-192 (values; NaN == "NA")
                                 const std::vector<uint8_t>& is_na);
    void compute_markers_statistics();                            // data.cpp:451-484 (scalar path, guarded)
    std::vector<double> Ax(const double* x);                      // data.cpp:951-1007 (scalar path, masked)
    std::vector<double> ATx(const double* p);                     // data.cpp:810-835 + :758-779
    std::vector<double> filter_pheno() const;                     // data.cpp:1065-1079
    // data.cpp:558-716 -- per-individual mean / inverse std / count of the standardised genotypes (bed branch)
    std::vector<double> mave_people, msig_people, numb_people;    // 4*mbytes each
    void compute_people_statistics();
    // data.cpp:1108-1226 (nE = 1): leave-one-out t-test p-value of every local marker
    std::vector<double> pvals_calc(const std::vector<double>& z1, const std::vector<double>& y,
                                   const std::vector<double>& x1_hat);
    // data.cpp:1235-1353 (nE = 1): leave-one-chromosome-out; chrom[m] in 1..23 (read_chromosome_info, :346-380)
    std::vector<double> pvals_calc_LOCO(const std::vector<double>& z1, const std::vector<double>& y,
                                        const std::vector<double>& x1_hat, const std::vector<int>& chrom);
};

// ---- class vamp (vamp.hpp) : linear model only -------------------------------------------------
struct VampOpts {
    int iterations = 1;
    int CG_max_iter = 60, EM_max_iter = 2;
    double EM_err_thr = 1e-2, stop_criteria_thr = 1e-4;
    double rho = 0.15;
    int learn_vars = 1;
    unsigned long seed = 1;
    int use_lmmse_damp = 0;
    double gam1 = 1e-8, gamw = 2.0;
    std::vector<double> probs, vars;      // unscaled (infere multiplies vars by N, vamp.cpp:154)
    std::vector<double> true_signal;      // local shard, may be empty (treated as zeros)
    std::string out_prefix;               // "" = no files; else out_dir+out_name as in vamp.cpp:435
    int verbose = 0;
    int use_XXT_denoiser = 0;             // vamp.cpp:68,169-170,599-606: LMMSE via CG in N-space (denoiserXXT.cpp)
    int bin_class = 0;                    // --model bin_class: vamp::infere_bin_class (vamp_probit.cpp:20-658)
    double probit_var = 1.0;              // options.hpp:124
    int C = 0;                            // --C: number of covariates of the probit model (options.cpp:376)
    std::vector<double> covs;             // N x C row-major: data::covs of read_covariates (data.cpp:286-331)
    std::vector<double> r1_init;          // restart (vamp.cpp:226-233): local shard of the r1 file, non-empty = restart
    std::vector<double> x_init;           // --init-est 1 (vamp.cpp:244-258): local shard of the estimate file
    std::vector<double> freeze_ind;       // --use-freeze 1 (vamp.cpp:205-209,:308,:353): local shard, non-zero = frozen
};

struct IterTrace {
    double gam1_denoise, alpha1, eta1, gam2, alpha2, eta2, gam2_reest, gam1_next, gamw, rho;
    int cg_iters, onsager_iters, revar_rounds, L_after;
    long n_ax, n_atx;
    double seconds;
    double beta1, tau2, tau1_next;   // bin_class only (vamp_probit.cpp:361-372, :460, :579-581)
};

struct Vamp {
    int N, M, Mt;
    VampOpts o;
    Data* d = nullptr;
    Comm* comm = nullptr;
    double gam1, gam2 = 0, gam_before = 0, eta1 = 0, eta2 = 0, alpha1 = 0, alpha2 = 0, gamw, rho;
    std::vector<double> x1_hat, x2_hat, r1, r2, r2_prev, z1, y, mu_CG_last, bern_vec, invQ_bern_vec;
    std::vector<double> probs, vars;
    std::vector<double> gam1s, gam2s, R2trains;
    std::vector<IterTrace> trace;
    std::vector<std::vector<double>> cg_relres;   // per solve: ||r||/||v|| after each step (vamp.cpp:1215)
    std::vector<std::vector<double>> x1_hist, x2_hist, r1_hist;  // stored per iteration, already /sqrt(N)
    int last_cg_iters = 0;

    Vamp(Data* d, const VampOpts& o);
    double g1(double y, double gam1) const;                        // vamp.cpp:805-834
    double g1d(double y, double gam1) const;                       // vamp.cpp:836-869
    void updatePrior(int verbose);                                 // vamp.cpp:929-1072
    std::vector<double> lmmse_mult(const std::vector<double>& v, double tau);              // :1074-1118
    std::vector<double> precondCG_solver(const std::vector<double>& v, const std::vector<double>& mu_start,
                                         double tau, int denoiser);                        // :1130-1229
    double g2d_onsager(double gam2, double tau);                   // :871-889
    double g1_bin_class(double p, double tau1, double y, double m_cov) const;              // vamp_probit.cpp:661-687
    double g1d_bin_class(double p, double tau1, double y, double m_cov) const;             // vamp_probit.cpp:689-705
    std::vector<double> infere_bin_class();                                                // vamp_probit.cpp:20-658
    // covariates of the probit model (--C > 0): effects fitted once, in iteration 1 (vamp_probit.cpp:110-126)
    std::vector<double> cov_eff;                                                           // vamp.hpp:24
    double cov_dot(int i, const std::vector<double>& eta) const;                           // inner_prod(Z[i], eta, 0)
    std::vector<double> grad_cov(const std::vector<double>& y, const std::vector<double>& gg,
                                 const std::vector<double>& eta) const;                    // vamp_probit.cpp:813-838
    double mlogL_probit(const std::vector<double>& y, const std::vector<double>& gg,
                        const std::vector<double>& eta) const;                             // vamp_probit.cpp:840-857
    std::vector<double> Newton_method_cov(const std::vector<double>& y, const std::vector<double>& gg,
                                          std::vector<double> eta) const;                  // vamp_probit.cpp:936-1062
    std::vector<double> lmmse_multAAT(const std::vector<double>& u, double tau);            // denoiserXXT.cpp:15-35
    std::vector<double> CG_solverAAT(const std::vector<double>& v, const std::vector<double>& mu_start, double tau,
                                     int save);                                            // denoiserXXT.cpp:52-130
    std::vector<double> lmmse_denoiserAAT(const std::vector<double>& r2, const std::vector<double>& mu_last);  // :37-50
    void updateNoisePrec();                                        // :892-927
    void err_measures(int ind);                                    // :1232-1318
    std::vector<double> infere();                                  // :149-183 + :190-803 (linear)
};

}  // namespace gvo
