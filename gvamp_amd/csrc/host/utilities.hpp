// utilities.hpp -- host helpers with the reference's names and semantics (utilities.hpp:11-51, utilities.cpp).
// MPI is replaced by the process environment (RANK / WORLD_SIZE as set by any one-process-per-GPU launcher) and
// by the communicator inside libgvamp; file outputs keep the reference's formats.
#pragma once
#include <string>
#include <vector>

int gv_env_rank();     // RANK (default 0)
int gv_env_nranks();   // WORLD_SIZE (default 1)
int gv_env_local_rank();
bool gv_host_quiet();
void gv_host_set_quiet(bool quiet);   // silence the informational prints of the helpers below (library use)

// utilities.cpp:259-291 -- {M, S, Mm} of this rank
std::vector<double> divide_work(int Mt);
void divide_work(int Mt, int nranks, int rank, int* M, int* S, int* Mm);
// utilities.cpp:48-88 -- spike-and-slab draw per marker from mt19937{seed + i}
std::vector<double> simulate(int M, const std::vector<double>& eta, const std::vector<double>& pi, unsigned long seed);
// utilities.cpp:91-140 -- default 23-component prior when neither --probs nor --vars is given
void initialize_prior(std::vector<double>& probs, std::vector<double>& vars, int N, int Mt, int rank);
double noise_prec_calc(double SNR, const std::vector<double>& vars, const std::vector<double>& probs, int Mt, int N);
std::vector<double> read_vec_from_file(const std::string& filename, int M, int S);          // utilities.cpp:156-174 (text)
void store_vec_to_file(const std::string& filepath, const std::vector<double>& vec);                // utilities.cpp:178-187 (text)
void mpi_store_vec_to_file(const std::string& filepath_out, const std::vector<double>& vec, int S, int M);   // :293-301 (binary at S*8)
std::vector<double> mpi_read_vec_from_file(const std::string& filename, int M, int S);       // :303-319
double inner_prod(const std::vector<double>& u, const std::vector<double>& v, int sync);   // host vectors; sync must be 0 (throws otherwise)
double l2_norm2(const std::vector<double>& u, int sync);
double calc_stdev(const std::vector<double>& vec, int sync = 0);
double normal_cdf(double value);                                                        // utilities.cpp:336-339
double erfcx(double x);                                                                // utilities.cpp:345-409: exp(x^2) erfc(x)
// A x = b for a dense n x n system (row-major) by LU with partial pivoting -- what the reference takes from Boost uBLAS
// in Newton_method_cov (vamp_probit.cpp:983-996).  false: a pivot is exactly zero, b is left unspecified.
bool lu_solve(std::vector<double> A, std::vector<double>& b, int n);
