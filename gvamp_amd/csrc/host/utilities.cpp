// utilities.cpp -- see utilities.hpp.
#include "utilities.hpp"

#include <fcntl.h>
#include <unistd.h>

#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <numeric>
#include <random>
#include <stdexcept>
#include <string>

static int env_int(const char* k, int dflt) {
    const char* v = getenv(k);
    return v ? atoi(v) : dflt;
}
static bool g_quiet = false;
void gv_host_set_quiet(bool quiet) { g_quiet = quiet; }
bool gv_host_quiet() { return g_quiet; }
int gv_env_rank() { return env_int("RANK", 0); }
int gv_env_nranks() { return env_int("WORLD_SIZE", 1); }
int gv_env_local_rank() { return env_int("LOCAL_RANK", gv_env_rank()); }

void divide_work(int Mt, int nranks, int rank, int* M, int* S, int* Mm) {
    const int base = Mt / nranks, extra = Mt % nranks;   // the first `extra` ranks get one more marker
    int start = 0;
    for (int r = 0; r < rank; r++) start += base + (r < extra ? 1 : 0);
    *M = base + (rank < extra ? 1 : 0);
    *S = start;
    if (Mm) *Mm = base + (extra ? 1 : 0);
}

std::vector<double> divide_work(int Mt) {
    int M, S, Mm;
    const int rank = gv_env_rank(), nranks = gv_env_nranks();
    divide_work(Mt, nranks, rank, &M, &S, &Mm);
    printf("INFO   : rank %4d has %d markers over tot Mt = %d, max Mm = %d, starting at S = %d\n", rank, M, Mt, Mm, S);
    return {(double)M, (double)S, (double)Mm};
}

std::vector<double> simulate(int M, const std::vector<double>& eta, const std::vector<double>& pi, unsigned long seed) {
    std::vector<double> signal(M, 0.0);
    const size_t K = eta.size();
    for (int i = 0; i < M; i++) {
        std::mt19937 gen{seed + i};                      // a fresh engine per marker (utilities.cpp:52)
        std::uniform_real_distribution<double> unif(0.0, 1.0);
        const double u = unif(gen);
        double cum = 0;
        for (size_t j = 0; j < K; j++) {
            cum += pi[j];
            if (u <= cum) {
                if (eta[j] != 0) {
                    std::normal_distribution<double> slab(0.0, sqrt(eta[j]));
                    signal[i] = slab(gen);
                }
                break;
            }
        }
    }
    return signal;
}

void initialize_prior(std::vector<double>& probs, std::vector<double>& vars, int N, int Mt, int rank) {
    if (!probs.empty() || !vars.empty()) return;
    const int L = 23;
    if (Mt <= 50000) throw std::invalid_argument("No probabilities or variances were specified and Mt < 50,000.");
    double p = std::min(50000.0 / Mt, 1.0) / (2 - 1.0 / pow(2, 21));
    probs.push_back(1 - 50000.0 / Mt);
    for (int k = 1; k < L; k++, p /= 2) probs.push_back(p);
    const double ratio = pow(10, log10(1e2 / 1e-5) / (L - 2));
    double v = 1e-5;
    vars.push_back(0);
    for (int k = 1; k < L; k++, v *= ratio) vars.push_back(v);
    for (double& x : vars) x /= N;
    if (rank == 0 && !g_quiet) {
        std::cout << "probs = ";
        for (double x : probs) std::cout << x << ' ';
        std::cout << std::endl << "scaled variances = ";
        for (double x : vars) std::cout << x * N << ' ';
        std::cout << std::endl;
    }
}

double noise_prec_calc(double SNR, const std::vector<double>& vars, const std::vector<double>& probs, int Mt, int N) {
    (void)N;
    double expe = 0;
    for (size_t i = 0; i < vars.size(); i++) expe += vars[i] * probs[i];
    return SNR / Mt / expe;
}

std::vector<double> read_vec_from_file(const std::string& filename, int M, int S) {
    std::vector<double> v;
    std::ifstream in(filename);
    double value;
    for (int it = 0; it < S + M && (in >> value); it++)
        if (it >= S) v.push_back(value);
    return v;
}

void store_vec_to_file(const std::string& filepath, const std::vector<double>& vec) {
    std::ofstream file(filepath);
    for (double x : vec) file << x << std::endl;   // default ostream precision, one value per line
}

void mpi_store_vec_to_file(const std::string& filepath_out, const std::vector<double>& vec, int S, int M) {
    int fd = open(filepath_out.c_str(), O_CREAT | O_WRONLY, 0644);
    if (fd < 0) return;
    ssize_t w = pwrite(fd, vec.data(), sizeof(double) * (size_t)M, (off_t)S * (off_t)sizeof(double));
    (void)w;
    close(fd);
}

std::vector<double> mpi_read_vec_from_file(const std::string& filename, int M, int S) {
    std::vector<double> vec(M, 0.0);
    int fd = open(filename.c_str(), O_RDONLY);
    if (fd < 0) return vec;
    ssize_t r = pread(fd, vec.data(), sizeof(double) * (size_t)M, (off_t)S * (off_t)sizeof(double));
    (void)r;
    close(fd);
    return vec;
}

// Host-vector forms of utilities.cpp:190-214.  Every caller of this build passes sync = 0 (replicated N-vectors, covariate
// rows); the all-reduced products of the reference (sync = 1) run on the device (vamp::dotM -> gv_vec_dots).  A sync = 1
// call has no communicator to use here, so it fails loudly instead of returning a rank-local sum.
static void need_local(int sync, const char* who) {
    if (sync != 0) throw std::invalid_argument(std::string(who) + ": sync = 1 is not available on host vectors (use gv_vec_dots / gv_allreduce_host)");
}
double inner_prod(const std::vector<double>& u, const std::vector<double>& v, int sync) {
    need_local(sync, "inner_prod");
    double acc = 0;
    for (size_t i = 0; i < u.size(); i++) acc += u[i] * v[i];
    return acc;
}
double l2_norm2(const std::vector<double>& u, int sync) { return inner_prod(u, u, sync); }

double calc_stdev(const std::vector<double>& vec, int sync) {
    need_local(sync, "calc_stdev");
    const double sum = std::accumulate(vec.begin(), vec.end(), 0.0);
    const double sq = std::inner_product(vec.begin(), vec.end(), vec.begin(), 0.0);
    const int n = (int)vec.size();
    const double mean = sum / n;
    return std::sqrt((sq - n * mean * mean) / (n - 1));
}

double normal_cdf(double value) { return 0.5 * erfc(-value * M_SQRT1_2); }

// exp(x^2) erfc(x) without overflow / cancellation: the product of the two libm functions while exp(x^2) is finite and
// erfc(x) is a normal number, the asymptotic expansion 1/(x sqrt(pi)) (1 - 1/(2x^2) + 3/(4x^4) - ...) beyond, and
// erfcx(-x) = 2 exp(x^2) - erfcx(x) on the negative axis.
double erfcx(double x) {
    if (std::isnan(x)) return x;
    const double ax = fabs(x);
    double pos;
    if (ax < 25.0)
        pos = exp(ax * ax) * erfc(ax);
    else {
        const double h = 0.5 / (ax * ax);
        double t = 1.0, acc = 1.0;
        for (int k = 1; k <= 12; k++) {
            t *= -(2 * k - 1) * h;
            acc += t;
        }
        pos = acc / (ax * sqrt(M_PI));
    }
    if (x >= 0) return pos;
    const double big = exp(ax * ax);
    return std::isinf(big) ? big : 2.0 * big - pos;
}

bool lu_solve(std::vector<double> A, std::vector<double>& b, int n) {
    auto at = [&](int i, int j) -> double& { return A[(size_t)i * n + j]; };
    for (int col = 0; col < n; col++) {
        int piv = col;
        for (int i = col + 1; i < n; i++)
            if (fabs(at(i, col)) > fabs(at(piv, col))) piv = i;
        if (at(piv, col) == 0.0) return false;
        if (piv != col) {
            for (int j = 0; j < n; j++) std::swap(at(col, j), at(piv, j));
            std::swap(b[col], b[piv]);
        }
        for (int i = col + 1; i < n; i++) {
            const double m = at(i, col) / at(col, col);
            if (m == 0.0) continue;
            for (int j = col + 1; j < n; j++) at(i, j) -= m * at(col, j);
            b[i] -= m * b[col];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double acc = b[i];
        for (int j = i + 1; j < n; j++) acc -= at(i, j) * b[j];
        b[i] = acc / at(i, i);
    }
    return true;
}
