// gv_oracle.cpp -- CPU restatement of the gVAMP linear-model hot path (see gv_oracle.hpp header:
// TEST INFRASTRUCTURE ONLY; parity status stated there).  Plain C++17 + libstdc++ <random>
// (the reference's RNG streams are libstdc++'s, so the same library reproduces them).
#include "gv_oracle.hpp"

#include <algorithm>
#include <cassert>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <numeric>
#include <random>
#include <stdexcept>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace gvo {

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ================================ utilities.cpp ===============================================

// utilities.cpp:259-291 -- contiguous marker blocks, remainder to the low ranks.
void divide_work(int Mt, int nranks, int rank, int* M, int* S, int* Mm) {
    const int modu = Mt % nranks;
    const int size = Mt / nranks;
    int cum = 0, m = 0, s = 0;
    for (int i = 0; i < nranks; i++) {
        int len = i < modu ? size + 1 : size;
        if (i == rank) { m = len; s = cum; }
        cum += len;
    }
    assert(cum == Mt);
    *M = m; *S = s;
    if (Mm) *Mm = modu != 0 ? size + 1 : size;
}

// utilities.cpp:48-73 -- one draw: fresh mt19937{seed}, one uniform, then (if slab) one normal.
static double generate_mixture_gaussians(int K_grp, const std::vector<double>& eta,
                                         const std::vector<double>& pi, unsigned long seed) {
    std::mt19937 generator{seed};
    std::uniform_real_distribution<double> unif(0.0, 1.0);
    double u = unif(generator);
    double c_sum = 0, out_val = 0;
    for (int j = 0; j < K_grp; j++) {
        c_sum += pi[j];
        if (u <= c_sum) {
            if (eta[j] != 0) {
                std::normal_distribution<double> gauss_beta_gen(0.0, sqrt(eta[j]));
                out_val = gauss_beta_gen(generator);
            } else
                out_val = 0;
            break;
        }
    }
    return out_val;
}

// utilities.cpp:77-88
std::vector<double> simulate(int M, const std::vector<double>& eta, const std::vector<double>& pi,
                             unsigned long seed) {
    int K_grp = (int)eta.size();
    std::vector<double> signal(M, 0.0);
    for (int i = 0; i < M; i++) signal[i] = generate_mixture_gaussians(K_grp, eta, pi, seed + i);
    return signal;
}

// utilities.cpp:91-140 -- default 23-component prior (only when neither probs nor vars given).
void initialize_prior(std::vector<double>& probs, std::vector<double>& vars, int N, int Mt) {
    if (probs.size() != 0 || vars.size() != 0) return;
    int num_mix = 23;
    double probs_1 = std::min(50000.0 / Mt, 1.0) / (2 - 1.0 / pow(2, 21));
    if (Mt <= 50000) throw std::invalid_argument("No probabilities or variances were specified and Mt < 50,000.");
    double curr_prob = probs_1;
    probs.push_back(1 - 50000.0 / Mt);
    for (int i0 = 0; i0 < num_mix - 1; i0++) {
        probs.push_back(curr_prob);
        curr_prob /= 2;
    }
    double start_var = 1e-5, stop_var = 1e2, temp_var = start_var;
    double c_var = pow(10, log10(stop_var / start_var) / (num_mix - 1 - 1));
    vars.push_back(0);
    for (int i0 = 0; i0 < num_mix - 1; i0++) {
        vars.push_back(temp_var);
        temp_var *= c_var;
    }
    for (size_t i0 = 0; i0 < vars.size(); i0++) vars[i0] /= N;
}

// utilities.cpp:190-210 -- sequential sum (the reference's OpenMP reduction is order-free anyway).
double inner_prod(const std::vector<double>& u, const std::vector<double>& v, int sync, Comm* c) {
    double accum = 0;
    for (size_t i = 0; i < u.size(); i++) accum += u[i] * v[i];
    if (sync == 1 && c) c->allreduce_sum(&accum, 1);
    return accum;
}
double l2_norm2(const std::vector<double>& u, int sync, Comm* c) { return inner_prod(u, u, sync, c); }

// utilities.cpp:235-257 with sync = 0
double calc_stdev(const std::vector<double>& vec) {
    double sum = std::accumulate(vec.begin(), vec.end(), 0.0);
    double sq_sum = std::inner_product(vec.begin(), vec.end(), vec.begin(), 0.0);
    int vec_len = (int)vec.size();
    double mean = sum / vec_len;
    return std::sqrt((sq_sum - vec_len * mean * mean) / (vec_len - 1));
}

// Student-t two-sided tail through the regularised incomplete beta function.
static double betacf(double a, double b, double x) {
    const double tiny = 1e-300, eps = 1e-16;
    double qab = a + b, qap = a + 1, qam = a - 1, c = 1, d = 1 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1 / d;
    double h = d;
    for (int m = 1; m <= 100000; m++) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1 / d;
        h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1) < eps) break;
    }
    return h;
}
double student_t_two_sided(double t, double nu) {
    if (std::isnan(t) || !(nu > 0)) return NAN;
    if (t == 0) return 1.0;
    if (std::isinf(t)) return 0.0;
    // x = nu / (nu + t^2) with its logarithms formed from w = t^2 / nu (a * log(x) would lose a * eps otherwise)
    const double a = 0.5 * nu, b = 0.5, w = t * t / nu, x = 1 / (1 + w);
    const double lx = -log1p(w), l1mx = log(w) - log1p(w);
    double lg_ratio;
    if (a >= 30.0) {
        const double ia = 1 / a, ia2 = ia * ia;
        lg_ratio = 0.5 * log(a) - ia * (1.0 / 8 - ia2 * (1.0 / 192 - ia2 * (1.0 / 640 - ia2 * (17.0 / 14336)))) - lgamma(0.5);
    } else
        lg_ratio = lgamma(a + b) - lgamma(a) - lgamma(b);
    const double bt = exp(lg_ratio + a * lx + b * l1mx);
    if (x < (a + 1) / (a + b + 2)) return bt * betacf(a, b, x) / a;
    return 1 - bt * betacf(b, a, w / (1 + w)) / b;
}
// utilities.cpp:321-334
double linear_reg1d_pvals(double sumx, double sumsqx, double sumxy, double sumy, double sumsqy, int n) {
    double s2y = (sumsqy - sumy * sumy / n) / (n - 1);
    double s2x = (sumsqx - sumx * sumx / n) / (n - 1);
    double sxy = (sumxy - sumx * sumy / n) / (n - 1);
    double rxy = sxy / sqrt(s2x * s2y);
    double t = rxy * sqrt((n - 2) / (1 - rxy * rxy));
    return student_t_two_sided(t > 0 ? t : (0 - t), n - 2);   // = 2 * cdf(complement(students_t(n-2), |t|))
}

// ================================ data.cpp ======================================================

// data.cpp:86-100 -- vector-phenotype ctor: every individual present, pad bits of the last nibble cleared.
void Data::init_full_mask() {
    mbytes = (N % 4) ? (size_t)N / 4 + 1 : (size_t)N / 4;
    mask4.assign(mbytes, 0x0F);
    int m4 = N % 4;
    if (m4 != 0)
        for (int i = m4; i < 4; i++) mask4[N / 4] &= ~(0b1 << i);
    nonas = N;
}

// data.cpp:128-192 (read_phen) after tokenising: NA -> DBL_MAX + mask bit cleared; values scaled by
// sqrt((nonas-1)/sum((y-avg)^2)) but NOT centred (:172-182); DBL_MAX entries are scaled too (-> inf).
void Data::set_phen_file_semantics(const std::vector<double>& raw, const std::vector<uint8_t>& is_na) {
    assert((int)raw.size() == N);
    mbytes = (N % 4) ? (size_t)N / 4 + 1 : (size_t)N / 4;
    mask4.clear();
    phen.clear();
    double sum = 0.0;
    int nas = 0;
    nonas = 0;
    for (int line_n = 0; line_n < N; line_n++) {
        int m4 = line_n % 4;
        if (m4 == 0) mask4.push_back(0x0F);
        if (is_na[line_n]) {
            nas += 1;
            phen.push_back(std::numeric_limits<double>::max());
            mask4[line_n / 4] &= ~(0b1 << m4);
        } else {
            nonas += 1;
            phen.push_back(raw[line_n]);
            sum += raw[line_n];
        }
    }
    const int m4 = N % 4;
    if (m4 != 0)
        for (int i = m4; i < 4; i++) mask4[N / 4] &= ~(0b1 << i);
    double avg = sum / double(nonas);
    double sqn = 0.0;
    for (size_t i = 0; i < phen.size(); i++)
        if (phen[i] != std::numeric_limits<double>::max()) sqn += (phen[i] - avg) * (phen[i] - avg);
    sqn = sqrt(double(nonas - 1) / sqn);
    for (size_t i = 0; i < phen.size(); i++) phen[i] *= sqn;
}

// data.cpp:451-484 -- scalar (non-MANVECT) path: masked sums, sumb==0 and sumsqr==0 guards, alpha_scale.
void Data::compute_markers_statistics() {
    mave.assign(M, 0.0);
    msig.assign(M, 0.0);
    const int im4 = (int)mbytes;
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int i = 0; i < M; i++) {
        const uint8_t* bedm = &bed[size_t(i) * mbytes];
        double suma = 0.0, sumb = 0.0;
        for (int j = 0; j < im4; j++)
            for (int k = 0; k < 4; k++) {
                suma += lut_a(bedm[j], k) * lut_na(mask4[j], k);
                sumb += lut_b(bedm[j], k) * lut_na(mask4[j], k);
            }
        if (sumb != 0) mave[i] = suma / sumb;
        else mave[i] = 0.0;
        double sumsqr = 0.0;
        for (int j = 0; j < im4; j++)
            for (int k = 0; k < 4; k++) {
                double val = (lut_a(bedm[j], k) - mave[i]) * lut_b(bedm[j], k) * lut_na(mask4[j], k);
                sumsqr += val * val;
            }
        if (sumsqr != 0) {
            if (alpha_scale == 1.0) msig[i] = 1.0 / sqrt(sumsqr / (double(nonas) - 1.0));
            else msig[i] = 1.0 / pow(sqrt(sumsqr / (double(nonas) - 1.0)), alpha_scale);
        } else
            msig[i] = 1.0;
    }
}

// data.cpp:951-1007 -- scalar path: out[4j+k] += (a - ave) * (msig*x) * b * na, markers in order,
// MPI_Allreduce(SUM) over ranks (:995), then * 1/sqrt(N) (:998-1005).  Threads split the byte columns
// j, so every out[n] sees the markers in the same order as the serial loop: result independent of nthreads.
std::vector<double> Data::Ax(const double* x) {
    n_ax++;
    std::vector<double> out(4 * mbytes, 0.0);
#ifdef _OPENMP
#pragma omp parallel num_threads(nthreads)
#endif
    {
        int tid = 0, nt = 1;
#ifdef _OPENMP
        tid = omp_get_thread_num();
        nt = omp_get_num_threads();
#endif
        // column blocks of 2048 bytes (64 KiB of doubles) so the accumulators stay in L1/L2
        const size_t BLK = 2048;
        size_t nblk = (mbytes + BLK - 1) / BLK;
        for (size_t b = tid; b < nblk; b += nt) {
            size_t j0 = b * BLK, j1 = std::min(mbytes, j0 + BLK);
            for (int i = 0; i < M; i++) {
                const uint8_t* bedm = &bed[size_t(i) * mbytes];
                double ave = mave[i];
                double sig_phen_i = msig[i] * x[i];
                double val[4];
                for (int c = 0; c < 4; c++) val[c] = (lut_a(c, 0) - ave) * sig_phen_i * lut_b(c, 0);
                for (size_t j = j0; j < j1; j++) {
                    unsigned byte = bedm[j], nib = mask4[j];
                    for (int k = 0; k < 4; k++) out[4 * j + k] += val[(byte >> (2 * k)) & 3u] * lut_na(nib, k);
                }
            }
        }
    }
    if (comm) comm->allreduce_sum(out.data(), out.size());
    double scale = 1.0 / sqrt(N);
    for (size_t i = 0; i < out.size(); i++) out[i] *= scale;
    return out;
}

// data.cpp:810-835 (ATx) over :758-779 (scalar dot_product): sigma * (sum a*p - mu * sum b*p), * 1/sqrt(N).
// No mask: relies on p == 0 at NA / pad positions, as the reference does.
std::vector<double> Data::ATx(const double* p) {
    n_atx++;
    std::vector<double> out(M, 0.0);
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int mloc = 0; mloc < M; mloc++) {
        const uint8_t* bedm = &bed[size_t(mloc) * mbytes];
        double dpa = 0.0, dpb = 0.0;
        for (size_t i = 0; i < mbytes; i++) {
            unsigned byte = bedm[i];
            for (int j = 0; j < 4; j++) {
                dpa += lut_a(byte, j) * p[i * 4 + j];
                dpb += lut_b(byte, j) * p[i * 4 + j];
            }
        }
        out[mloc] = msig[mloc] * (dpa - mave[mloc] * dpb);
    }
    double scale = 1.0 / sqrt(N);
    for (int mloc = 0; mloc < M; mloc++) out[mloc] *= scale;
    return out;
}

// data.cpp:1065-1079 -- length N (the reference hands y.data() to ATx, which reads 4*mbytes doubles:
// out of bounds when N%4 != 0).  The restatement pads with zeros to 4*mbytes instead.
std::vector<double> Data::filter_pheno() const {
    std::vector<double> y(4 * mbytes, 0.0);
    for (int i = 0; i < N; i++) y[i] = phen[i];
    for (size_t j = 0; j < mbytes; j++)
        for (int k = 0; k < 4; k++)
            if ((int)(4 * j + k) < N)
                if (lut_na(mask4[j], k) == 0) y[4 * j + k] = 0;
    return y;
}

// data.cpp:558-716 (bed branch): sums over the LOCAL markers, all-reduced over ranks (:604-606), then
// mean = S1 / cnt, msig = sqrt((cnt - 1) / (S2 - cnt * mean^2)); individuals without a phenotype get 0 / 0.
void Data::compute_people_statistics() {
    std::vector<double> s1(4 * mbytes, 0.0), s2(4 * mbytes, 0.0), cnt(4 * mbytes, 0.0);
    for (int i = 0; i < M; i++) {
        const uint8_t* bedm = &bed[size_t(i) * mbytes];
        for (size_t j = 0; j < mbytes; j++)
            for (int k = 0; k < 4; k++) {
                double bm = lut_b(bedm[j], k) * lut_na(mask4[j], k);
                double value = (lut_a(bedm[j], k) - mave[i]) * msig[i] * bm;
                s1[4 * j + k] += value;
                cnt[4 * j + k] += bm;
                s2[4 * j + k] += value * value;
            }
    }
    if (comm) {
        comm->allreduce_sum(s1.data(), s1.size());
        comm->allreduce_sum(cnt.data(), cnt.size());
        comm->allreduce_sum(s2.data(), s2.size());
    }
    for (size_t j = 0; j < mbytes; j++)
        for (int k = 0; k < 4; k++) {
            size_t n = 4 * j + k;
            if ((int)n >= N) continue;
            if (lut_na(mask4[j], k) == 1) {
                s1[n] /= cnt[n];
                s2[n] = (cnt[n] - 1) / (s2[n] - cnt[n] * s1[n] * s1[n]);
            } else {
                s1[n] = 0;
                s2[n] = 0;
            }
        }
    mave_people = s1;
    msig_people = s2;
    numb_people = cnt;
    for (int i = 0; i < N; i++) msig_people[i] = sqrt(msig_people[i]);
}

// data.cpp:1108-1226, one estimator.  y_mod = y - z1; for marker k the marker's own contribution is added back
// (y_mark = y_mod + A_k x_k) and y_mark is regressed on the standardised column.
std::vector<double> Data::pvals_calc(const std::vector<double>& z1, const std::vector<double>& y,
                                     const std::vector<double>& x1_hat) {
    std::vector<double> pvals(M, 0.0), y_mod(4 * mbytes, 0.0);
    for (int i = 0; i < N; i++) y_mod[i] = y[i] - z1[i];
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
    for (int k = 0; k < M; k++) {
        const uint8_t* bedk = &bed[size_t(k) * mbytes];
        double sumx = 0, sumsqx = 0, sumxy = 0, sumy = 0, sumsqy = 0;
        int count = 0;
        for (size_t i = 0; i < mbytes; i++)
            for (int j = 0; j < 4; j++) {
                double bm = lut_b(bedk[i], j) * lut_na(mask4[i], j);
                double value = (lut_a(bedk[i], j) - mave[k]) * msig[k] * bm;
                double gen_part = value / sqrt(N);
                double ym = y_mod[4 * i + j] + gen_part * x1_hat[k];
                sumx += value;
                sumsqx += value * value;
                sumxy += value * ym;
                sumy += ym * bm;
                sumsqy += ym * ym * bm;
                count += (int)bm;
            }
        pvals[k] = linear_reg1d_pvals(sumx, sumsqx, sumxy, sumy, sumsqy, count);
    }
    return pvals;
}

// data.cpp:1235-1353, one estimator.  For chromosome ch the whole chromosome's predictor (all ranks, :1268-1272) is
// added back to y_mod, then its markers are tested against that.
std::vector<double> Data::pvals_calc_LOCO(const std::vector<double>& z1, const std::vector<double>& y,
                                          const std::vector<double>& x1_hat, const std::vector<int>& chrom) {
    std::vector<double> pvals(M, 0.0), y_mod(4 * mbytes, 0.0);
    for (int i = 0; i < N; i++) y_mod[i] = y[i] - z1[i];
    for (int ch = 1; ch <= 23; ch++) {
        std::vector<double> y_chrom(4 * mbytes, 0.0);
        for (int m = 0; m < M; m++) {
            if (chrom[m] != ch) continue;
            const uint8_t* bedm = &bed[size_t(m) * mbytes];
            for (size_t i = 0; i < mbytes; i++)
                for (int j = 0; j < 4; j++) {
                    double gen_part = (lut_a(bedm[i], j) - mave[m]) * msig[m] * lut_b(bedm[i], j) * lut_na(mask4[i], j) / sqrt(N);
                    y_chrom[4 * i + j] += gen_part * x1_hat[m];
                }
        }
        if (comm) comm->allreduce_sum(y_chrom.data(), y_chrom.size());
        for (size_t i = 0; i < y_chrom.size(); i++) y_chrom[i] += y_mod[i];
#ifdef _OPENMP
#pragma omp parallel for num_threads(nthreads) schedule(static)
#endif
        for (int m = 0; m < M; m++) {
            if (chrom[m] != ch) continue;
            const uint8_t* bedm = &bed[size_t(m) * mbytes];
            double sumx = 0, sumsqx = 0, sumxy = 0, sumy = 0, sumsqy = 0;
            int count = 0;
            for (size_t i = 0; i < mbytes; i++)
                for (int j = 0; j < 4; j++) {
                    double bm = lut_b(bedm[i], j) * lut_na(mask4[i], j);
                    double value = (lut_a(bedm[i], j) - mave[m]) * msig[m] * bm;
                    double yc = y_chrom[4 * i + j];
                    sumx += value;
                    sumsqx += value * value;
                    sumxy += value * yc;
                    sumy += yc * bm;
                    sumsqy += yc * yc * bm;
                    count += (int)bm;
                }
            pvals[m] = linear_reg1d_pvals(sumx, sumsqx, sumxy, sumy, sumsqy, count);
        }
    }
    return pvals;
}

// ================================ vamp.cpp ======================================================

static const double gamma_min = 1e-11, gamma_max = 1e11;   // vamp.hpp:31-32
static void store_bin(const std::string& path, const std::vector<double>& v, int S);
static const int auto_var_max_iter = 5;                    // vamp.hpp:37

// vamp.cpp:32-82 (ctor 1) / :89-139 (ctor 2): state initialisation; initialize_prior at :78/:136.
Vamp::Vamp(Data* d_, const VampOpts& o_) : N(d_->N), M(d_->M), Mt(d_->Mt), o(o_), d(d_), comm(d_->comm) {
    gam1 = o.gam1;
    gamw = o.gamw;
    rho = o.rho;
    probs = o.probs;
    vars = o.vars;
    x1_hat.assign(M, 0.0);
    x2_hat.assign(M, 0.0);
    r1.assign(M, 0.0);
    r2.assign(M, 0.0);
    if (o.true_signal.empty()) o.true_signal.assign(M, 0.0);
    initialize_prior(probs, vars, N, Mt);
}

// vamp.cpp:805-834
double Vamp::g1(double y, double gam1) const {
    double sigma = 1 / gam1;
    double eta_max = *(std::max_element(vars.begin(), vars.end()));
    double pk = 0, pkd = 0, val;
    if (sigma < 1e-10 && sigma > -1e-10) return y;
    for (size_t i = 0; i < probs.size(); i++) {
        double expe_sum = -0.5 * pow(y, 2) * (eta_max - vars[i]) / (vars[i] + sigma) / (eta_max + sigma);
        double z = probs[i] / sqrt(vars[i] + sigma) * exp(expe_sum);
        pk = pk + z;
        z = z / (vars[i] + sigma) * y;
        pkd = pkd - z;
    }
    val = (y + sigma * pkd / pk);
    return val;
}

// vamp.cpp:836-869
double Vamp::g1d(double y, double gam1) const {
    double sigma = 1 / gam1;
    double eta_max = *std::max_element(vars.begin(), vars.end());
    double pk = 0, pkd = 0, pkdd = 0;
    if (sigma < 1e-10 && sigma > -1e-10) return 1;
    for (size_t i = 0; i < probs.size(); i++) {
        double expe_sum = -0.5 * pow(y, 2) * (eta_max - vars[i]) / (vars[i] + sigma) / (eta_max + sigma);
        double z = probs[i] / sqrt(vars[i] + sigma) * exp(expe_sum);
        pk = pk + z;
        z = z / (vars[i] + sigma) * y;
        pkd = pkd - z;
        double z2 = z / (vars[i] + sigma) * y;
        pkdd = pkdd - probs[i] / pow(vars[i] + sigma, 1.5) * exp(expe_sum) + z2;
    }
    return (1 + sigma * (pkdd / pk - pow(pkd / pk, 2)));
}

// vamp.cpp:929-1072 -- EM for mixture weights / variances, then merge of close variances.
void Vamp::updatePrior(int verbose) {
    double noise_var = 1 / gam1;
    double lambda = 1 - probs[0];
    std::vector<double> omegas = probs;
    for (size_t j = 1; j < omegas.size(); j++) omegas[j] /= lambda;
    int it;
    for (it = 0; it < o.EM_max_iter; it++) {
        double max_sigma = *std::max_element(vars.begin(), vars.end());
        std::vector<double> probs_prev = probs, vars_prev = vars;
        const size_t Lm1 = probs.size() - 1;
        std::vector<double> beta(size_t(M) * Lm1), gammas(size_t(M) * Lm1), pin(M, 0.0), v;
        for (int i = 0; i < M; i++) {
            double sum_of_elems = 0;
            for (size_t j = 1; j < probs.size(); j++) {
                double num = lambda * omegas[j] *
                             exp(-pow(r1[i], 2) / 2 * (max_sigma - vars[j]) / (vars[j] + noise_var) / (max_sigma + noise_var)) /
                             sqrt(vars[j] + noise_var) / sqrt(2 * M_PI);
                double num_gammas = gam1 * r1[i] / (1 / vars[j] + gam1);
                beta[i * Lm1 + j - 1] = num;
                gammas[i * Lm1 + j - 1] = num_gammas;
                sum_of_elems += num;      // std::accumulate in index order (:970)
            }
            for (size_t j = 0; j < Lm1; j++) beta[i * Lm1 + j] /= sum_of_elems;
            pin[i] = 1 / (1 + (1 - lambda) / sqrt(2 * M_PI * noise_var) *
                                  exp(-pow(r1[i], 2) / 2 * max_sigma / noise_var / (noise_var + max_sigma)) / sum_of_elems);
        }
        for (size_t j = 1; j < probs.size(); j++) v.push_back(1.0 / (1.0 / vars[j] + gam1));
        lambda = std::accumulate(pin.begin(), pin.end(), 0.0);
        double lambda_total = lambda;
        if (comm) comm->allreduce_sum(&lambda_total, 1);
        lambda = lambda_total / Mt;
        for (int i = 0; i < M; i++)
            for (size_t j = 0; j < Lm1; j++)
                gammas[i * Lm1 + j] = beta[i * Lm1 + j] * (gammas[i * Lm1 + j] * gammas[i * Lm1 + j] + v[j]);
        double sum_of_pin = lambda_total;
        for (size_t j = 0; j < Lm1; j++) {
            double res = 0, res_gammas = 0;
            for (int i = 0; i < M; i++) {
                res += beta[i * Lm1 + j] * pin[i];
                res_gammas += gammas[i * Lm1 + j] * pin[i];
            }
            double tot[2] = {res_gammas, res};
            if (comm) comm->allreduce_sum(tot, 2);
            if (o.learn_vars == 1) vars[j + 1] = tot[0] / tot[1];
            omegas[j + 1] = tot[1] / sum_of_pin;
            probs[j + 1] = lambda * omegas[j + 1];
        }
        probs[0] = 1 - lambda;
        double distance_probs = 0, norm_probs = 0, distance_vars = 0, norm_vars = 0;
        for (size_t j = 0; j < probs.size(); j++) {
            distance_probs += (probs[j] - probs_prev[j]) * (probs[j] - probs_prev[j]);
            norm_probs += probs[j] * probs[j];
            distance_vars += (vars[j] - vars_prev[j]) * (vars[j] - vars_prev[j]);
            norm_vars += vars[j] * vars[j];
        }
        double dist_probs = sqrt(distance_probs / norm_probs);
        double dist_vars = sqrt(distance_vars / norm_vars);
        if (verbose == 1 && o.verbose && (!comm || comm->rank == 0))
            printf("it = %d: dist_probs = %g & dist_vars = %g\n", it, dist_probs, dist_vars);
        if (dist_probs < o.EM_err_thr && dist_vars < o.EM_err_thr) break;
    }
    // merging close variances (:1054-1071)
    for (size_t j = 0; j < vars.size(); j++) {
        for (size_t k = j + 1; k < vars.size(); k++) {
            double denom;
            if (vars[j] != 0) denom = std::min(vars[j], vars[k]);
            else denom = 1e-7;
            if (std::abs(vars[j] - vars[k]) / denom < 5e-1) {
                double sum2probs = probs[j] + probs[k];
                vars.erase(vars.begin() + k);
                probs.erase(probs.begin() + k);
                probs[j] = sum2probs;
                k--;
            }
        }
    }
}

// vamp.cpp:1074-1118 (red == 0 branch).  The all-zero shortcut (:1079) is kept: A*0 == 0 either way.
std::vector<double> Vamp::lmmse_mult(const std::vector<double>& v, double tau) {
    if (v == std::vector<double>(M, 0.0)) return std::vector<double>(M, 0.0);
    std::vector<double> res_temp = d->Ax(v.data());
    std::vector<double> res = d->ATx(res_temp.data());
    for (int i = 0; i < M; i++) {
        res[i] *= tau;
        res[i] += gam2 * v[i];
    }
    return res;
}

// vamp.cpp:1130-1229
std::vector<double> Vamp::precondCG_solver(const std::vector<double>& v, const std::vector<double>& mu_start,
                                           double tau, int denoiser) {
    std::vector<double> diag(M, 1.0);
    for (int j = 0; j < M; j++) diag[j] = tau * (N - 1) / N + gam2;
    std::vector<double> mu = mu_start;
    std::vector<double> dvec;
    std::vector<double> r = lmmse_mult(mu, tau);
    for (int i0 = 0; i0 < M; i0++) r[i0] = v[i0] - r[i0];
    std::vector<double> z(M, 0.0);
    for (int j = 0; j < M; j++) z[j] = r[j] / diag[j];
    std::vector<double> p = z;
    std::vector<double> Apalpha(M, 0.0), palpha(M, 0.0);
    double alpha, beta, prev_onsager = 0;
    std::vector<double> relres;
    int iters = 0;
    for (int i = 0; i < o.CG_max_iter; i++) {
        iters = i + 1;
        dvec = lmmse_mult(p, tau);
        alpha = inner_prod(r, z, 1, comm) / inner_prod(dvec, p, 1, comm);
        for (int j = 0; j < M; j++) palpha[j] = alpha * p[j];
        for (int j = 0; j < M; j++) mu[j] = mu[j] + palpha[j];
        if (denoiser == 0) {
            double onsager = gam2 * inner_prod(v, mu, 1, comm);
            double rel_err;
            if (onsager != 0) rel_err = std::abs((onsager - prev_onsager) / onsager);
            else rel_err = 1;
            if (rel_err < 1e-8) break;
            prev_onsager = onsager;
            if (o.verbose && (!comm || comm->rank == 0))
                printf("[CG onsager] it = %d: relative error for onsager is %.10g\n", i, rel_err);
        }
        for (int j = 0; j < M; j++) Apalpha[j] = dvec[j] * alpha;
        beta = pow(inner_prod(r, z, 1, comm), -1);
        for (int j = 0; j < M; j++) r[j] = r[j] - Apalpha[j];
        for (int j = 0; j < M; j++) z[j] = r[j] / diag[j];
        beta *= inner_prod(r, z, 1, comm);
        for (int j = 0; j < M; j++) p[j] = z[j] + beta * p[j];
        double norm_v = sqrt(l2_norm2(v, 1, comm));
        double norm_z = sqrt(l2_norm2(z, 1, comm));
        double rel_err = sqrt(l2_norm2(r, 1, comm)) / norm_v;
        double norm_mu = sqrt(l2_norm2(mu, 1, comm));
        double err_tol = 1e-5;
        relres.push_back(rel_err);
        if (o.verbose && (!comm || comm->rank == 0))
            printf("[CG] it = %d: ||r_it|| / ||RHS|| = %.10g, ||x_it|| = %.10g, ||z|| / ||RHS|| = %.10g\n", i, rel_err,
                   norm_mu, norm_z / norm_v);
        if (rel_err < err_tol) break;
    }
    last_cg_iters = iters;
    cg_relres.push_back(relres);
    if (denoiser == 1) mu_CG_last = mu;
    return mu;
}

// exp(x^2) erfc(x).  |x| < 25: libm erfc and exp are both accurate to an ulp or two and neither under/overflows;
// beyond: asymptotic series 1/(x sqrt(pi)) * sum_k (-1)^k (2k-1)!! / (2x^2)^k, whose terms fall below 1e-17 within 12 terms.
double erfcx(double x) {
    if (std::isnan(x)) return x;
    const double a = fabs(x);
    double r;
    if (a < 25.0)
        r = exp(a * a) * erfc(a);
    else {
        const double inv2 = 1.0 / (2.0 * a * a);
        double term = 1.0, sum = 1.0;
        for (int k = 1; k <= 12; k++) {
            term *= -(2.0 * k - 1.0) * inv2;
            sum += term;
        }
        r = sum / (a * sqrt(M_PI));
    }
    if (x < 0) {
        const double e = exp(a * a);
        r = std::isinf(e) ? e : 2.0 * e - r;
    }
    return r;
}

// vamp_probit.cpp:661-687 -- posterior mean of z given the probit likelihood and the cavity N(p, 1/tau1)
double Vamp::g1_bin_class(double p, double tau1, double y, double m_cov) const {
    double c = (p + m_cov) / sqrt(o.probit_var + 1.0 / tau1);
    double normalPdf_normalCdf = 2.0 / sqrt(2 * M_PI) / erfcx(-(2 * y - 1) * c / sqrt(2));
    return p + (2 * y - 1) * normalPdf_normalCdf / tau1 / sqrt(o.probit_var + 1.0 / tau1);
}
// vamp_probit.cpp:689-705
double Vamp::g1d_bin_class(double p, double tau1, double y, double m_cov) const {
    double c = (p + m_cov) / sqrt(o.probit_var + 1.0 / tau1);
    double Nc_phiyc = 2.0 / sqrt(2 * M_PI) / erfcx(-(2 * y - 1) * c / sqrt(2));
    return 1 - Nc_phiyc / (1 + tau1 * o.probit_var) * ((2 * y - 1) * c + Nc_phiyc);
}

double normal_cdf(double value) { return 0.5 * erfc(-value * M_SQRT1_2); }

bool lu_solve(std::vector<double> A, std::vector<double>& b, int n) {
    std::vector<int> piv(n);
    for (int k = 0; k < n; k++) {
        int p = k;
        double best = fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; i++)
            if (fabs(A[(size_t)i * n + k]) > best) { best = fabs(A[(size_t)i * n + k]); p = i; }
        if (best == 0.0) return false;
        if (p != k) {
            for (int j = 0; j < n; j++) std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]);
            std::swap(b[k], b[p]);
        }
        for (int i = k + 1; i < n; i++) {
            const double f = A[(size_t)i * n + k] / A[(size_t)k * n + k];
            A[(size_t)i * n + k] = f;
            for (int j = k + 1; j < n; j++) A[(size_t)i * n + j] -= f * A[(size_t)k * n + j];
            b[i] -= f * b[k];
        }
    }
    for (int i = n - 1; i >= 0; i--) {
        double s = b[i];
        for (int j = i + 1; j < n; j++) s -= A[(size_t)i * n + j] * b[j];
        b[i] = s / A[(size_t)i * n + i];
    }
    return true;
}

double Vamp::cov_dot(int i, const std::vector<double>& eta) const {
    double s = 0;
    for (int j = 0; j < o.C; j++) s += o.covs[(size_t)i * o.C + j] * eta[j];
    return s;
}

// vamp_probit.cpp:813-838 -- gradient of -logL / N of the probit regression of y on the covariates with offset gg
std::vector<double> Vamp::grad_cov(const std::vector<double>& y, const std::vector<double>& gg,
                                   const std::vector<double>& eta) const {
    const int C = o.C;
    std::vector<double> grad(C, 0.0);
    for (int j = 0; j < C; j++)
        for (int i = 0; i < N; i++) {
            double g_i = gg[i] + cov_dot(i, eta);
            double arg = (2 * y[i] - 1) / sqrt(o.probit_var) * g_i;
            double ratio = 2.0 / sqrt(2 * M_PI) / erfcx(-arg / sqrt(2));
            grad[j] += (-1) * ratio * (2 * y[i] - 1) / sqrt(o.probit_var) * o.covs[(size_t)i * C + j];
        }
    for (int j = 0; j < C; j++) grad[j] /= N;
    return grad;
}

// vamp_probit.cpp:840-857
double Vamp::mlogL_probit(const std::vector<double>& y, const std::vector<double>& gg, const std::vector<double>& eta) const {
    double mlogL = 0;
    for (int i = 0; i < N; i++) {
        double g_i = gg[i] + cov_dot(i, eta);
        double arg = (2 * y[i] - 1) / sqrt(o.probit_var) * g_i;
        mlogL -= log(normal_cdf(arg));
    }
    return mlogL / N;
}

// vamp_probit.cpp:936-1062 -- Newton-Raphson with a backtracking line search on the covariate effects.  Kept as the
// reference has them: lambda is formed WITHOUT the 1/sqrt(probit_var) that grad_cov / mlogL_probit apply (:951-957); the
// stopping test on the relative step returns the PREVIOUS eta, not eta_new (:1044-1049, `return eta`); a singular
// Hessian gives a zero step (:991-994).
std::vector<double> Vamp::Newton_method_cov(const std::vector<double>& y, const std::vector<double>& gg,
                                            std::vector<double> eta) const {
    const int C = o.C;
    const int rank0 = (!comm || comm->rank == 0);
    std::vector<double> eta_new;
    for (int it = 0; it <= 500; it++) {
        std::vector<double> H((size_t)C * C, 0.0), RHS(C, 0.0), lambda(N);
        for (int i = 0; i < N; i++) {
            double g_i = gg[i] + cov_dot(i, eta);
            double arg = (2 * y[i] - 1) * g_i;
            double ratio = 2.0 / sqrt(2 * M_PI) / erfcx(-arg / sqrt(2));
            lambda[i] = ratio * (2 * y[i] - 1);
            const double w = lambda[i] * (lambda[i] + g_i);
            for (int j = 0; j < C; j++) {
                const double zij = o.covs[(size_t)i * C + j];
                RHS[j] += zij * lambda[i];
                for (int k = 0; k < C; k++) H[(size_t)j * C + k] += zij * (o.covs[(size_t)i * C + k] * w);
            }
        }
        if (!lu_solve(H, RHS, C)) RHS.assign(C, 0.0);
        eta_new = eta;
        std::vector<double> displ(C, 0.0);
        std::vector<double> grad = grad_cov(y, gg, eta);
        double scale = 1;
        double init_val = mlogL_probit(y, gg, eta);
        for (int i = 1; i < 300; i++) {
            for (int j = 0; j < C; j++) displ[j] = scale * RHS[j];
            for (int j = 0; j < C; j++) eta_new[j] = eta[j] + displ[j];
            double curr_val = mlogL_probit(y, gg, eta_new);
            double dg = 0;
            for (int j = 0; j < C; j++) dg += displ[j] * grad[j];
            if (curr_val <= init_val + dg / 2) break;
            scale *= 0.9;
        }
        double d2 = 0, n2 = 0;
        for (int j = 0; j < C; j++) {
            d2 += (eta[j] - eta_new[j]) * (eta[j] - eta_new[j]);
            n2 += eta[j] * eta[j];
        }
        double norm_eta = sqrt(n2);
        double rel_err = (norm_eta == 0) ? 1 : sqrt(d2) / norm_eta;
        if (o.verbose && rank0) printf("[Newton_cov] it = %d, relative err = %.10g\n", it, rel_err);
        if (rel_err < 1e-4) break;
        init_val = mlogL_probit(y, gg, eta);
        eta = eta_new;
        double curr_val = mlogL_probit(y, gg, eta);
        if (curr_val > init_val) break;
    }
    return eta;
}

// vamp_probit.cpp:20-658 : generalised VAMP for y in {0,1}.  The signal side (g1/g1d,
// updatePrior, CG, Onsager probe) is shared with the linear model; the z side adds the probit denoiser over N.
// Diagnostics against the true signal (probit_err_measures, "true gam" prints) are not restated: they only print.
std::vector<double> Vamp::infere_bin_class() {
    double tau1 = gam1, tau2 = 0;
    double sqrtN = sqrt(N);
    std::vector<double> x1_hat_prev(M, 0.0);
    std::vector<double> p1 = simulate(N, std::vector<double>{1.0 / 1}, std::vector<double>{1}, 1);   // :52, default seed 1
    r1 = std::vector<double>(M, 0.0);
    r2 = r1;
    alpha1 = 0;
    std::vector<double> z1_hat(N, 0.0), p2(4 * d->mbytes, 0.0), z2_hat;
    const int rank0 = (!comm || comm->rank == 0);
    const int C = o.C;
    if (C > 0) cov_eff = std::vector<double>(C, 0.0);                    // :84-85
    std::vector<double> m_cov(N, 0.0);                                   // inner_prod(Z[i], cov_eff) of :347,:364
    for (int it = 1; it <= o.iterations; it++) {
        double t_start = now_s();
        long ax0 = d->n_ax, atx0 = d->n_atx;
        IterTrace tr;
        memset(&tr, 0, sizeof(tr));
        if (it == 1 && C > 0) {                                          // :110-126: once, with gg = z1_hat = 0
            // the reference passes its member y = get_phen() (vamp.cpp:151), where an NA phenotype is DBL_MAX and the fit
            // degenerates to inf/NaN; the filtered phenotype (NA -> 0, what the z-side denoiser sees, :320) is used here.
            // The two agree whenever no phenotype is missing.
            cov_eff = Newton_method_cov(d->filter_pheno(), z1_hat, cov_eff);
            for (int i = 0; i < N; i++) m_cov[i] = cov_dot(i, cov_eff);
        }
        double rho_it = 1;                                               // :71 (overrides :66-70)
        x1_hat_prev = x1_hat;
        double rho_it2 = o.rho, alpha1_prev = alpha1, gam1_reEst_prev;
        int it_revar = 1;
        for (; it_revar <= 50; it_revar++) {                            // :117-160 (auto_var_max_iter = 50)
            for (int i = 0; i < M; i++) x1_hat[i] = g1(r1[i], gam1);
            std::vector<double> x1_hat_m_r1 = x1_hat;
            for (int i0 = 0; i0 < M; i0++) x1_hat_m_r1[i0] -= r1[i0];
            double sum_d = 0;
            for (int i = 0; i < M; i++) sum_d += g1d(r1[i], gam1);
            alpha1 = sum_d;
            if (comm) comm->allreduce_sum(&alpha1, 1);
            alpha1 /= Mt;
            eta1 = gam1 / alpha1;
            if (it <= 1) break;
            gam1_reEst_prev = gam1;
            gam1 = std::min(std::max(1 / (1 / eta1 + l2_norm2(x1_hat_m_r1, 1, comm) / Mt), gamma_min), gamma_max);
            updatePrior(0);
            if (std::abs(gam1 - gam1_reEst_prev) < 1e-3) break;
        }
        tr.gam1_denoise = gam1;
        tr.revar_rounds = std::max(it_revar - 1, 1);
        if (it > 1) {                                                   // :197-203
            for (int i = 0; i < M; i++) x1_hat[i] = rho_it2 * x1_hat[i] + (1 - rho_it2) * x1_hat_prev[i];
            alpha1 = rho_it2 * alpha1 + (1 - rho_it2) * alpha1_prev;
        }
        {   // :205-224: _probit_it_<it>.bin, _probit_r1_it_<it>.bin hold x1_hat / sqrt(N), r1 / sqrt(N)
            std::vector<double> xs = x1_hat, rs = r1;
            for (double& v : xs) v /= sqrtN;
            for (double& v : rs) v /= sqrtN;
            x1_hist.push_back(xs);
            r1_hist.push_back(rs);
            if (!o.out_prefix.empty()) {
                store_bin(o.out_prefix + "_probit_it_" + std::to_string(it) + ".bin", xs, d->S);
                store_bin(o.out_prefix + "_probit_r1_it_" + std::to_string(it) + ".bin", rs, d->S);
            }
        }
        gam_before = gam2;
        gam2 = std::min(std::max(eta1 - gam1, gamma_min), gamma_max);   // :283
        for (int i = 0; i < M; i++) r2[i] = (eta1 * x1_hat[i] - gam1 * r1[i]) / gam2;
        tr.alpha1 = alpha1; tr.eta1 = eta1; tr.gam2 = gam2;
        std::vector<double> y = d->filter_pheno();
        double beta1 = 0;
        {   // :335-379, auto_var_max_iter = 1: one round
            for (int i = 0; i < N; i++) z1_hat[i] = g1_bin_class(p1[i], tau1, y[i], m_cov[i]);
            std::vector<double> z1_hat_m_p1 = z1_hat;
            for (int i0 = 0; i0 < N; i0++) z1_hat_m_p1[i0] -= p1[i0];
            for (int i = 0; i < N; i++) beta1 += g1d_bin_class(p1[i], tau1, y[i], m_cov[i]);
            beta1 /= N;
            double zeta1 = tau1 / beta1;
            if (it > 1) tau1 = std::min(std::max(1 / (1 / zeta1 + l2_norm2(z1_hat_m_p1, 0, comm) / N), gamma_min), gamma_max);
        }
        for (int i = 0; i < N; i++) p2[i] = (z1_hat[i] - beta1 * p1[i]) / (1 - beta1);      // :447-448
        tau2 = tau1 * (1 - beta1) / beta1;                              // :460
        tr.beta1 = beta1;
        if (o.verbose && rank0) printf("beta1 = %.10g\ntau2 = %.10g\n", beta1, tau2);
        std::vector<double> v = d->ATx(p2.data());                       // :492-497
        for (int i = 0; i < M; i++) v[i] = tau2 * v[i] + gam2 * r2[i];
        x2_hat = precondCG_solver(v, std::vector<double>(M, 0.0), tau2, 1);
        tr.cg_iters = last_cg_iters;
        {
            std::vector<double> xs = x2_hat;
            for (double& u : xs) u /= sqrtN;
            x2_hist.push_back(xs);
        }
        double alpha2_ = g2d_onsager(gam2, tau2);                        // :512
        tr.onsager_iters = last_cg_iters;
        tr.alpha2 = alpha2_;
        eta2 = gam2 / alpha2_;
        std::vector<double> x2_hat_m_r2 = x2_hat;
        for (int i0 = 0; i0 < M; i0++) x2_hat_m_r2[i0] -= r2[i0];
        if (it > 1) gam2 = std::min(std::max(1 / (1 / eta2 + l2_norm2(x2_hat_m_r2, 1, comm) / Mt), gamma_min), gamma_max);
        tr.eta2 = eta2; tr.gam2_reest = gam2;
        std::vector<double> r1_prev = r1;
        for (int i = 0; i < M; i++) r1[i] = rho_it * (x2_hat[i] - alpha2_ * r2[i]) / (1 - alpha2_) + (1 - rho_it) * r1_prev[i];
        double gam1_prev = gam1;
        gam1 = gam2 * (1 - alpha2_) / alpha2_;                           // :548-549
        gam1 = rho_it * gam1 + (1 - rho_it) * gam1_prev;
        tr.gam1_next = gam1;
        z2_hat = d->Ax(x2_hat.data());                                   // :555
        double beta2 = (double)Mt / N * (1 - alpha2_);                   // :561
        std::vector<double> z2_hat_m_p2 = z2_hat;
        for (int i0 = 0; i0 < N; i0++) z2_hat_m_p2[i0] -= p2[i0];
        z2_hat_m_p2.resize(N);
        double zeta2 = tau2 / beta2;
        if (it > 1) tau2 = 1.0 / (1.0 / zeta2 + l2_norm2(z2_hat_m_p2, 0, comm) / N);   // :573-574
        std::vector<double> p1_prev = p1;
        for (int i = 0; i < N; i++) p1[i] = rho_it * (z2_hat[i] - beta2 * p2[i]) / (1 - beta2) + (1 - rho_it) * p1_prev[i];
        double tau1_prev = tau1;
        tau1 = tau2 * (1 - beta2) / beta2;                               // :594-595
        tau1 = rho_it * tau1 + (1 - rho_it) * tau1_prev;
        tr.tau2 = tau2; tr.tau1_next = tau1;
        tr.L_after = (int)probs.size();
        tr.n_ax = d->n_ax - ax0;
        tr.n_atx = d->n_atx - atx0;
        tr.seconds = now_s() - t_start;
        trace.push_back(tr);
        std::vector<double> x1_hat_diff(M, 0.0);
        for (int i0 = 0; i0 < M; i0++) x1_hat_diff[i0] = x1_hat_prev[i0] - x1_hat[i0];
        double rel_err_x1 = sqrt(l2_norm2(x1_hat_diff, 1, comm) / l2_norm2(x1_hat_prev, 1, comm));
        if (it > 1 && rel_err_x1 < o.stop_criteria_thr) break;           // :636-640
    }
    return x1_hat;                                                        // unscaled (:657)
}

// denoiserXXT.cpp:15-35: (tau A A^T + gam2 I) u, matrix-free (ATx then Ax)
std::vector<double> Vamp::lmmse_multAAT(const std::vector<double>& u, double tau) {
    size_t phen_size = 4 * d->mbytes;
    bool all_zero = true;
    for (double x : u) if (x != 0.0) { all_zero = false; break; }
    if (all_zero) return std::vector<double>(phen_size, 0.0);
    std::vector<double> up(u);
    up.resize(phen_size, 0.0);
    std::vector<double> res_temp = d->ATx(up.data());
    std::vector<double> res = d->Ax(res_temp.data());
    for (int i = 0; i < N; i++) {
        res[i] *= tau;
        res[i] += gam2 * u[i];
    }
    return res;
}

// denoiserXXT.cpp:52-130: CG in N-space with the per-individual diagonal preconditioner, tolerance 1e-4,
// un-synchronised inner products (N-vectors are replicated over ranks).
std::vector<double> Vamp::CG_solverAAT(const std::vector<double>& v, const std::vector<double>& mu_start, double tau,
                                       int save) {
    int mbytes4 = 4 * (int)d->mbytes;
    const std::vector<double>&mave_people = d->mave_people, &msig_people = d->msig_people, &numb_people = d->numb_people;
    std::vector<double> diag(N, 1.0);
    for (int i = 0; i < N; i++)
        diag[i] = tau * ((numb_people[i] - 1) / msig_people[i] / msig_people[i] + mave_people[i] * mave_people[i] * numb_people[i]) / N + gam2;
    std::vector<double> mu = mu_start;
    mu.resize(mbytes4, 0.0);
    std::vector<double> dd;
    std::vector<double> r = lmmse_multAAT(mu, tau);
    for (int i0 = 0; i0 < N; i0++) r[i0] = v[i0] - r[i0];
    std::vector<double> z(mbytes4, 0.0);
    for (int i0 = 0; i0 < N; i0++) z[i0] = r[i0] / diag[i0];
    std::vector<double> p = z, Apalpha(mbytes4, 0.0), palpha(mbytes4, 0.0);
    double alpha, beta;
    std::vector<double> vpad(v);
    vpad.resize(mbytes4, 0.0);
    std::vector<double> relres;
    int iters = 0;
    for (int i = 0; i < o.CG_max_iter; i++) {
        iters = i + 1;
        dd = lmmse_multAAT(p, tau);
        alpha = inner_prod(r, z, 0, comm) / inner_prod(dd, p, 0, comm);
        for (int j = 0; j < N; j++) palpha[j] = alpha * p[j];
        for (size_t j = 0; j < mu.size(); j++) mu[j] += palpha[j];
        for (int j = 0; j < N; j++) Apalpha[j] = dd[j] * alpha;
        beta = pow(inner_prod(r, z, 0, comm), -1);
        for (size_t j = 0; j < r.size(); j++) r[j] -= Apalpha[j];
        for (int j = 0; j < N; j++) z[j] = r[j] / diag[j];
        beta *= inner_prod(r, z, 0, comm);
        for (int j = 0; j < N; j++) p[j] = z[j] + beta * p[j];
        double rel_err = sqrt(l2_norm2(r, 0, comm) / l2_norm2(vpad, 0, comm));
        relres.push_back(rel_err);
        if (o.verbose && (!comm || comm->rank == 0)) printf("[CG] it = %d: ||r_it|| / ||RHS|| = %.10g\n", i, rel_err);
        if (rel_err < 1e-4) break;
    }
    last_cg_iters = iters;
    cg_relres.push_back(relres);
    if (save == 1) mu_CG_last = mu;
    return mu;
}

// denoiserXXT.cpp:37-50: x2 = r2 + gamw A^T u, (gamw A A^T + gam2 I) u = y - A r2
std::vector<double> Vamp::lmmse_denoiserAAT(const std::vector<double>& r2_, const std::vector<double>& mu_last) {
    std::vector<double> z2 = d->Ax(r2_.data());
    std::vector<double> v(N, 0.0);
    for (int i = 0; i < N; i++) v[i] = y[i] - z2[i];
    std::vector<double> u = CG_solverAAT(v, mu_last, gamw, 1);
    std::vector<double> res = d->ATx(u.data());
    for (int i = 0; i < M; i++) res[i] = gamw * res[i] + r2_[i];
    return res;
}

// vamp.cpp:871-889 -- Hutchinson probe u in {+-1/sqrt(Mt)}^M from mt19937{seed + S} + bernoulli(0.5).
double Vamp::g2d_onsager(double gam2_, double tau) {
    std::mt19937 rd{o.seed + (unsigned long)d->S};
    std::bernoulli_distribution bern(0.5);
    bern_vec = std::vector<double>(M, 0.0);
    for (int i = 0; i < M; i++) bern_vec[i] = (2 * bern(rd) - 1) / sqrt(Mt);
    invQ_bern_vec = precondCG_solver(bern_vec, std::vector<double>(M, 0.0), tau, 0);
    return gam2_ * inner_prod(bern_vec, invQ_bern_vec, 1, comm);
}

// vamp.cpp:892-927
void Vamp::updateNoisePrec() {
    y = d->filter_pheno();
    std::vector<double> temp = d->Ax(x2_hat.data());
    for (int i = 0; i < N; i++) temp[i] -= y[i];
    double temp_norm2 = l2_norm2(temp, 0, comm);
    std::vector<double> trace_corr_vec_N = d->Ax(invQ_bern_vec.data());
    std::vector<double> trace_corr_vec_M = d->ATx(trace_corr_vec_N.data());
    double trace_corr = inner_prod(bern_vec, trace_corr_vec_M, 1, comm) * Mt;
    if (o.verbose && (!comm || comm->rank == 0)) {
        printf("l2_norm2(temp) / N = %.10g\n", temp_norm2 / N);
        printf("trace_correction / N = %.10g\n", trace_corr / N);
    }
    gamw = (double)N / (temp_norm2 + trace_corr);
}

// vamp.cpp:1232-1318 (redglob == 0): only R2trains feeds back into outputs; the Ax call for ind == 2
// (:1301) is kept because it is part of the per-iteration matvec count.
void Vamp::err_measures(int ind) {
    y = d->filter_pheno();
    std::vector<double> Axest;
    if (ind == 1) {
        if (z1.size() > 0) Axest = z1;
        else Axest = d->Ax(x1_hat.data());
    } else
        Axest = d->Ax(x2_hat.data());
    std::vector<double> tempNest(4 * d->mbytes, 0.0);
    for (int i = 0; i < N; i++) tempNest[i] = -Axest[i] + y[i];
    double l2_pred_err = sqrt(l2_norm2(tempNest, 0, comm) / l2_norm2(y, 0, comm));
    double R2 = 1 - l2_pred_err * l2_pred_err;
    R2trains.push_back(R2);
    if (o.verbose && (!comm || comm->rank == 0)) printf("R2 = %.10g\n", R2);
}

static void store_bin(const std::string& path, const std::vector<double>& v, int S) {
    // utilities.cpp:293-301 -- raw native doubles at byte offset S*8 of a shared file.
    FILE* f = fopen(path.c_str(), "r+b");
    if (!f) f = fopen(path.c_str(), "w+b");
    if (!f) return;
    fseek(f, (long)S * 8, SEEK_SET);
    fwrite(v.data(), 8, v.size(), f);
    fclose(f);
}

// vamp.cpp:149-183 (infere: vars *= N, model dispatch) + :190-803 (infere_linear; no restart, no
// init_est, optional freeze mask, no cross-val, reverse == 0).  Statement order follows the reference line by line.
std::vector<double> Vamp::infere() {
    y = d->phen;
    for (size_t i = 0; i < vars.size(); i++) vars[i] *= N;
    if (o.use_XXT_denoiser == 1) d->compute_people_statistics();        // vamp.cpp:169-170
    if (o.bin_class) return infere_bin_class();                         // vamp.cpp:175-176

    std::vector<double> x1_hat_d(M, 0.0), x1_hat_stored(M, 0.0), x1_hat_prev(M, 0.0);
    alpha1 = 0;
    alpha2 = 0;   // uninitialised member in the reference (read at :501 before :631); SURVEY App. B
    std::vector<double> y = d->filter_pheno();
    r1 = std::vector<double>(M, 0.0);
    const int rank0 = (!comm || comm->rank == 0);
    if (!o.r1_init.empty()) {            // restart (vamp.cpp:226-233): gam1 / gamw come from the options (ctor), and the
        r1 = o.r1_init;                  // stored r1 / sqrt(N) is divided by sqrt(N) once more
        for (int i = 0; i < M; i++) r1[i] /= sqrt(N);
    }
    const bool init_est = !o.x_init.empty();
    if (init_est) {                      // vamp.cpp:244-258
        std::vector<double> x_est = o.x_init;
        for (size_t i0 = 0; i0 < x_est.size(); i0++) x_est[i0] *= sqrt((double)N);
        x1_hat = x_est;
        r1 = x_est;
    }

    for (int it = 1; it <= o.iterations; it++) {
        double t_start = now_s();
        long ax0 = d->n_ax, atx0 = d->n_atx;
        IterTrace tr;
        memset(&tr, 0, sizeof(tr));
        x1_hat_prev = x1_hat;
        double alpha1_prev = alpha1;
        double gam1_reEst_prev;
        int it_revar = 1;
        for (; it_revar <= auto_var_max_iter; it_revar++) {            // :289-338
            for (int i = 0; i < M; i++) x1_hat[i] = g1(r1[i], gam1);
            if (it == 1 && init_est) x1_hat = r1;                          // :295-296
            std::vector<double> x1_hat_m_r1 = x1_hat;
            for (size_t i0 = 0; i0 < x1_hat_m_r1.size(); i0++) x1_hat_m_r1[i0] = x1_hat_m_r1[i0] - r1[i0];
            double sum_d = 0;
            const bool use_freeze = !o.freeze_ind.empty();                 // :205-209 (--use-freeze 1)
            for (int i = 0; i < M; i++) {
                x1_hat_d[i] = g1d(r1[i], gam1);
                if (!use_freeze || (use_freeze && o.freeze_ind[i] == 0)) sum_d += x1_hat_d[i];   // :308
            }
            alpha1 = sum_d;
            if (comm) comm->allreduce_sum(&alpha1, 1);
            alpha1 /= Mt;
            eta1 = gam1 / alpha1;
            if (it <= 1) break;
            gam1_reEst_prev = gam1;
            gam1 = std::min(std::max(1.0 / (1.0 / eta1 + l2_norm2(x1_hat_m_r1, 1, comm) / Mt), gamma_min), gamma_max);
            updatePrior(0);
            if (o.verbose && rank0) printf("[old] it_revar = %d: gam1 = %.10g\n", it_revar, gam1);
            if (std::abs(gam1 - gam1_reEst_prev) < 1e-3) break;
        }
        gam1s.push_back(gam1);
        tr.gam1_denoise = gam1;
        tr.revar_rounds = std::max(it_revar - 1, 1);

        if (it > 1) {                                                   // :348-423
            const bool use_freeze = !o.freeze_ind.empty();
            for (int i = 0; i < M; i++)
                if (!use_freeze || (use_freeze && o.freeze_ind[i] == 0))   // :353: frozen markers are not damped
                    x1_hat[i] = rho * x1_hat[i] + (1 - rho) * x1_hat_prev[i];
            alpha1 = rho * alpha1 + (1 - rho) * alpha1_prev;
        }
        z1 = d->Ax(x1_hat.data());                                      // :429

        double scale = sqrt(N);
        for (size_t i0 = 0; i0 < x1_hat_stored.size(); i0++) x1_hat_stored[i0] = x1_hat[i0] / scale;
        std::vector<double> r1_stored = r1;
        for (size_t i0 = 0; i0 < r1_stored.size(); i0++) r1_stored[i0] = r1[i0] / scale;
        x1_hist.push_back(x1_hat_stored);
        r1_hist.push_back(r1_stored);
        if (!o.out_prefix.empty()) {
            store_bin(o.out_prefix + "_it_" + std::to_string(it) + ".bin", x1_hat_stored, d->S);
            store_bin(o.out_prefix + "_r1_it_" + std::to_string(it) + ".bin", r1_stored, d->S);
        }

        gam_before = gam2;
        gam2 = std::min(std::max(eta1 - gam1, gamma_min), gamma_max);   // :472
        r2_prev = r2;                                                   // :483
        for (int i = 0; i < M; i++) r2[i] = (eta1 * x1_hat[i] - gam1 * r1[i]) / gam2;
        if (o.use_lmmse_damp == 1) {                                    // :488-498
            double xi = std::min(2 * rho, 1.0);
            if (it > 1) gam2 = 1.0 / pow(xi / sqrt(gam2) + (1 - xi) / sqrt(gam_before), 2);
        }
        double xi = std::min(2 * std::min(alpha1, alpha2), 1.0);        // :501-502
        rho = std::max(rho, xi);
        tr.alpha1 = alpha1;
        tr.eta1 = eta1;
        tr.gam2 = gam2;
        tr.rho = rho;
        if (o.verbose && rank0) printf("eta1 = %.10g\ngam2 = %.10g\n", eta1, gam2);

        if (auto_var_max_iter == 0 || it <= 1) updatePrior(1);          // :518-519
        err_measures(1);                                                // :525

        if (!o.out_prefix.empty()) {
            std::vector<double> r2_stored = r2;
            for (size_t i0 = 0; i0 < r2_stored.size(); i0++) r2_stored[i0] = r2[i0] / scale;
            store_bin(o.out_prefix + "_r2_it_" + std::to_string(it) + ".bin", r2_stored, d->S);
        }

        // ---- LMMSE step (:584-596)
        if (o.use_XXT_denoiser == 0) {
            std::vector<double> v = d->ATx(y.data());
            for (int i = 0; i < M; i++) v[i] = gamw * v[i] + gam2 * r2[i];
            if (it == 1) x2_hat = precondCG_solver(v, std::vector<double>(M, 0.0), gamw, 1);
            else x2_hat = precondCG_solver(v, mu_CG_last, gamw, 1);
        } else {                                                        // vamp.cpp:599-606
            if (it == 1) mu_CG_last = std::vector<double>(4 * d->mbytes, 0.0);
            x2_hat = lmmse_denoiserAAT(r2, mu_CG_last);
        }
        tr.cg_iters = last_cg_iters;
        std::vector<double> x2_hat_stored = x2_hat;
        for (size_t i0 = 0; i0 < x2_hat_stored.size(); i0++) x2_hat_stored[i0] = x2_hat[i0] / scale;
        x2_hist.push_back(x2_hat_stored);
        if (!o.out_prefix.empty())
            store_bin(o.out_prefix + "_it_" + std::to_string(it) + "_x2_hat.bin", x2_hat_stored, d->S);

        alpha2 = g2d_onsager(gam2, gamw);                               // :631
        tr.onsager_iters = last_cg_iters;
        tr.alpha2 = alpha2;
        if (o.verbose && rank0) printf("alpha2 = %.10g\n", alpha2);

        if (it > 1) {
            // :646-681 "polynomial onsager" diagnostics: printed only, but they cost 3 Ax per iteration,
            // which belongs to the reference's per-iteration work.
            std::vector<double> Xr2 = d->Ax(r2.data());
            std::vector<double> Xx2hat_m_y = d->Ax(x2_hat.data());
            std::vector<double> Xx2b = d->Ax(x2_hat.data());
            (void)Xr2; (void)Xx2hat_m_y; (void)Xx2b;
        }
        eta2 = gam2 / alpha2;                                           // :682
        std::vector<double> x2_hat_m_r2 = x2_hat;
        for (size_t i0 = 0; i0 < x2_hat_m_r2.size(); i0++) x2_hat_m_r2[i0] = x2_hat_m_r2[i0] - r2[i0];
        if (auto_var_max_iter >= 1 && it > 2)
            gam2 = std::min(std::max(1 / (1 / eta2 + l2_norm2(x2_hat_m_r2, 1, comm) / Mt), gamma_min), gamma_max);
        gam2s.push_back(gam2);
        tr.eta2 = eta2;
        tr.gam2_reest = gam2;
        gam1 = std::min(std::max(eta2 - gam2, gamma_min), gamma_max);   // :702
        for (int i = 0; i < M; i++) r1[i] = (eta2 * x2_hat[i] - gam2 * r2[i]) / gam1;
        tr.gam1_next = gam1;
        if (o.verbose && rank0) printf("gam2 re-est = %.10g\ngam1 = %.10g\n", gam2, gam1);

        updateNoisePrec();                                              // :726
        tr.gamw = gamw;
        err_measures(2);                                                // :731
        if (o.verbose && rank0) printf("gamw = %.10g\n", gamw);

        tr.L_after = (int)probs.size();
        tr.n_ax = d->n_ax - ax0;
        tr.n_atx = d->n_atx - atx0;
        tr.seconds = now_s() - t_start;
        trace.push_back(tr);

        std::vector<double> x1_hat_diff = x1_hat;                       // :741-749
        for (size_t i0 = 0; i0 < x1_hat_diff.size(); i0++) x1_hat_diff[i0] = x1_hat_prev[i0] - x1_hat_diff[i0];
        if (it > 1 && sqrt(l2_norm2(x1_hat_diff, 1, comm) / l2_norm2(x1_hat_prev, 1, comm)) < o.stop_criteria_thr) break;
    }
    return x1_hat_stored;
}

}  // namespace gvo
