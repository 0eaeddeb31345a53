// gv_pval_dev.h -- device functions of the per-marker association test (data::pvals_calc / pvals_calc_LOCO, data.cpp:1150-1226,
// :1290-1353; linear_reg1d_pvals, utilities.cpp:321-334), shared by the stand-alone test kernel (gv_kernels.hip) and the epilogue of
// the p-value pass (gv_mfma.hip: k_fin_pvals).
#pragma once
#include <hip/hip_runtime.h>

namespace gvp {

// Student-t two-sided tail P(|T_nu| > t) = I_x(nu/2, 1/2), x = nu / (nu + t^2); the reference calls Boost's students_t
// (utilities.cpp:330-331).
//
// Large samples (a = nu/2 >= 15, t^2 <= 0.42 nu): the asymptotic expansion of the incomplete beta function for large a and small b
// of DiDonato & Morris (ACM TOMS 18 (1992), section 9; what Boost's ibeta uses in this corner),
//     I_x(a, b) = [h Gamma(a + b) / (Gamma(a) T^b)] sum_n p_n J_n ,   T = a + (b - 1)/2 ,  u = -T ln x ,  h = u^b e^-u / Gamma(b) ,
//     J_0 = Q(b, u) / h ,   J_{n+1} = [(b + 2n)(b + 2n + 1) J_n + (u + b + 2n + 1) (ln x / 2)^{2n}] / (4 T^2) ,
// with b = 1/2: Q(1/2, u) = erfc(sqrt u), and p_n constants (below).  The terms fall like 1 / (4 T^2): three or four of them reach
// 1e-16 at the sample sizes of a genotype panel, where the Lentz continued fraction needs O(sqrt(nu)) ~ 450 steps of four fp64
// divisions -- 7.4 ms for the 1M markers of the headline shard against 15 ms for the pass over the shard itself.  Checked against
// scipy.stats.t.sf for nu = 28 ... 1e6, t = 1e-6 ... 37: relative error <= 4e-15 * (a few) (tests/test_gpu_pvals.py runs the kernel).
// p_n for b = 1/2: p_0 = 1, p_n = (b - 1)/(2n + 1)! + (1/n) sum_{m=1}^{n-1} (m b - n) p_{n-m} / (2m + 1)!
__device__ __forceinline__ double bgrat_half_pn(int n) {
    const double P[10] = {1.0, -8.333333333333333e-02, 6.25e-03, -5.042989417989418e-04, 4.343722442680776e-05, -3.896385732323232e-06,
                          3.583335463450789e-07, -3.349747072060576e-08, 3.167585434316112e-09, -3.021036851737094e-10};
    return P[n];
}

// Lentz continued fraction of the incomplete beta function (DLMF 8.17.22): small samples and the far corner t^2 > 0.42 nu
__device__ inline double beta_cf(double a, double b, double x) {
    const double tiny = 1e-300;
    double c = 1.0, d = 1.0 - (a + b) * x / (a + 1.0);
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 100000; m++) {
        const double m2 = 2.0 * m;
        double num = m * (b - m) * x / ((a - 1.0 + m2) * (a + m2));
        d = 1.0 + num * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + num / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        num = -(a + m) * (a + b + m) * x / ((a + m2) * (a + 1.0 + m2));
        d = 1.0 + num * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + num / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double delta = d * c;
        h *= delta;
        if (fabs(delta - 1.0) < 1e-16) break;
    }
    return h;
}
__device__ inline double t_two_sided(double t, double nu) {
    if (isnan(t) || !(nu > 0)) return NAN;
    if (t == 0) return 1.0;
    if (isinf(t)) return 0.0;
    const double a = 0.5 * nu, b = 0.5, w = t * t / nu, x = 1.0 / (1.0 + w);
    double lnB;   // -ln B(a, 1/2); asymptotic series of lgamma(a + 1/2) - lgamma(a) for large a (DLMF 5.11.13)
    if (a >= 30.0) {
        const double ia = 1.0 / a, ia2 = ia * ia;
        lnB = 0.5 * log(a) - ia * (1.0 / 8 - ia2 * (1.0 / 192 - ia2 * (1.0 / 640 - ia2 * (17.0 / 14336)))) - lgamma(0.5);
    } else
        lnB = lgamma(a + b) - lgamma(a) - lgamma(b);
    if (a >= 15.0 && w <= 0.42) {
        const double lx = -log1p(w), T = a - 0.25, u = -T * lx;
        const double su = sqrt(u), h = su * exp(-u) * 0.5641895835477563;      // u^(1/2) e^-u / Gamma(1/2)
        if (h == 0.0) return 0.0;                                              // (p below 1e-300)
        const double q = erfc(su);
        double j = q / h, lxp = 1.0, b2n = b, sum = 0.0;
        const double lx2 = 0.25 * lx * lx, t4 = 4.0 * T * T;
        for (int n = 1; n < 10; n++) {
            j = (b2n * (b2n + 1.0) * j + (u + b2n + 1.0) * lxp) / t4;
            lxp *= lx2;
            b2n += 2.0;
            const double r = bgrat_half_pn(n) * j;
            sum += r;
            if (fabs(r) * h < 1e-17 * q) break;
        }
        // Gamma(a + 1/2) / (Gamma(a) sqrt T) = exp(lnB + ln Gamma(1/2) - ln(T)/2)
        return exp(lnB + 0.5723649429247001 - 0.5 * log(T)) * (q + h * sum);
    }
    const double front = exp(lnB - a * log1p(w) + b * (log(w) - log1p(w)));
    if (x < (a + 1.0) / (a + b + 2.0)) return front * beta_cf(a, b, x) / a;
    return 1.0 - front * beta_cf(b, a, w / (1.0 + w)) / b;
}
// utilities.cpp:321-334
__device__ inline double reg1d_pval(double sumx, double sumsqx, double sumxy, double sumy, double sumsqy, double n) {
    const double s2y = (sumsqy - sumy * sumy / n) / (n - 1), s2x = (sumsqx - sumx * sumx / n) / (n - 1);
    const double sxy = (sumxy - sumx * sumy / n) / (n - 1);
    const double rxy = sxy / sqrt(s2x * s2y);
    const double t = rxy * sqrt((n - 2) / (1 - rxy * rxy));
    return t_two_sided(fabs(t), n - 2);
}
// The test of one marker from the sums of a p-value pass: s4 = {sum a p, sum b p, sum a p^2, sum b p^2} of the phenotype residual p
// over the marker's column; the column's own sums are exact, from the genotype counts (n2, n1, n0 present individuals with a = 2,
// 1, 0).  cself != 0 (leave-one-out): the marker's own effect is added back analytically (data.cpp:1145-1148).
__device__ inline double marker_pval(double n2, double n1, double n0, double mu, double sg, const double (&s4)[4], double cself) {
    const double count = n0 + n1 + n2;
    const double sumx = sg * (2.0 * n2 + n1 - mu * count);
    const double sumsqx = sg * sg * (n2 * (2.0 - mu) * (2.0 - mu) + n1 * (1.0 - mu) * (1.0 - mu) + n0 * mu * mu);
    const double svy = sg * (s4[0] - mu * s4[1]);                            // sum value * y
    const double sumxy = svy + cself * sumsqx;
    const double sumy = s4[1] + cself * sumx;
    const double sumsqy = s4[3] + 2.0 * cself * svy + cself * cself * sumsqx;
    return reg1d_pval(sumx, sumsqx, sumxy, sumy, sumsqy, count);
}

}  // namespace gvp
