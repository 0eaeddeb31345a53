// harness.cpp -- golden-vector dumper that links the REAL reference translation units (data.cpp, vamp.cpp,
// utilities.cpp, options.cpp from /root/reference, compiled where they lie by build_ref.sh) and calls their functions
// directly.  Test infrastructure, container-only: it is built only when real Boost headers exist (the reference includes
// <boost/math/distributions/students_t.hpp>, data.cpp:17 / vamp.cpp:19 / utilities.cpp:8) and nothing of it travels.
//
//   harness <bed-file> <N> <M> <phen-file|-> <out-prefix> <seed>
//
// writes, as raw native-endian doubles unless noted (SURVEY 8c's list):
//   <out>_mave.bin, <out>_msig.bin          data::compute_markers_statistics      data.cpp:392-546
//   <out>_x.bin, <out>_Ax.bin               data::Ax(x), x ~ N(0,1) mt19937(seed)    data.cpp:848-1009
//   <out>_p.bin, <out>_ATx.bin              data::ATx(p), p filtered N(0,1)          data.cpp:810-835
//   <out>_g1_grid.txt                       vamp::g1 / g1d on a (gam1, y) grid, 3-component and default 23-component prior
//   <out>_prior_{in,out}.txt                one vamp::updatePrior(0) step: gam1, probs, vars before / after   vamp.cpp:929-1072
//   <out>_cg_{v,mu}.bin, <out>_cg_trace.txt one vamp::precondCG_solver solve (denoiser = 1) and its printed residual trace
//                                           (captured from stdout by make_fixtures.py)                       vamp.cpp:1130-1229
//   <out>_pvals.bin                         data::pvals_calc (LOO) for one estimator                          data.cpp:1108-1226
//   <out>_people_{mave,msig,numb}.bin       data::compute_people_statistics                                   data.cpp:558-716
//
// The private state the calls need (r1, gam1, probs, vars, ...) is reached with -Dprivate=public on THIS file's includes
// only; the reference sources themselves are compiled untouched.
#include <mpi.h>

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <random>
#include <string>
#include <vector>

#define private public
#include "data.hpp"
#include "vamp.hpp"
#undef private
#include "utilities.hpp"

static void dump(const std::string& path, const double* v, size_t n) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f || fwrite(v, sizeof(double), n, f) != n) { perror(path.c_str()); exit(2); }
    fclose(f);
}
static void dump(const std::string& path, const std::vector<double>& v) { dump(path, v.data(), v.size()); }

int main(int argc, char** argv) {
    if (argc < 7) { fprintf(stderr, "usage: harness bed N M phen|- out seed\n"); return 1; }
    int prov;
    MPI_Init_thread(&argc, &argv, MPI_THREAD_MULTIPLE, &prov);
    const std::string bed = argv[1], phen = argv[4], out = argv[5];
    const int N = atoi(argv[2]), M = atoi(argv[3]);
    const unsigned long seed = strtoul(argv[6], nullptr, 10);
    const size_t n4 = 4 * (size_t)((N + 3) / 4);
    std::mt19937 gen(seed);
    std::normal_distribution<double> nrm(0.0, 1.0);

    // ---- dataset: vector-phenotype ctor (all present; data.cpp:69-113) or file ctor with NA handling (data.cpp:30-61)
    std::vector<double> y0(N);
    for (double& v : y0) v = nrm(gen);
    data* d = (phen == "-") ? new data(y0, bed, N, M, M, 0, 0) : new data(phen, bed, N, M, M, 0, 0);
    dump(out + "_mave.bin", d->get_mave(), M);
    dump(out + "_msig.bin", d->get_msig(), M);

    std::vector<double> x(M);
    for (double& v : x) v = nrm(gen);
    dump(out + "_x.bin", x);
    dump(out + "_Ax.bin", d->Ax(x.data()));

    std::vector<double> yf = d->filter_pheno();          // zero at NA individuals (data.cpp:1065-1079)
    yf.resize(n4, 0.0);
    std::vector<double> p(n4, 0.0);
    for (int i = 0; i < N; i++) p[i] = (phen == "-" || yf[i] != 0.0) ? nrm(gen) : 0.0;
    dump(out + "_p.bin", p);
    dump(out + "_ATx.bin", d->ATx(p.data()));

    d->compute_people_statistics();
    dump(out + "_people_mave.bin", d->get_mave_people());
    dump(out + "_people_msig.bin", d->get_msig_people());
    dump(out + "_people_numb.bin", d->get_numb_people());

    // ---- vamp pieces
    std::vector<double> vars3{0, 2.0, 20.0}, probs3{0.9, 0.07, 0.03};
    {
        vamp v(N, M, M, 1e-8, 2.0, 1, 0.5, vars3, probs3, std::vector<double>(M, 0.0), 0, "/tmp/", "h", "linear");
        std::ofstream g(out + "_g1_grid.txt");
        g.precision(17);
        const double gs[3] = {1e-8, 0.7, 12.5}, ys[5] = {-3.0, -0.4, 0.0, 0.05, 6.0};
        for (double gg : gs) for (double yy : ys) g << gg << ' ' << yy << ' ' << v.g1(yy, gg) << ' ' << v.g1d(yy, gg) << '\n';
        // one updatePrior step on a seeded r1
        v.gam1 = 3.5;
        v.r1.assign(M, 0.0);
        for (int i = 0; i < M; i++) v.r1[i] = nrm(gen) * (i % 10 == 0 ? 3.0 : 0.6);
        dump(out + "_prior_r1.bin", v.r1);
        std::ofstream pi(out + "_prior_in.txt"), po(out + "_prior_out.txt");
        pi.precision(17); po.precision(17);
        pi << v.gam1 << '\n';
        for (size_t j = 0; j < v.probs.size(); j++) pi << v.probs[j] << ' ' << v.vars[j] << '\n';
        v.updatePrior(0);
        for (size_t j = 0; j < v.probs.size(); j++) po << v.probs[j] << ' ' << v.vars[j] << '\n';
        // one CG solve: (tau A^T A + gam2 I) mu = rhs, cold start, LMMSE stopping rule; the trace goes to stdout
        v.gam2 = 1.35;
        v.mu_CG_last.assign(M, 0.0);
        std::vector<double> rhs(M);
        for (double& q : rhs) q = nrm(gen);
        dump(out + "_cg_v.bin", rhs);
        std::cout << "BEGIN_CG_TRACE" << std::endl;
        std::vector<double> mu = v.precondCG_solver(rhs, std::vector<double>(M, 0.0), 2.0, 1, d);
        std::cout << "END_CG_TRACE" << std::endl;
        dump(out + "_cg_mu.bin", mu);
        // LOO p-values for one estimator (needs a real Student-t: this is where Boost is used, utilities.cpp:330-331)
        std::vector<double> x1(M, 0.0);
        for (int i = 0; i < M; i += 17) x1[i] = nrm(gen);
        std::vector<double> z1 = d->Ax(x1.data());
        std::vector<std::vector<double>> pv = d->pvals_calc({z1}, yf, {x1}, {out + "_pvals.bin"});
        dump(out + "_pvals_x1.bin", x1);
    }
    if (M > 50000) {   // default 23-component prior (utilities.cpp:91-140) needs Mt > 50 000
        std::vector<double> pr, vr;
        initialize_prior(pr, vr, N, M, 0);
        for (double& q : vr) q *= N;
        vamp v(N, M, M, 1e-8, 2.0, 1, 0.5, vr, pr, std::vector<double>(M, 0.0), 0, "/tmp/", "h", "linear");
        std::ofstream g(out + "_g1_grid23.txt");
        g.precision(17);
        for (double gg : {1e-8, 0.7, 12.5}) for (double yy : {-3.0, -0.4, 0.0, 0.05, 6.0}) g << gg << ' ' << yy << ' ' << v.g1(yy, gg) << ' ' << v.g1d(yy, gg) << '\n';
    }
    delete d;
    MPI_Finalize();
    return 0;
}
