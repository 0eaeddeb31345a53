// harness.cpp -- C entry points over the `data` class of data_binding.cpp so that tests/test_gpu_binding.py can drive it through
// ctypes (numpy arrays in, numpy arrays out) and compare with the oracle.  Test infrastructure.
#include <chrono>
#include <cstring>

#include "data_binding.hpp"

extern "C" {

// the sum standing in for MPI_Allreduce (nranks > 1 only)
void bh_set_transport(gv_allreduce_fn fn, void* user) {
    binding_sum = fn;
    binding_sum_user = user;
}

void* bh_create(unsigned char* bed_slab, const unsigned char* mask4_or_null, int nonas, int N, int M, int Mt, int S, int rank,
                int nranks, double alpha_scale) {
    return new data(bed_slab, mask4_or_null, nonas, N, M, Mt, S, rank, nranks, alpha_scale);
}
void bh_destroy(void* h) { delete static_cast<data*>(h); }

void bh_stats(void* h, int M, double* mave, double* msig) {
    data* d = static_cast<data*>(h);
    memcpy(mave, d->get_mave(), sizeof(double) * (size_t)M);
    memcpy(msig, d->get_msig(), sizeof(double) * (size_t)M);
}
// one call each through the class, exactly as vamp.cpp makes them: (*dataset).Ax(x.data())
void bh_ax(void* h, double* x, double* out) {
    data* d = static_cast<data*>(h);
    std::vector<double> z = d->Ax(x);
    memcpy(out, z.data(), sizeof(double) * z.size());
}
void bh_atx(void* h, double* p, double* out) {
    data* d = static_cast<data*>(h);
    std::vector<double> w = d->ATx(p);
    memcpy(out, w.data(), sizeof(double) * w.size());
}
// seconds per call over `reps` calls of the class method (by-value vectors and all), after one untimed call
double bh_time_ax(void* h, double* x, int reps) {
    data* d = static_cast<data*>(h);
    (void)d->Ax(x);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) (void)d->Ax(x);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
}
double bh_time_atx(void* h, double* p, int reps) {
    data* d = static_cast<data*>(h);
    (void)d->ATx(p);
    const auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) (void)d->ATx(p);
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / reps;
}
int bh_kernel_mode(void* h) { return gv_get_kernel_mode(static_cast<data*>(h)->get_gv()); }
int bh_layout(void* h) { return gv_get_layout(static_cast<data*>(h)->get_gv()); }

}  // extern "C"
