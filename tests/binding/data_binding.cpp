// data_binding.cpp -- INTEGRATION.md section B as a translation unit: the bodies a maintainer of the reference pastes into
// data.cpp, compiled by __graft_entry__.build() and run by tests/test_gpu_binding.py against the oracle.  The lines between the
// BEGIN / END markers are the code block of INTEGRATION.md section B, character for character (tests/test_cpu_build.py checks
// that); only gv_create / gv_set_dims / gv_set_mask / gv_upload_bed / gv_marker_stats / gv_get_marker_stats / gv_ax / gv_atx /
// gv_comm_* / gv_destroy are called, and neither gv_set_kernel_mode nor gv_set_layout: the binding runs on the library's defaults.
#include "data_binding.hpp"

#include <cstdio>
#include <cstdlib>
#include <immintrin.h>

gv_allreduce_fn binding_sum = nullptr;
void* binding_sum_user = nullptr;
static const int gpus_per_node = 1;      // the boxes of this pool expose one GPU

data::data(unsigned char* bed_slab, const unsigned char* mask4_or_null, int nonas_, int N_, int M_, int Mt_, int S_, int rank_,
           int nranks_, double alpha_scale_)
    : Mt(Mt_), N(N_), M(M_), S(S_), rank(rank_), nranks(nranks_), nonas(nonas_), bed_data(bed_slab), alpha_scale(alpha_scale_) {
    mbytes = (N % 4) ? (size_t)N / 4 + 1 : (size_t)N / 4;                       // data.cpp:40
    vector_ctor = mask4_or_null == nullptr;
    if (mask4_or_null) mask4.assign(mask4_or_null, mask4_or_null + mbytes);
    mave = (double*)_mm_malloc(size_t(M > 0 ? M : 1) * sizeof(double), 64);     // data.cpp:396-397
    msig = (double*)_mm_malloc(size_t(M > 0 ? M : 1) * sizeof(double), 64);
    read_genotype_data();
    compute_markers_statistics();
}

// ---- BEGIN INTEGRATION.md section B ----
// data.cpp: end of read_genotype_data()  (data.cpp:201-234) -- hand the slab to the GPU, keep or free bed_data
void data::read_genotype_data() {
    int dev = rank % gpus_per_node;
    if (gv_abi_version() != GV_ABI_VERSION) {   // a library built from another header: structs and defaults may differ
        printf("libgvamp.so speaks ABI %d, this binding was compiled against %d\n", gv_abi_version(), GV_ABI_VERSION);
        exit(EXIT_FAILURE);
    }
    if (gv_create(dev, &gv) || gv_set_dims(gv, N, M, Mt, S) ||
        gv_set_mask(gv, vector_ctor ? NULL : mask4.data(), nonas) ||   // after read_phen(); NULL for the vector ctor
        gv_upload_bed(gv, bed_data, size_t(M) * mbytes)) { puts(gv_last_error(gv)); exit(EXIT_FAILURE); }
    if (nranks > 1) {
#ifdef GVAMP_BINDING_MPI
        // one communicator per MPI_COMM_WORLD: rank 0 makes the id, MPI ships the 128 bytes
        unsigned char id[128];
        if (rank == 0) gv_comm_unique_id(id);
        MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
        if (gv_comm_init(gv, nranks, rank, id)) { puts(gv_last_error(gv)); exit(EXIT_FAILURE); }   // RCCL over xGMI
#else
        // where RCCL is not wanted (ranks sharing a GPU, exotic fabrics) the reference's own transport carries the sums:
        //   static int mpi_sum(void*, double* buf, size_t n) {
        //       return MPI_Allreduce(MPI_IN_PLACE, buf, (int)n, MPI_DOUBLE, MPI_SUM, MPI_COMM_WORLD) != MPI_SUCCESS; }
        // (binding_sum is that function; two PCIe hops per message, same results)
        if (gv_comm_init_callback(gv, nranks, rank, binding_sum, binding_sum_user)) { puts(gv_last_error(gv)); exit(EXIT_FAILURE); }
#endif
    }
}

// data.cpp: compute_markers_statistics()  (data.cpp:392-546)
void data::compute_markers_statistics() {
    if (gv_marker_stats(gv, alpha_scale) ||
        gv_get_marker_stats(gv, mave, msig)) { puts(gv_last_error(gv)); exit(EXIT_FAILURE); }   // mave / msig stay valid for pvals_calc etc.
}

// data.cpp: Ax / ATx  (data.cpp:848, :810) -- same signatures, the MPI_Allreduce of :928/:995 happens inside
std::vector<double> data::Ax(double* __restrict__ phen) {
    std::vector<double> out(4 * mbytes);
    if (gv_ax(gv, phen, out.data())) { puts(gv_last_error(gv)); exit(EXIT_FAILURE); }
    return out;
}
std::vector<double> data::ATx(double* __restrict__ phen) {     // phen: 4*mbytes doubles, 0 at NA / pad slots
    std::vector<double> out(M);
    if (gv_atx(gv, phen, out.data())) { puts(gv_last_error(gv)); exit(EXIT_FAILURE); }
    return out;
}

data::~data() {
    if (mave != nullptr) _mm_free(mave);
    if (msig != nullptr) _mm_free(msig);
    gv_destroy(gv);
}
// ---- END INTEGRATION.md section B ----
