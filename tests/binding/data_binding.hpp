// data_binding.hpp -- the part of the reference's `class data` (data.hpp:12-57) that INTEGRATION.md section B touches: the
// members its five method bodies read, plus the ONE member the binding adds (`gv`).  Test infrastructure: a stand-in for the
// reference's class so that the bodies of section B compile and run here exactly as a maintainer would paste them (the
// reference itself cannot be built in this image: every translation unit includes Boost).  Nothing here is product code.
#pragma once
#include <cstddef>
#include <string>
#include <vector>

#include "gvamp.h"

class data {
private:
    int Mt;      // total number of markers                         (data.hpp:16)
    int N;       // number of individuals
    int M;       // number of markers attributed to the rank
    int S;       // marker starting index
    int rank;    // rank of the MPI process
    int nranks;  // MPI_Comm_size (a local of the reference's drivers)
    int nonas;   //                                                 (data.hpp:31)
    std::vector<unsigned char> mask4;
    double* mave = nullptr;
    double* msig = nullptr;
    unsigned char* bed_data = nullptr;
    size_t mbytes;
    double alpha_scale;
    bool vector_ctor;         // phenotype handed in as a vector (data.cpp:86-100): every individual present

    gv_ctx* gv = nullptr;     // <-- the member section B adds

public:
    // what the two constructors of data.cpp:30-113 leave behind before read_genotype_data(): sizes, mask4 / nonas (read_phen) and
    // the slab of the .bed this rank owns
    data(unsigned char* bed_slab, const unsigned char* mask4_or_null, int nonas, int N, int M, int Mt, int S, int rank,
         int nranks, double alpha_scale);
    ~data();
    void read_genotype_data();
    void compute_markers_statistics();
    std::vector<double> Ax(double* __restrict__ phen);
    std::vector<double> ATx(double* __restrict__ phen);
    double* get_mave() { return mave; }
    double* get_msig() { return msig; }
    size_t get_mbytes() { return mbytes; }
    gv_ctx* get_gv() { return gv; }
};

// the transport of the harness, standing in for MPI_Allreduce where the image has no MPI (tests/binding/harness.cpp)
extern gv_allreduce_fn binding_sum;
extern void* binding_sum_user;
