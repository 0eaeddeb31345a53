// shm_comm.hpp -- a host transport for the sums the reference does with MPI_Allreduce (data.cpp:928/:995, utilities.cpp:203):
// the ranks of ONE node are processes that meet in a POSIX shared-memory segment.  It plugs into libgvamp through
// gv_comm_init_callback (include/gvamp.h) -- device -> host, sum here, host -> device -- and exists for the cases RCCL does not
// cover: ranks sharing one GPU (how the sharded drivers are exercised on a 1-GPU box), and hosts without a working fabric.
// Sums are taken in rank order by every rank, so all ranks hold bit-identical results.  `GVAMP_COMM=host` selects it in the
// drivers (host/data.cpp); production multi-GPU jobs use RCCL.
#pragma once
#include <cstddef>
#include <string>

struct gvh_shm_comm;   // opaque

// Ranks 0..nranks-1 that name the same segment form a communicator.  cap_doubles: largest piece summed at once (longer
// messages go in pieces).  Rank 0 creates the segment and unlinks its name once everybody has attached.  Returns nullptr and
// fills err on failure (120 s rendezvous timeout).
gvh_shm_comm* gvh_shm_open_impl(const std::string& name, int nranks, int rank, size_t cap_doubles, std::string& err);
// in-place SUM of n doubles over the ranks; 0 = ok.  Signature of gv_allreduce_fn (user = the communicator).
extern "C" int gvh_shm_allreduce(void* comm, double* buf, size_t n);
void gvh_shm_close_impl(gvh_shm_comm* c);
// What the ranks of one job agree on without talking: $GVAMP_RENDEZVOUS when set (launchers hand every job a fresh one), else the
// job id of the launcher (TORCHELASTIC_RUN_ID, SLURM_JOB_ID + SLURM_STEP_ID, PMIX_NAMESPACE, PMI_JOBID, OMPI jobid) with
// MASTER_ADDR:MASTER_PORT, and only as the last resort MASTER_PORT + the parent's pid (ranks must then be direct children of one
// launcher).  The RCCL id file of host/data.cpp and the segment name below are both derived from it.
std::string gvh_job_key();
std::string gvh_shm_default_name();
// The 128-byte RCCL unique id from rank 0 to the other ranks through a file: rank 0 writes id128 (atomically: tmp + rename), the
// others poll for a file of this job (written after they started) and read it into id128.  0 = ok; err names the path and the key.
std::string gvh_id_file_default();
int gvh_exchange_id_impl(const std::string& path, int rank, unsigned char* id128, double timeout_s, std::string& err);
