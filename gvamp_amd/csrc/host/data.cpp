// data.cpp -- see data.hpp.
#include "data.hpp"

#include <cassert>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>
#include <sys/stat.h>
#include <ctime>

#include <unistd.h>

#include "shm_comm.hpp"
#include "utilities.hpp"

namespace {
[[noreturn]] void die(const std::string& msg) {
    std::cout << msg << std::endl;
    exit(EXIT_FAILURE);
}
void ck(gv_ctx* ctx, int rc, const char* what) {
    if (rc) die(std::string("FATAL: ") + what + ": " + gv_last_error(ctx));
}
// whitespace split that keeps the reference's quirk: a leading blank yields an empty token 0 (std::regex
// "\\s+" with sregex_token_iterator(-1), data.cpp:143-144)
std::vector<std::string> split_ws(const std::string& line) {
    std::vector<std::string> tok;
    size_t i = 0, n = line.size();
    if (n && isspace((unsigned char)line[0])) {
        tok.push_back("");
        while (i < n && isspace((unsigned char)line[i])) i++;
    }
    while (i < n) {
        size_t j = i;
        while (j < n && !isspace((unsigned char)line[j])) j++;
        tok.push_back(line.substr(i, j - i));
        i = j;
        while (i < n && isspace((unsigned char)line[i])) i++;
    }
    return tok;
}
}  // namespace

void data::open_device(int device, int kernel_mode) {
    if (device < 0) device = gv_env_local_rank();
    if (gv_env_nranks() > 1) (void)gv_bind_host_numa(device, nullptr);   // one process per GPU: stay on the CPUs next to it
    if (gv_create(device, &ctx)) die(std::string("FATAL: ") + gv_last_error(nullptr));
    ck(ctx, gv_set_dims(ctx, N, M, Mt, S), "gv_set_dims");
    // kernel mode 1: two stripe sets (1), the single tile layout (2: half the HBM, same bits), or -- the default, 3 -- the tile
    // layout unless the run is long (>= 1000 expected ATx passes) and two stripe sets fit the free HBM (--resident-layout)
    const char* lay = getenv("GVAMP_RESIDENT_LAYOUT");
    const int stripes = kernel_mode != 0 ? ((lay && atoi(lay) >= 1 && atoi(lay) <= 3) ? atoi(lay) : 3) : 0;
    ck(ctx, gv_set_layout(ctx, kernel_mode == 0, stripes), "gv_set_layout");
    ck(ctx, gv_set_kernel_mode(ctx, kernel_mode), "gv_set_kernel_mode");
    if (const char* ep = getenv("GVAMP_EXPECTED_PASSES")) ck(ctx, gv_set_expected_passes(ctx, atoll(ep) > 0 ? atoll(ep) : 0), "gv_set_expected_passes");
    if (kernel_mode == 0 && rank == 0)
        std::cerr << "WARNING: --kernel-mode 0 selects the fp64 VALU kernels (parity anchor, 4-9 % of the HBM roofline): "
                     "expect Ax / ATx 10-20x slower than the default --kernel-mode 1" << std::endl;
    if (gv_env_nranks() > 1) {
        gv_ctx* world = gv_host_world(device);          // the process's one communicator (MPI_COMM_WORLD of the reference)
        ck(ctx, gv_comm_share(ctx, world), "gv_comm_share");
    }
}

// ---- the process's communicator ------------------------------------------------------------------------------------------
// The reference has ONE MPI_COMM_WORLD per process however many `data` objects it builds (main_real --run-mode both: a training
// and a test set).  Here a small context that holds no data owns the communicator -- RCCL over xGMI by default, the shared-
// memory host transport with GVAMP_COMM=host -- and every data object's context joins it (gv_comm_share): one rendezvous per
// process, so a rendezvous file can never be read by a later object of the same run.
namespace {
gv_ctx* g_world = nullptr;
gvh_shm_comm* g_shm = nullptr;
}
gv_ctx* gv_host_world(int device) {
    if (g_world) return g_world;
    const int rank = gv_env_rank(), nranks = gv_env_nranks();
    if (device < 0) device = gv_env_local_rank();
    if (gv_create(device, &g_world)) die(std::string("FATAL: ") + gv_last_error(nullptr));
    const char* kind = getenv("GVAMP_COMM");
    if (kind && (!strcmp(kind, "host") || !strcmp(kind, "shm"))) {
        // ranks are processes of one node meeting in shared memory (shm_comm.hpp): device -> host, sum, host -> device per message
        std::string err;
        g_shm = gvh_shm_open_impl(gvh_shm_default_name(), nranks, rank, (size_t)1 << 20, err);
        if (!g_shm) die("FATAL: " + err);
        ck(g_world, gv_comm_init_callback(g_world, nranks, rank, gvh_shm_allreduce, g_shm), "gv_comm_init_callback");
        return g_world;
    }
    if (kind && strcmp(kind, "rccl") != 0) die(std::string("FATAL: GVAMP_COMM must be rccl (default) or host, not ") + kind);
    // one process per GPU: rank 0 publishes the RCCL unique id through a file whose name is private to the job -- the launcher's
    // $GVAMP_RENDEZVOUS, else derived from the launcher's job id (gvh_job_key: torchrun / Slurm / PMIx ids first, the parent's
    // pid only as the last resort)
    const std::string path = gvh_id_file_default();
    unsigned char id[128];
    if (rank == 0) ck(nullptr, gv_comm_unique_id(id), "gv_comm_unique_id");
    {
        std::string err;
        if (gvh_exchange_id_impl(path, rank, id, 60.0, err)) die("FATAL: " + err);
    }
    ck(g_world, gv_comm_init(g_world, nranks, rank, id), "gv_comm_init");   // returns after a collective: every rank has the id
    if (rank == 0) remove(path.c_str());
    return g_world;
}
// MPI_Finalize of the drivers: after the last data / vamp object is gone
void gv_host_finalize() {
    if (g_world) gv_destroy(g_world);
    g_world = nullptr;
    if (g_shm) gvh_shm_close_impl(g_shm);
    g_shm = nullptr;
}

void data::push_mask() { ck(ctx, gv_set_mask(ctx, mask4.data(), nonas), "gv_set_mask"); }

data::data(std::vector<double> y, std::string genofp, const int N, const int M, const int Mt, const int S,
           const int rank, std::string type_data, double alpha_scale, std::string bimfp, int device, int kernel_mode)
    : bimfp(bimfp), type_data(type_data), N(N), M(M), Mt(Mt), S(S), rank(rank), phen_data(y), alpha_scale(alpha_scale) {
    if (type_data != "bed") die("FATAL: only type_data == \"bed\" is supported by this build");
    mbytes = (N % 4) ? (size_t)N / 4 + 1 : (size_t)N / 4;
    im4 = (int)mbytes;
    mask4.assign(mbytes, 0x0F);                       // data.cpp:86-89
    if (N % 4) {                                      // data.cpp:92-98
        for (int i = N % 4; i < 4; i++) mask4[N / 4] &= ~(0b1 << i);
        std::cout << "rank = " << rank << ": setting last " << 4 - N % 4 << " bits to NAs" << std::endl;
    }
    set_nonas(N);
    open_device(device, kernel_mode);
    push_mask();
    bedfp = genofp;
    read_genotype_data();
    compute_markers_statistics();
}

data::data(std::string fp, std::string genofp, const int N, const int M, const int Mt, const int S, const int rank,
           std::string type_data, double alpha_scale, std::string bimfp, int device, int kernel_mode)
    : phenfp(fp), bimfp(bimfp), type_data(type_data), N(N), M(M), Mt(Mt), S(S), rank(rank), alpha_scale(alpha_scale) {
    if (type_data != "bed") die("FATAL: only type_data == \"bed\" is supported by this build");
    mbytes = (N % 4) ? (size_t)N / 4 + 1 : (size_t)N / 4;
    im4 = (int)mbytes;
    bedfp = genofp;
    read_phen();
    open_device(device, kernel_mode);
    push_mask();
    read_genotype_data();
    compute_markers_statistics();
}

data::data(gv_ctx* resident, std::vector<double> y, const int N, const int M, const int Mt, const int S, const int rank,
           const std::vector<unsigned char>* m4, int nonas_, double alpha_scale)
    : N(N), M(M), Mt(Mt), S(S), rank(rank), phen_data(y), alpha_scale(alpha_scale), ctx(resident), owns_ctx(false) {
    mbytes = (N % 4) ? (size_t)N / 4 + 1 : (size_t)N / 4;
    im4 = (int)mbytes;
    if (m4) {
        mask4 = *m4;
        nonas = nonas_;
    } else {
        mask4.assign(mbytes, 0x0F);
        if (N % 4)
            for (int i = N % 4; i < 4; i++) mask4[N / 4] &= ~(0b1 << i);
        nonas = N;
    }
    push_mask();
    compute_markers_statistics();
}

data::~data() {
    if (ctx && owns_ctx) gv_destroy(ctx);
}

// data.cpp:128-192: 3rd token of every line; "NA" -> DBL_MAX and mask bit cleared; scaled (not centred) by
// sqrt((nonas-1) / sum (y - mean)^2) -- NA slots are scaled too and become +inf, as in the reference.
void data::read_phen() {
    std::ifstream infile(phenfp);
    if (!infile.is_open()) die("FATAL: could not open phenotype file: " + phenfp);
    std::string line;
    double sum = 0.0;
    int line_n = 0;
    nonas = nas = 0;
    mask4.clear();
    phen_data.clear();
    while (getline(infile, line)) {
        const int k = line_n % 4;
        if (k == 0) mask4.push_back(0x0F);
        std::vector<std::string> tokens = split_ws(line);
        if (tokens.size() < 3) die("FATAL: phenotype line with fewer than 3 columns in " + phenfp);
        if (tokens[2] == "NA") {
            nas++;
            phen_data.push_back(std::numeric_limits<double>::max());
            mask4[line_n / 4] &= ~(0b1 << k);
        } else {
            nonas++;
            const double v = atof(tokens[2].c_str());
            phen_data.push_back(v);
            sum += v;
        }
        line_n++;
    }
    assert(nas + nonas == N);
    if (line_n % 4) {
        for (int i = line_n % 4; i < 4; i++) mask4[line_n / 4] &= ~(0b1 << i);
        std::cout << "rank = " << rank << ": setting last " << 4 - line_n % 4 << " bits to NAs" << std::endl;
    }
    const double avg = sum / double(nonas);
    double sqn = 0.0;
    for (double v : phen_data)
        if (v != std::numeric_limits<double>::max()) sqn += (v - avg) * (v - avg);
    sqn = sqrt(double(nonas - 1) / sqn);
    for (double& v : phen_data) v *= sqn;
    intercept = avg;
    scale = sqn;
}

// data.cpp:201-234: this rank's slab of the SNP-major .bed at byte offset 3 + S*mbytes (magic bytes skipped, not
// validated), handed to the device in bounded pieces.
void data::read_genotype_data() {
    const size_t size_bytes = size_t(M) * mbytes;
    printf("INFO   : rank %d streams %zu bytes (%.3f GB) of raw data to the device.\n", rank, size_bytes, double(size_bytes) / 1.0E9);
    const auto t0 = std::chrono::steady_clock::now();
    ck(ctx, gv_upload_bed_file(ctx, bedfp.c_str(), (int64_t)(3 + size_t(S) * mbytes)), "gv_upload_bed_file");
    if (rank == 0) {                                                                 // data.cpp:227-232
        std::cout << "reading genotype data took "
                  << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " seconds." << std::endl;
        gv_ingest_stats st{};
        if (gv_ingest_info2(ctx, &st) == 0 && st.layout != 0)      // [ext] what --resident-layout 3 decided, and from what
            std::cout << "resident layout: " << (st.layout == 1 ? "two stripe sets" : "one tile layout") << ", " << st.resident_bytes / 1e9
                      << " GB (expected ATx passes " << (long long)st.expected_passes
                      << (st.expected_passes >= 1000 ? ": a long run -- two sets when they fit" : ": not announced as a long run -- one layout")
                      << "; --resident-layout / GVAMP_EXPECTED_PASSES override)" << std::endl;
    }
}

std::vector<int> data::read_chromosome_info(std::string bim_file) {
    std::vector<int> chroms;
    std::ifstream infile(bim_file);
    if (!infile.is_open()) die("FATAL: could not open bim file: " + bim_file);
    std::string line;
    for (int line_n = 0; getline(infile, line); line_n++) {
        if (line_n < S || line_n >= S + M) continue;
        std::vector<std::string> tokens = split_ws(line);
        if (tokens.empty()) die("FATAL: empty line " + std::to_string(line_n + 1) + " in bim file " + bim_file);
        chroms.push_back(tokens[0] == "X" ? 23 : (int)atof(tokens[0].c_str()));
    }
    return chroms;
}

void data::compute_markers_statistics() {
    const auto t0 = std::chrono::steady_clock::now();
    ck(ctx, gv_marker_stats(ctx, alpha_scale), "gv_marker_stats");
    mave.assign(M > 0 ? M : 1, 0.0);
    msig.assign(M > 0 ? M : 1, 0.0);
    ck(ctx, gv_get_marker_stats(ctx, mave.data(), msig.data()), "gv_get_marker_stats");
    if (rank == 0 && !gv_host_quiet())                                                // data.cpp:543-545
        std::cout << "rank = " << rank << ": statistics took "
                  << std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() << " seconds to run."
                  << std::endl;
}

std::vector<double> data::Ax(double* __restrict__ phen) {
    std::vector<double> out(4 * mbytes, 0.0);
    ck(ctx, gv_ax(ctx, phen, out.data()), "gv_ax");
    return out;
}

std::vector<double> data::ATx(double* __restrict__ phen) {
    std::vector<double> out(M > 0 ? M : 0, 0.0);
    ck(ctx, gv_atx(ctx, phen, out.data()), "gv_atx");
    return out;
}

std::vector<double> data::pvals_calc_dev(gv_vec* z1, gv_vec* y, gv_vec* x1_hat, bool loco, const std::string& pred_prefix) {
    std::vector<double> pv(M > 0 ? M : 1, 0.0);
    if (loco) {
        std::vector<int> ch_info = read_chromosome_info(bimfp);
        ch_info.resize(M > 0 ? M : 1, 0);
        if (pred_prefix.empty())
            ck(ctx, gv_pvals_loco(ctx, z1, y, x1_hat, ch_info.data(), pv.data()), "gv_pvals_loco");
        else {
            // the per-chromosome predictors of data.cpp:1276-1281: <prefix>_LOCO_chr_<ch>.csv, 4*mbytes values, rank 0
            std::vector<double> pred((size_t)23 * 4 * mbytes, 0.0);
            ck(ctx, gv_pvals_loco_pred(ctx, z1, y, x1_hat, ch_info.data(), pv.data(), pred.data()), "gv_pvals_loco_pred");
            for (int ch = 1; ch <= 23 && rank == 0; ch++) {
                const std::string fp = pred_prefix + "_LOCO_chr_" + std::to_string(ch) + ".csv";
                store_vec_to_file(fp, std::vector<double>(pred.begin() + (size_t)(ch - 1) * 4 * mbytes, pred.begin() + (size_t)ch * 4 * mbytes));
                if (!gv_host_quiet()) std::cout << "filepath predictors = " << fp << std::endl;
            }
        }
    } else
        ck(ctx, gv_pvals_loo(ctx, z1, y, x1_hat, pv.data()), "gv_pvals_loo");
    pv.resize(M > 0 ? M : 0);
    return pv;
}

static std::vector<std::vector<double>> pvals_host(data* d, gv_ctx* ctx, int M, size_t mbytes,
                                                   std::vector<std::vector<double>>& z1, std::vector<double>& y,
                                                   std::vector<std::vector<double>>& x1_hat, bool loco,
                                                   const std::vector<std::string>& prefixes = std::vector<std::string>()) {
    std::vector<std::vector<double>> out;
    gv_vec *dz = nullptr, *dy = nullptr, *dx = nullptr;
    ck(ctx, gv_vec_alloc(ctx, GV_SPACE_N, &dz), "gv_vec_alloc");
    ck(ctx, gv_vec_alloc(ctx, GV_SPACE_N, &dy), "gv_vec_alloc");
    ck(ctx, gv_vec_alloc(ctx, GV_SPACE_M, &dx), "gv_vec_alloc");
    std::vector<double> yp(y);
    yp.resize(4 * mbytes, 0.0);
    ck(ctx, gv_vec_upload(ctx, dy, yp.data()), "gv_vec_upload");
    for (size_t ie = 0; ie < z1.size(); ie++) {
        std::vector<double> zp(z1[ie]);
        zp.resize(4 * mbytes, 0.0);
        ck(ctx, gv_vec_upload(ctx, dz, zp.data()), "gv_vec_upload");
        std::vector<double> xp(x1_hat[ie]);
        xp.resize(M > 0 ? M : 1, 0.0);
        ck(ctx, gv_vec_upload(ctx, dx, xp.data()), "gv_vec_upload");
        out.push_back(d->pvals_calc_dev(dz, dy, dx, loco, (loco && ie < prefixes.size()) ? prefixes[ie] : std::string()));
    }
    gv_vec_free(ctx, dz);
    gv_vec_free(ctx, dy);
    gv_vec_free(ctx, dx);
    return out;
}

std::vector<std::vector<double>> data::pvals_calc(std::vector<std::vector<double>> z1, std::vector<double> y,
                                                  std::vector<std::vector<double>> x1_hat,
                                                  std::vector<std::string> filepath) {
    std::vector<std::vector<double>> pv = pvals_host(this, ctx, M, mbytes, z1, y, x1_hat, false);
    for (size_t ie = 0; ie < pv.size() && ie < filepath.size(); ie++) mpi_store_vec_to_file(filepath[ie], pv[ie], S, M);
    return pv;
}

std::vector<std::vector<double>> data::pvals_calc_LOCO(std::vector<std::vector<double>> z1, std::vector<double> y,
                                                       std::vector<std::vector<double>> x1_hat,
                                                       std::vector<std::string> filepath) {
    std::vector<std::vector<double>> pv = pvals_host(this, ctx, M, mbytes, z1, y, x1_hat, true, filepath);
    for (size_t ie = 0; ie < pv.size() && ie < filepath.size(); ie++)
        mpi_store_vec_to_file(filepath[ie] + "_pvals_LOCO.bin", pv[ie], S, M);          // data.cpp:1347-1350
    return pv;
}

// length 4*mbytes (the reference returns N entries and lets ATx read past the end when N % 4 != 0)
std::vector<double> data::filter_pheno() {
    std::vector<double> y(4 * mbytes, 0.0);
    for (int n = 0; n < N; n++)
        if ((mask4[n / 4] >> (n % 4)) & 1) y[n] = phen_data[n];
    return y;
}

std::vector<double> data::filter_pheno(int* nonnan) {
    std::vector<double> y = filter_pheno();
    int cnt = 0;
    for (int n = 0; n < N; n++) cnt += (mask4[n / 4] >> (n % 4)) & 1;
    *nonnan = cnt;
    return y;
}

// data.cpp:286-331 -- covariates file of the probit model: one line per individual, C whitespace-separated numbers.
// (The reference splits on the regex \s+ and dies in std::stod on a leading blank; a stream extraction reads the same
// well-formed files.)  A line with another number of values is fatal, as in the reference.
void data::read_covariates(std::string covfp, int C) {
    if (C == 0) return;
    std::ifstream covf(covfp);
    if (!covf.is_open()) {
        std::cout << "FATAL: cannot open covariates file " << covfp << std::endl;
        exit(EXIT_FAILURE);
    }
    std::string line;
    while (std::getline(covf, line)) {
        std::istringstream is(line);
        std::vector<double> entries;
        std::string tok;
        while (is >> tok) entries.push_back(std::stod(tok));
        if ((int)entries.size() != C) {
            std::cout << "FATAL: number of covariates = " << entries.size()
                      << " does not match to the specified number of covariates = " << C << std::endl;
            exit(EXIT_FAILURE);
        }
        covs.push_back(entries);
    }
    if ((int)covs.size() < N) {
        std::cout << "FATAL: covariates file " << covfp << " holds " << covs.size() << " lines for N = " << N << std::endl;
        exit(EXIT_FAILURE);
    }
}

// data.cpp:1050-1058
std::vector<double> data::Zx(std::vector<double> phen) {
    std::vector<double> out(4 * mbytes, 0.0);
    for (int i = 0; i < N; i++) out[i] = inner_prod(covs[i], phen, 0);
    return out;
}
