"""The reference-side binding of INTEGRATION.md section B, compiled (tests/binding/, built by __graft_entry__.build()) and run:
a `data`-shaped class whose read_genotype_data / compute_markers_statistics / Ax / ATx bodies are the section's code block and
call nothing but gv_create, gv_set_dims, gv_set_mask, gv_upload_bed, gv_marker_stats, gv_get_marker_stats, gv_ax, gv_atx
(include/gvamp.h) -- neither gv_set_kernel_mode nor gv_set_layout.  What it gets must be the engine bench.py measures: kernel
mode 1 on a re-encoded layout, the oracle's results, and the streaming rate through the host-pointer entry points
(data.hpp:117-121: std::vector<double> Ax(double*), ATx(double*))."""
import ctypes as C
import os

import numpy as np
import pytest

from gvamp_amd import capi, synth

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "binding", "libgvbinding.so")


def rel(a, b):
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)


def load():
    if not os.path.exists(LIB):
        raise RuntimeError("tests/binding/libgvbinding.so is not built: python -c 'import __graft_entry__ as g; g.build()'")
    L = C.CDLL(LIB)
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_ubyte)
    L.bh_create.restype = C.c_void_p
    L.bh_create.argtypes = [up, up, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double]
    L.bh_destroy.argtypes = [C.c_void_p]
    L.bh_destroy.restype = None
    L.bh_stats.argtypes = [C.c_void_p, C.c_int, dp, dp]
    L.bh_ax.argtypes = [C.c_void_p, dp, dp]
    L.bh_atx.argtypes = [C.c_void_p, dp, dp]
    for f in (L.bh_time_ax, L.bh_time_atx):
        f.argtypes = [C.c_void_p, dp, C.c_int]
        f.restype = C.c_double
    L.bh_kernel_mode.argtypes = [C.c_void_p]
    L.bh_layout.argtypes = [C.c_void_p]
    return L


class Bound:
    """the reference's `data` object behind the harness: construct = ctor + read_genotype_data + compute_markers_statistics"""

    def __init__(self, bed, N, M, mask4=None, nonas=None, Mt=None, S=0, rank=0, nranks=1, alpha_scale=1.0):
        self.L = load()
        self.N, self.M = N, M
        self.mb = (N + 3) // 4
        self.bed = np.ascontiguousarray(bed, dtype=np.uint8)          # bed_data: owned by the caller, as in the reference
        assert self.bed.size == M * self.mb
        up = C.POINTER(C.c_ubyte)
        self.m4 = np.ascontiguousarray(mask4, dtype=np.uint8) if mask4 is not None else None
        self.h = self.L.bh_create(self.bed.ctypes.data_as(up), self.m4.ctypes.data_as(up) if self.m4 is not None else None,
                                  N if nonas is None else nonas, N, M, M if Mt is None else Mt, S, rank, nranks, alpha_scale)
        assert self.h

    def close(self):
        if self.h:
            self.L.bh_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def stats(self):
        a, b = np.empty(self.M), np.empty(self.M)
        self.L.bh_stats(self.h, self.M, capi._dp(a), capi._dp(b))
        return a, b

    def Ax(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(4 * self.mb)
        self.L.bh_ax(self.h, capi._dp(x), capi._dp(out))
        return out

    def ATx(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert p.size == 4 * self.mb
        out = np.empty(self.M)
        self.L.bh_atx(self.h, capi._dp(p), capi._dp(out))
        return out

    def engine(self):
        return self.L.bh_kernel_mode(self.h), self.L.bh_layout(self.h)


def alg_bytes(N, M):
    mb = (N + 3) // 4
    return M * mb + 24 * M + 32 * mb


@pytest.mark.parametrize("N,M,miss,fna", [(2000, 300, 10000, 0.0), (1003, 129, 20000, 0.02), (4100, 2049, 5000, 0.0)])
def test_section_b_binding_vs_oracle(oracle, N, M, miss, fna):
    rng = np.random.default_rng(N + M)
    bed = synth.synth_bed(N, M, seed=99, miss_ppm=miss)
    m4 = nonas = None
    present = np.ones(N, bool)
    if fna > 0 or N % 4:                       # the phenotype-file constructor: read_phen() left mask4 / nonas behind
        present = rng.random(N) >= fna
        m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
        for n in np.nonzero(present)[0]:
            m4[n >> 2] |= 1 << (n & 3)
        nonas = int(present.sum())
    with Bound(bed, N, M, mask4=m4, nonas=nonas) as d:
        mode, layout = d.engine()
        assert mode == 1 and layout in (1, 2), (mode, layout)       # the measured engine, without having asked for it
        mave, msig = d.stats()
        o_mave, o_msig = oracle.marker_stats(bed, N, M, mask4=m4, nonas=N if nonas is None else nonas)
        assert np.allclose(mave, o_mave, rtol=1e-13, atol=1e-15) and np.allclose(msig, o_msig, rtol=1e-12, atol=0)
        x = rng.standard_normal(M)
        z = d.Ax(x)
        oz = oracle.ax(bed, N, M, o_mave, o_msig, x, mask4=m4)
        assert z.shape == oz.shape and rel(z, oz) < 1e-12
        assert np.all(z[N:] == 0) and np.all(z[:N][~present] == 0)
        p = np.zeros(4 * ((N + 3) // 4))
        p[:N] = rng.standard_normal(N) * present
        assert rel(d.ATx(p), oracle.atx(bed, N, M, o_mave, o_msig, p)) < 1e-12


def _device_bed(N, M, seed, S0=0, Mt=None, chunk=125000):
    """M * ceil(N/4) bytes of the seeded synthetic .bed (SURVEY 8d recipe) as a HOST array, generated on the device chunk by chunk
    (the generator is a function of the global marker index) and read back through the raw-row layout"""
    mb = (N + 3) // 4
    out = np.empty(M * mb, dtype=np.uint8)
    for m0 in range(0, M, chunk):
        mc = min(chunk, M - m0)
        with capi.Shard(N, mc, Mt=(M if Mt is None else Mt), S=S0 + m0) as sh:
            sh.set_layout(True, 0)             # raw rows only: the bytes are all that is wanted
            sh.synth_bed(seed, 5000)
            out[m0 * mb:(m0 + mc) * mb] = sh.download_bed()
    return out


def _check_big(oracle, d, bed, N, M, rng, nsub=1500):
    """oracle checks that stay cheap at any size: marker statistics and ATx restricted to a block of markers, Ax of a vector
    supported on that block (the columns are independent), and the adjoint identity over the whole shard"""
    mb = (N + 3) // 4
    m0 = int(rng.integers(0, M - nsub))
    sub = bed[m0 * mb:(m0 + nsub) * mb]
    mave, msig = d.stats()
    o_mave, o_msig = oracle.marker_stats(sub, N, nsub)
    # (the oracle adds N squared deviations in fp64, the device works from exact genotype counts: 1e-10 at N = 400k)
    assert np.allclose(mave[m0:m0 + nsub], o_mave, rtol=1e-12, atol=1e-15)
    assert np.allclose(msig[m0:m0 + nsub], o_msig, rtol=1e-10, atol=0)
    x = np.zeros(M)
    x[m0:m0 + nsub] = rng.standard_normal(nsub)
    z = d.Ax(x)
    assert rel(z, oracle.ax(sub, N, nsub, o_mave, o_msig, x[m0:m0 + nsub])) < 1e-11     # fp64 sums of the oracle at N = 400k: 2e-12
    p = np.zeros(4 * mb)
    p[:N] = rng.standard_normal(N)
    w = d.ATx(p)
    assert rel(w[m0:m0 + nsub], oracle.atx(sub, N, nsub, o_mave, o_msig, p)) < 1e-11
    xf = rng.standard_normal(M)
    lhs, rhs = float(d.Ax(xf) @ p), float(xf @ w)
    assert abs(lhs - rhs) < 1e-10 * max(abs(lhs), abs(rhs), 1.0)


def test_section_b_binding_at_config2_size_runs_the_measured_engine(oracle):
    """BASELINE config 2 (N=100k x M=500k, 12.5 GB of genotypes) through the binding: correct against the oracle, kernel mode 1 on
    a re-encoded layout, and a streaming rate through std::vector<double> Ax(double*) / ATx(double*) -- M- or N-vector over PCIe
    each way, by-value result -- of 5.6-5.9 TB/s of algorithmic bytes, asserted at 5.2 (bench.py reports the same number at the
    headline size as hostptr_GBps)."""
    N, M = 100000, 500000
    rng = np.random.default_rng(5)
    bed = _device_bed(N, M, seed=20240)
    with Bound(bed, N, M) as d:
        assert d.engine()[0] == 1 and d.engine()[1] in (1, 2)
        _check_big(oracle, d, bed, N, M, rng)
        x = rng.standard_normal(M)
        p = np.zeros(4 * ((N + 3) // 4))
        p[:N] = rng.standard_normal(N)
        t_ax, t_atx = _best_of(d, x, p, 20, 2 * alg_bytes(N, M) / 5600e9)
        rate = 2 * alg_bytes(N, M) / (t_ax + t_atx) / 1e9
        print("config-2 binding: Ax %.3f ms, ATx %.3f ms per call through the class -> %.0f GB/s" % (t_ax * 1e3, t_atx * 1e3, rate))
        _report("binding_config2", {"N": N, "M": M, "ax_ms": t_ax * 1e3, "atx_ms": t_atx * 1e3, "GBps": rate, "layout": d.engine()[1]})
        # 5.6-5.9 TB/s on the boxes of round 3 (profiles/README.md); the bar leaves the few per cent by which boxes and hosts of
        # the pool differ -- what it guards against is the wrong engine (the fp64 family streams at 0.3-0.7 TB/s)
        assert rate >= 5200.0, (t_ax, t_atx, rate)


def _best_of(d, x, p, reps, good_enough_s, budget_s=12.0):
    """seconds per call of Ax and of ATx through the class, best batch of `reps` calls: a box that is still wiping the 100+ GB the
    previous test freed (a background job of the driver that shares the HBM: 30-60 GB/s, i.e. several seconds) measures that
    wipe, not the product -- batches go on until a pair is as fast as the kernels allow, or for `budget_s` seconds"""
    import time
    best = (1e9, 1e9)
    t0 = time.time()
    while True:
        t = (d.L.bh_time_ax(d.h, capi._dp(x), reps), d.L.bh_time_atx(d.h, capi._dp(p), reps))
        if sum(t) < sum(best):
            best = t
        if sum(best) <= good_enough_s or time.time() - t0 > budget_s:
            return best


def _report(name, obj):
    """measured rates of the size tests, kept when the run has a gpurun_out/ to leave them in (profiles/ quotes them)"""
    import json
    out = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, name + ".json"), "w") as f:
            json.dump(obj, f)


def _mem_available_gb():
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def test_section_b_binding_holds_the_headline_shard(oracle):
    """N=400k x Mt=1M on ONE GPU through the unchanged binding: 100 GB of .bed handed over by gv_upload_bed from host memory.
    With the round-2 defaults (raw rows + two stripe sets = 300 GB) this ingest failed on a 288 GB part."""
    N, M = 400000, 1000000
    need = M * ((N + 3) // 4) / 1e9 + 20
    if _mem_available_gb() < need:
        pytest.skip("needs %.0f GB of host memory for the .bed slab a rank of the reference holds (have %.0f)" % (need, _mem_available_gb()))
    rng = np.random.default_rng(6)
    bed = _device_bed(N, M, seed=777)
    with Bound(bed, N, M) as d:
        assert d.engine()[0] == 1 and d.engine()[1] in (1, 2)
        _check_big(oracle, d, bed, N, M, rng, nsub=500)
        x = rng.standard_normal(M)
        p = np.zeros(4 * ((N + 3) // 4))
        t_ax, _ = _best_of(d, x, p, 5, 2 * alg_bytes(N, M) / 5800e9)
        print("headline binding: Ax %.2f ms per call -> %.0f GB/s" % (t_ax * 1e3, alg_bytes(N, M) / t_ax / 1e9))
        _report("binding_headline", {"N": N, "M": M, "ax_ms": t_ax * 1e3, "GBps": alg_bytes(N, M) / t_ax / 1e9, "layout": d.engine()[1]})
        assert alg_bytes(N, M) / t_ax / 1e9 >= 5500.0


TWO_RANK_WORKER = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import test_gpu_binding as tb
from gvamp_amd import hostapi, synth
rank, world, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
H = hostapi.load()
H.gvh_shm_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
comm = C.c_void_p()
assert H.gvh_shm_open(sys.argv[4].encode(), world, rank, 1 << 16, C.byref(comm)) == 0, H.gvh_last_error()
L = tb.load()
L.bh_set_transport.argtypes = [C.c_void_p, C.c_void_p]
L.bh_set_transport(C.cast(H.gvh_shm_allreduce, C.c_void_p), comm)      # the "MPI_Allreduce" of this job
N, Mt = 3001, 2500
bed = synth.synth_bed(N, Mt, seed=31, miss_ppm=8000)
mb = (N + 3) // 4
size, modu = divmod(Mt, world)
M = size + 1 if rank < modu else size
S = sum(size + 1 if r < modu else size for r in range(rank))
present = np.random.default_rng(4).random(N) >= 0.01
m4 = np.zeros(mb, dtype=np.uint8)
for n in np.nonzero(present)[0]:
    m4[n >> 2] |= 1 << (n & 3)
x = np.random.default_rng(5).standard_normal(Mt)
p = np.zeros(4 * mb)
p[:N] = np.random.default_rng(6).standard_normal(N) * present
with tb.Bound(bed[S * mb:(S + M) * mb], N, M, mask4=m4, nonas=int(present.sum()), Mt=Mt, S=S, rank=rank, nranks=world) as d:
    mode, layout = d.engine()
    z = d.Ax(x[S:S + M])              # summed over the ranks inside (data.cpp:928/:995)
    w = d.ATx(p)
    mave, msig = d.stats()
np.savez(out, S=S, M=M, z=z, w=w, mave=mave, msig=msig, mode=mode, layout=layout)
"""


def test_section_b_binding_as_two_ranks_with_the_callers_transport(tmp_path, oracle):
    """nranks > 1 through the binding's `gv_comm_init_callback(gv, nranks, rank, mpi_sum, ...)` branch: two processes sharing
    GPU 0, the caller's transport -- here the shared-memory sum of libgvamp_host, where the reference would hand in its
    MPI_Allreduce -- carries the N-vector of every Ax."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    name = "/gvamp_bind_%d" % os.getpid()
    procs = [subprocess.Popen([sys.executable, "-c", TWO_RANK_WORKER % {"root": root}, str(r), "2", str(tmp_path / ("r%d.npz" % r)), name],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    res = [np.load(tmp_path / ("r%d.npz" % r)) for r in range(2)]
    N, Mt = 3001, 2500
    bed = synth.synth_bed(N, Mt, seed=31, miss_ppm=8000)
    present = np.random.default_rng(4).random(N) >= 0.01
    m4 = np.zeros((N + 3) // 4, dtype=np.uint8)
    for n in np.nonzero(present)[0]:
        m4[n >> 2] |= 1 << (n & 3)
    o_mave, o_msig = oracle.marker_stats(bed, N, Mt, mask4=m4, nonas=int(present.sum()))
    x = np.random.default_rng(5).standard_normal(Mt)
    p = np.zeros(4 * ((N + 3) // 4))
    p[:N] = np.random.default_rng(6).standard_normal(N) * present
    assert all(int(r["mode"]) == 1 and int(r["layout"]) in (1, 2) for r in res)
    assert np.array_equal(res[0]["z"], res[1]["z"])                               # replicated, bit for bit
    assert rel(res[0]["z"], oracle.ax(bed, N, Mt, o_mave, o_msig, x, mask4=m4)) < 1e-12
    assert rel(np.concatenate([r["w"] for r in res]), oracle.atx(bed, N, Mt, o_mave, o_msig, p)) < 1e-12
    assert np.allclose(np.concatenate([r["msig"] for r in res]), o_msig, rtol=1e-12)
