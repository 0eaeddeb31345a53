"""ctypes binding of libgvamp.so (include/gvamp.h) -- what tests/ and bench.py drive.

This module is plumbing only: every method forwards to one C-ABI entry point.  It never computes on the
CPU and never imports oracle/.  If the library or a GPU is missing it raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgvamp.so")
_LIB = None

SPACE_M, SPACE_N = 0, 1
ABI_VERSION = 4          # GV_ABI_VERSION of include/gvamp.h this module's ctypes structs were written against

EXPORTS = [
    "gv_abi_version", "gv_create", "gv_destroy", "gv_last_error", "gv_synchronize", "gv_set_dims", "gv_mbytes",
    "gv_upload_bed", "gv_upload_bed_file", "gv_synth_bed", "gv_synth_bed_ld", "gv_download_bed", "gv_set_mask", "gv_marker_stats", "gv_get_marker_stats",
    "gv_ax", "gv_atx", "gv_set_layout", "gv_get_layout", "gv_set_kernel_mode", "gv_get_kernel_mode", "gv_vec_alloc", "gv_vec_free", "gv_vec_len",
    "gv_vec_upload", "gv_vec_download", "gv_vec_fill", "gv_vec_copy", "gv_vec_axpby", "gv_vec_mul", "gv_vec_dot", "gv_vec_dots", "gv_vec_dots_ex",
    "gv_ax_dev", "gv_atx_dev", "gv_ax2_dev", "gv_atx2_dev", "gv_set_phen", "gv_lmmse_mult", "gv_cg_solve", "gv_cg_solve2", "gv_cg_solve2x",
    "gv_denoise", "gv_prior_estep", "gv_denoise_global", "gv_prior_estep_global",
    "gv_probit_denoise", "gv_probit_denoise_cov", "gv_people_stats", "gv_cg_solve_aat", "gv_cg_solve_aat2", "gv_cg_solve_aat2w", "gv_cg_solve2w", "gv_pvals_loo", "gv_pvals_loco", "gv_pvals_loco_pred", "gv_allreduce_host", "gv_comm_unique_id", "gv_comm_init", "gv_comm_init_local", "gv_comm_init_callback", "gv_comm_share", "gv_set_overlap", "gv_debug_force_multi", "gv_comm_rank", "gv_comm_size", "gv_bind_host_numa", "gv_set_timing",
    "gv_get_counters", "gv_reset_counters", "gv_get_decomp", "gv_set_decomp", "gv_tune_info", "gv_ingest_info", "gv_ingest_info2", "gv_set_expected_passes", "gv_copy_bandwidth", "gv_read_bandwidth",
]


class GvError(RuntimeError):
    pass


class DotSpec(C.Structure):            # gv_dot_spec (include/gvamp.h)
    _fields_ = [("xa", C.c_void_p), ("xb", C.c_void_p), ("ya", C.c_void_p), ("yb", C.c_void_p), ("sync", C.c_int)]


class CgStats(C.Structure):
    _fields_ = [("iters", C.c_int), ("converged", C.c_int), ("rel_res", C.c_double), ("onsager", C.c_double),
                ("n_ax", C.c_int), ("n_atx", C.c_int), ("n_relres", C.c_int)]


class CgExtras(C.Structure):
    _fields_ = [("ride_x", C.c_void_p), ("ride_out", C.c_void_p), ("a_mu_a", C.c_void_p), ("ata_mu_b", C.c_void_p)]


class CgWarm(C.Structure):
    _fields_ = [("ata_mu_start_a", C.c_void_p), ("a_mu_start_a", C.c_void_p), ("ata_mu_a", C.c_void_p),
                ("ata_v_b", C.c_void_p), ("have_ata_v_b", C.c_int)]


class AatWarm(C.Structure):
    _fields_ = [("aat_mu_start_a", C.c_void_p), ("at_mu_start_a", C.c_void_p), ("accumulate_at_mu_a", C.c_int),
                ("ata_v_b", C.c_void_p), ("have_ata_v_b", C.c_int), ("pre_x", C.c_void_p), ("pre_out", C.c_void_p),
                ("ride_x", C.c_void_p), ("ride_out", C.c_void_p), ("pre_scale", C.c_double)]


class Counters(C.Structure):
    _fields_ = [("n_ax", C.c_int64), ("n_atx", C.c_int64), ("ms_ax", C.c_double), ("ms_atx", C.c_double),
                ("ms_allreduce", C.c_double), ("n_ax_kernel", C.c_int64), ("n_atx_kernel", C.c_int64),
                ("ms_ax_kernel", C.c_double), ("ms_atx_kernel", C.c_double), ("n_ax_pass", C.c_int64),
                ("n_atx_pass", C.c_int64), ("n_allreduce", C.c_int64)]


class IngestStats(C.Structure):      # gv_ingest_stats
    _fields_ = [("alloc_seconds", C.c_double), ("fill_seconds", C.c_double), ("overlap_seconds", C.c_double),
                ("resident_bytes", C.c_double), ("layout", C.c_int), ("expected_passes", C.c_int64)]


class DecompInfo(C.Structure):
    _fields_ = [("ks", C.c_int), ("balanced_cells", C.c_int64), ("prio", C.c_int), ("taper", C.c_float), ("tuned", C.c_int),
                ("whole_quads", C.c_int64), ("geo", C.c_float), ("wgs_per_cu", C.c_int), ("xcd_skew", C.c_float)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t)


def bind_host_numa(device):
    """gv_bind_host_numa: this process onto the CPUs next to its GPU; returns the NUMA node or -1 (nothing changed)"""
    L = load()
    node = C.c_int(-1)
    if L.gv_bind_host_numa(int(device), C.byref(node)):
        raise GvError(L.gv_last_error(None).decode())
    return node.value


def load():
    """dlopen libgvamp.so.  Raises if it has not been built (python gvamp_amd/build.py)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise GvError("libgvamp.so is not built: run `python gvamp_amd/build.py` (needs hipcc)")
    L = C.CDLL(LIB_PATH)
    if L.gv_abi_version() != ABI_VERSION:
        raise GvError("libgvamp.so speaks ABI %d, this binding %d: rebuild (python gvamp_amd/build.py --force)" % (L.gv_abi_version(), ABI_VERSION))
    vp, dp, up = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_ubyte)
    i64 = C.c_int64
    L.gv_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.gv_destroy.argtypes = [vp]
    L.gv_destroy.restype = None
    L.gv_last_error.argtypes = [vp]
    L.gv_last_error.restype = C.c_char_p
    L.gv_synchronize.argtypes = [vp]
    L.gv_set_dims.argtypes = [vp, i64, i64, i64, i64]
    L.gv_mbytes.argtypes = [vp]
    L.gv_mbytes.restype = i64
    L.gv_upload_bed.argtypes = [vp, up, C.c_size_t]
    L.gv_synth_bed.argtypes = [vp, C.c_uint64, C.c_uint32]
    L.gv_synth_bed_ld.argtypes = [vp, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.gv_upload_bed_file.argtypes = [vp, C.c_char_p, i64]
    L.gv_download_bed.argtypes = [vp, up, C.c_size_t]
    L.gv_set_mask.argtypes = [vp, up, i64]
    L.gv_marker_stats.argtypes = [vp, C.c_double]
    L.gv_get_marker_stats.argtypes = [vp, dp, dp]
    L.gv_ax.argtypes = [vp, dp, dp]
    L.gv_atx.argtypes = [vp, dp, dp]
    L.gv_set_kernel_mode.argtypes = [vp, C.c_int]
    L.gv_set_layout.argtypes = [vp, C.c_int, C.c_int]
    L.gv_get_layout.argtypes = [vp]
    L.gv_get_kernel_mode.argtypes = [vp]
    L.gv_vec_alloc.argtypes = [vp, C.c_int, C.POINTER(vp)]
    L.gv_vec_free.argtypes = [vp, vp]
    L.gv_vec_free.restype = None
    L.gv_vec_len.argtypes = [vp]
    L.gv_vec_len.restype = i64
    L.gv_vec_upload.argtypes = [vp, vp, dp]
    L.gv_vec_download.argtypes = [vp, vp, dp]
    L.gv_vec_fill.argtypes = [vp, vp, C.c_double]
    L.gv_vec_copy.argtypes = [vp, vp, vp]
    L.gv_vec_axpby.argtypes = [vp, vp, C.c_double, vp, C.c_double, vp]
    L.gv_vec_mul.argtypes = [vp, vp, vp, vp]
    L.gv_vec_dot.argtypes = [vp, vp, vp, C.c_int, dp]
    L.gv_vec_dots.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), C.c_int, dp]
    L.gv_vec_dots_ex.argtypes = [vp, C.c_int, C.POINTER(DotSpec), dp]
    L.gv_ax_dev.argtypes = [vp, vp, vp]
    L.gv_atx_dev.argtypes = [vp, vp, vp]
    L.gv_ax2_dev.argtypes = [vp, vp, vp, vp, vp]
    L.gv_atx2_dev.argtypes = [vp, vp, vp, vp, vp]
    L.gv_cg_solve2.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp, C.POINTER(CgStats),
                               C.POINTER(CgStats), dp, dp]
    L.gv_cg_solve2x.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp, C.POINTER(CgStats),
                                C.POINTER(CgStats), dp, dp, C.POINTER(CgExtras)]
    L.gv_cg_solve2w.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp, C.POINTER(CgStats),
                                C.POINTER(CgStats), dp, dp, C.POINTER(CgExtras), C.POINTER(CgWarm)]
    L.gv_set_phen.argtypes = [vp, vp, dp]
    L.gv_lmmse_mult.argtypes = [vp, vp, C.c_double, C.c_double, vp]
    L.gv_cg_solve.argtypes = [vp, vp, vp, C.c_double, C.c_double, C.c_int, C.c_int, vp, C.POINTER(CgStats), dp]
    L.gv_denoise.argtypes = [vp, vp, C.c_double, dp, dp, C.c_int, vp, vp, dp]
    L.gv_prior_estep.argtypes = [vp, vp, C.c_double, C.c_double, dp, dp, C.c_int, dp]
    L.gv_allreduce_host.argtypes = [vp, dp, C.c_int]
    L.gv_probit_denoise.argtypes = [vp, vp, vp, C.c_double, C.c_double, vp, dp]
    L.gv_probit_denoise_cov.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, vp, dp]
    L.gv_people_stats.argtypes = [vp, dp, dp, dp]
    L.gv_cg_solve_aat.argtypes = [vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, C.POINTER(CgStats), dp]
    L.gv_cg_solve_aat2.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp, vp, C.POINTER(CgStats),
                                   C.POINTER(CgStats), dp, dp, vp, vp]
    L.gv_cg_solve_aat2w.argtypes = [vp, vp, vp, vp, C.c_double, C.c_double, C.c_int, vp, vp, vp, C.POINTER(CgStats),
                                    C.POINTER(CgStats), dp, dp, vp, vp, C.POINTER(AatWarm)]
    L.gv_pvals_loo.argtypes = [vp, vp, vp, vp, dp]
    L.gv_pvals_loco.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int), dp]
    L.gv_pvals_loco_pred.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_int), dp, dp]
    L.gv_comm_unique_id.argtypes = [C.c_void_p]
    L.gv_comm_init.argtypes = [vp, C.c_int, C.c_int, C.c_void_p]
    L.gv_comm_init_local.argtypes = [vp, C.c_int, C.c_int, C.c_int]
    L.gv_comm_init_callback.argtypes = [vp, C.c_int, C.c_int, ALLREDUCE_FN, C.c_void_p]
    L.gv_comm_share.argtypes = [vp, vp]
    L.gv_set_overlap.argtypes = [vp, C.c_int]
    L.gv_comm_rank.argtypes = [vp]
    L.gv_comm_size.argtypes = [vp]
    L.gv_set_timing.argtypes = [vp, C.c_int]
    L.gv_get_counters.argtypes = [vp, C.POINTER(Counters)]
    L.gv_reset_counters.argtypes = [vp]
    L.gv_get_decomp.argtypes = [vp, C.POINTER(DecompInfo)]
    L.gv_set_decomp.argtypes = [vp, C.c_int, C.POINTER(DecompInfo)]
    L.gv_tune_info.argtypes = [vp, dp, C.POINTER(C.c_int)]
    L.gv_ingest_info.argtypes = [vp, dp, dp]
    L.gv_ingest_info2.argtypes = [vp, C.POINTER(IngestStats)]
    L.gv_set_expected_passes.argtypes = [vp, C.c_int64]
    L.gv_debug_force_multi.argtypes = [vp, C.c_int, C.c_int]
    L.gv_bind_host_numa.argtypes = [C.c_int, C.POINTER(C.c_int)]
    L.gv_copy_bandwidth.argtypes = [vp, C.c_size_t, C.c_int, dp]
    L.gv_read_bandwidth.argtypes = [vp, C.c_size_t, C.c_int, dp]
    _LIB = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _up(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def comm_unique_id():
    buf = (C.c_ubyte * 128)()
    if load().gv_comm_unique_id(buf):
        raise GvError(load().gv_last_error(None).decode())
    return bytes(buf)


class Vec:
    """Device-resident fp64 vector (gv_vec)."""

    def __init__(self, shard, space):
        self.shard = shard
        h = C.c_void_p()
        shard._ck(shard.L.gv_vec_alloc(shard.h, space, C.byref(h)))
        self.h = h
        self.space = space
        self.n = shard.L.gv_vec_len(h)

    def upload(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        assert a.size == self.n, (a.size, self.n)
        self.shard._ck(self.shard.L.gv_vec_upload(self.shard.h, self.h, _dp(a)))
        return self

    def download(self):
        out = np.empty(self.n)
        self.shard._ck(self.shard.L.gv_vec_download(self.shard.h, self.h, _dp(out)))
        return out

    def fill(self, v):
        self.shard._ck(self.shard.L.gv_vec_fill(self.shard.h, self.h, float(v)))
        return self

    def free(self):
        if self.h:
            self.shard.L.gv_vec_free(self.shard.h, self.h)
            self.h = None


class Shard:
    """One marker shard resident on one GPU: the device side of the reference's `class data`
    (data.hpp:93-140).  Method names follow the reference (Ax, ATx, compute_markers_statistics...)."""

    def __init__(self, N, M, Mt=None, S=0, device=0, anchor=False):
        """Nothing but gv_create + gv_set_dims: the context keeps the C ABI's defaults (kernel mode 1, no raw rows, layout
        picked at ingest).  anchor=True opts into the parity-anchor set-up of the tests that compare the two kernel families or
        read the rows back: raw rows + two stripe sets resident, fp64 VALU family selected (gv_set_layout(1, 1),
        gv_set_kernel_mode(0))."""
        self.L = load()
        h = C.c_void_p()
        if self.L.gv_create(device, C.byref(h)):
            raise GvError(self.L.gv_last_error(None).decode())
        self.h = h
        self.N, self.M, self.Mt, self.S = N, M, (M if Mt is None else Mt), S
        self._ck(self.L.gv_set_dims(h, N, M, self.Mt, S))
        self.mbytes = self.L.gv_mbytes(h)
        if anchor:
            self.set_layout(True, 1)
            self.set_kernel_mode(0)

    def _ck(self, rc):
        if rc:
            raise GvError(self.L.gv_last_error(self.h).decode())

    def close(self):
        if self.h:
            self.L.gv_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- dataset -------------------------------------------------------------------------------------
    def upload_bed(self, bed):
        bed = np.ascontiguousarray(bed, dtype=np.uint8)
        self._ck(self.L.gv_upload_bed(self.h, _up(bed), bed.size))

    def upload_bed_file(self, path, offset=None):
        off = 3 + self.S * self.mbytes if offset is None else offset
        self._ck(self.L.gv_upload_bed_file(self.h, path.encode(), off))

    def synth_bed(self, seed, miss_ppm=5000, ld_block=0, ld_ppm=0):
        if ld_block:
            self._ck(self.L.gv_synth_bed_ld(self.h, seed, miss_ppm, ld_block, ld_ppm))
        else:
            self._ck(self.L.gv_synth_bed(self.h, seed, miss_ppm))

    def download_bed(self):
        out = np.empty(self.M * self.mbytes, dtype=np.uint8)
        self._ck(self.L.gv_download_bed(self.h, _up(out), out.size))
        return out

    def set_mask(self, mask4, nonas):
        m = np.ascontiguousarray(mask4, dtype=np.uint8)
        assert m.size == self.mbytes
        self._ck(self.L.gv_set_mask(self.h, _up(m), nonas))

    def compute_markers_statistics(self, alpha_scale=1.0):
        self._ck(self.L.gv_marker_stats(self.h, alpha_scale))

    def marker_stats(self):
        mave, msig = np.empty(self.M), np.empty(self.M)
        self._ck(self.L.gv_get_marker_stats(self.h, _dp(mave), _dp(msig)))
        return mave, msig

    def set_layout(self, raw_rows=True, stripes=True):
        self._ck(self.L.gv_set_layout(self.h, int(raw_rows), int(stripes)))

    def get_layout(self):
        return self.L.gv_get_layout(self.h)

    def set_kernel_mode(self, mode):
        self._ck(self.L.gv_set_kernel_mode(self.h, mode))

    def get_kernel_mode(self):
        return self.L.gv_get_kernel_mode(self.h)

    # ---- host-signature matvecs (data::Ax / data::ATx) ---------------------------------------------------
    def Ax(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.M
        out = np.empty(4 * self.mbytes)
        self._ck(self.L.gv_ax(self.h, _dp(x), _dp(out)))
        return out

    def ATx(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        assert p.size == 4 * self.mbytes
        out = np.empty(self.M)
        self._ck(self.L.gv_atx(self.h, _dp(p), _dp(out)))
        return out

    # ---- device vectors -------------------------------------------------------------------------------
    def vec(self, space, data=None):
        v = Vec(self, space)
        if data is not None:
            v.upload(data)
        return v

    def vecM(self, data=None):
        return self.vec(SPACE_M, data)

    def vecN(self, data=None):
        return self.vec(SPACE_N, data)

    def set_phen(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        assert y.size == self.N
        v = Vec(self, SPACE_N)
        self._ck(self.L.gv_set_phen(self.h, v.h, _dp(y)))
        return v

    def ax_dev(self, x, out):
        self._ck(self.L.gv_ax_dev(self.h, x.h, out.h))

    def atx_dev(self, p, out):
        self._ck(self.L.gv_atx_dev(self.h, p.h, out.h))

    def axpby(self, out, a, x, b=0.0, y=None):
        self._ck(self.L.gv_vec_axpby(self.h, out.h, a, x.h, b, y.h if y is not None else None))

    def copy(self, dst, src):
        self._ck(self.L.gv_vec_copy(self.h, dst.h, src.h))

    def dot(self, x, y, sync=1):
        out = C.c_double()
        self._ck(self.L.gv_vec_dot(self.h, x.h, y.h, sync, C.byref(out)))
        return out.value

    def dots(self, pairs, sync=1):
        n = len(pairs)
        xs = (C.c_void_p * n)(*[p[0].h for p in pairs])
        ys = (C.c_void_p * n)(*[p[1].h for p in pairs])
        out = np.empty(n)
        self._ck(self.L.gv_vec_dots(self.h, n, xs, ys, sync, _dp(out)))
        return out

    def dots_ex(self, specs):
        """specs: (xa, xb, ya, yb, sync) tuples, xb / yb may be None: <xa - xb, ya - yb> in one launch and one read-back"""
        n = len(specs)
        arr = (DotSpec * n)()
        for k, (xa, xb, ya, yb, sync) in enumerate(specs):
            arr[k] = DotSpec(xa.h, xb.h if xb is not None else None, ya.h, yb.h if yb is not None else None, int(sync))
        out = np.empty(n)
        self._ck(self.L.gv_vec_dots_ex(self.h, n, arr, _dp(out)))
        return out

    def lmmse_mult(self, v, tau, gam2, out):
        self._ck(self.L.gv_lmmse_mult(self.h, v.h, tau, gam2, out.h))

    def cg_solve(self, v, mu_start, tau, gam2, denoiser, max_iter, mu_out):
        st = CgStats()
        rr = np.zeros(max(max_iter, 1))
        self._ck(self.L.gv_cg_solve(self.h, v.h, mu_start.h if mu_start is not None else None, tau, gam2, denoiser,
                                    max_iter, mu_out.h, C.byref(st), _dp(rr)))
        return st, rr[:st.n_relres].copy()

    def ax2_dev(self, xa, xb, outa, outb):
        self._ck(self.L.gv_ax2_dev(self.h, xa.h, xb.h, outa.h, outb.h))

    def atx2_dev(self, pa, pb, outa, outb):
        self._ck(self.L.gv_atx2_dev(self.h, pa.h, pb.h, outa.h, outb.h))

    def cg_solve2(self, v_a, mu_start_a, v_b, tau, gam2, max_iter, mu_a, mu_b):
        """LMMSE solve (a) and Onsager probe solve (b) in lock-step; returns ((stats_a, relres_a), (stats_b, relres_b))."""
        sa, sb = CgStats(), CgStats()
        ra, rb = np.zeros(max(max_iter, 1)), np.zeros(max(max_iter, 1))
        self._ck(self.L.gv_cg_solve2(self.h, v_a.h, mu_start_a.h if mu_start_a is not None else None, v_b.h, tau, gam2,
                                     max_iter, mu_a.h, mu_b.h, C.byref(sa), C.byref(sb), _dp(ra), _dp(rb)))
        return (sa, ra[:sa.n_relres].copy()), (sb, rb[:sb.n_relres].copy())

    def cg_solve2x(self, v_a, mu_start_a, v_b, tau, gam2, max_iter, mu_a, mu_b, ride_x=None, ride_out=None, a_mu_a=None,
                   ata_mu_b=None, ata_mu_start_a=None, a_mu_start_a=None, ata_mu_a=None, ata_v_b=None, have_ata_v_b=False):
        """gv_cg_solve2 plus its pass-free by-products (include/gvamp.h: gv_cg_extras) and, when one of the last three
        arguments is given, the warm start whose initial residual costs no pass (gv_cg_warm, gv_cg_solve2w)."""
        sa, sb = CgStats(), CgStats()
        ra, rb = np.zeros(max(max_iter, 1)), np.zeros(max(max_iter, 1))
        ex = CgExtras(*[v.h if v is not None else None for v in (ride_x, ride_out, a_mu_a, ata_mu_b)])
        ms = mu_start_a.h if mu_start_a is not None else None
        if ata_mu_start_a is None and a_mu_start_a is None and ata_mu_a is None and ata_v_b is None:
            self._ck(self.L.gv_cg_solve2x(self.h, v_a.h, ms, v_b.h, tau, gam2, max_iter, mu_a.h, mu_b.h, C.byref(sa),
                                          C.byref(sb), _dp(ra), _dp(rb), C.byref(ex)))
        else:
            wm = CgWarm(*[v.h if v is not None else None for v in (ata_mu_start_a, a_mu_start_a, ata_mu_a, ata_v_b)],
                        int(bool(have_ata_v_b)))
            self._ck(self.L.gv_cg_solve2w(self.h, v_a.h, ms, v_b.h, tau, gam2, max_iter, mu_a.h, mu_b.h, C.byref(sa),
                                          C.byref(sb), _dp(ra), _dp(rb), C.byref(ex), C.byref(wm)))
        return (sa, ra[:sa.n_relres].copy()), (sb, rb[:sb.n_relres].copy())

    def denoise(self, r1, gam1, probs, vars_scaled, x1_out, d_out=None):
        probs = np.ascontiguousarray(probs, dtype=np.float64)
        vs = np.ascontiguousarray(vars_scaled, dtype=np.float64)
        sums = np.empty(2)
        self._ck(self.L.gv_denoise(self.h, r1.h, gam1, _dp(probs), _dp(vs), probs.size, x1_out.h,
                                   d_out.h if d_out is not None else None, _dp(sums)))
        return sums

    def prior_estep(self, r1, gam1, lam, omegas, vars_scaled):
        om = np.ascontiguousarray(omegas, dtype=np.float64)
        vs = np.ascontiguousarray(vars_scaled, dtype=np.float64)
        sums = np.empty(1 + 2 * (om.size - 1))
        self._ck(self.L.gv_prior_estep(self.h, r1.h, gam1, lam, _dp(om), _dp(vs), om.size, _dp(sums)))
        return sums

    def probit_denoise(self, p1, y, tau1, probit_var, z1_out, m_cov=None):
        sums = np.empty(2)
        if m_cov is None:
            self._ck(self.L.gv_probit_denoise(self.h, p1.h, y.h, tau1, probit_var, z1_out.h, _dp(sums)))
        else:
            self._ck(self.L.gv_probit_denoise_cov(self.h, p1.h, y.h, m_cov.h, tau1, probit_var, z1_out.h, _dp(sums)))
        return sums

    def compute_people_statistics(self):
        """data::compute_people_statistics: (mave_people, msig_people, numb_people), 4*mbytes each."""
        n4 = 4 * self.mbytes
        a, b, c = np.empty(n4), np.empty(n4), np.empty(n4)
        self._ck(self.L.gv_people_stats(self.h, _dp(a), _dp(b), _dp(c)))
        return a, b, c

    def cg_solve_aat(self, v, mu_start, tau, gam2, max_iter, mu_out):
        st = CgStats()
        rr = np.zeros(max(max_iter, 1))
        self._ck(self.L.gv_cg_solve_aat(self.h, v.h, mu_start.h if mu_start is not None else None, tau, gam2, max_iter,
                                        mu_out.h, C.byref(st), _dp(rr)))
        return st, rr[:st.n_relres].copy()

    def cg_solve_aat2(self, v_a, mu_start_a, v_b, tau, gam2, max_iter, mu_a, at_mu_a, mu_b, aat_mu_a=None, ata_mu_b=None,
                      aat_mu_start_a=None, at_mu_start_a=None, accumulate_at_mu_a=False, ata_v_b=None, have_ata_v_b=False,
                      pre_x=None, pre_out=None, ride_x=None, ride_out=None, pre_scale=0.0):
        """gv_cg_solve_aat (system a, N-space) and the Onsager gv_cg_solve (system b, M-space) on shared passes; the last three
        arguments are gv_aat_warm (gv_cg_solve_aat2w): A A^T mu_start_a / A^T mu_start_a known from the previous call, and
        A^T mu_a accumulated inside the solve instead of by a closing pass."""
        sa, sb = CgStats(), CgStats()
        ra, rb = np.zeros(max(max_iter, 1)), np.zeros(max(max_iter, 1))
        wm = AatWarm(aat_mu_start_a.h if aat_mu_start_a is not None else None,
                     at_mu_start_a.h if at_mu_start_a is not None else None, int(bool(accumulate_at_mu_a)),
                     ata_v_b.h if ata_v_b is not None else None, int(bool(have_ata_v_b)),
                     *[q.h if q is not None else None for q in (pre_x, pre_out, ride_x, ride_out)], float(pre_scale))
        self._ck(self.L.gv_cg_solve_aat2w(self.h, v_a.h, mu_start_a.h if mu_start_a is not None else None, v_b.h, tau, gam2,
                                          max_iter, mu_a.h, at_mu_a.h, mu_b.h, C.byref(sa), C.byref(sb), _dp(ra), _dp(rb),
                                          aat_mu_a.h if aat_mu_a is not None else None,
                                          ata_mu_b.h if ata_mu_b is not None else None, C.byref(wm)))
        return (sa, ra[:sa.n_relres].copy()), (sb, rb[:sb.n_relres].copy())

    def pvals_calc_loco_pred(self, z1, y, x1_hat, chrom):
        """gv_pvals_loco_pred: (pvals[M], predictors[23, 4*mbytes])"""
        out = np.zeros(max(self.M, 1))
        pred = np.zeros((23, 4 * self.mbytes))
        ch = np.ascontiguousarray(chrom, dtype=np.int32)
        assert ch.size == self.M
        self._ck(self.L.gv_pvals_loco_pred(self.h, z1.h, y.h, x1_hat.h, ch.ctypes.data_as(C.POINTER(C.c_int)), _dp(out), _dp(pred)))
        return out[:self.M].copy(), pred

    def pvals_calc(self, z1, y, x1_hat, chrom=None):
        """data::pvals_calc (chrom None) / data::pvals_calc_LOCO on device handles; returns pvals[M]."""
        out = np.zeros(max(self.M, 1))
        if chrom is None:
            self._ck(self.L.gv_pvals_loo(self.h, z1.h, y.h, x1_hat.h, _dp(out)))
        else:
            ch = np.ascontiguousarray(chrom, dtype=np.int32)
            assert ch.size == self.M
            self._ck(self.L.gv_pvals_loco(self.h, z1.h, y.h, x1_hat.h, ch.ctypes.data_as(C.POINTER(C.c_int)), _dp(out)))
        return out[:self.M].copy()

    def allreduce_host(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        self._ck(self.L.gv_allreduce_host(self.h, _dp(a), a.size))
        return a

    # ---- communicator / instrumentation --------------------------------------------------------------------
    def comm_init(self, nranks, rank, uid):
        buf = (C.c_ubyte * 128).from_buffer_copy(uid) if uid is not None else None
        self._ck(self.L.gv_comm_init(self.h, nranks, rank, buf))

    def comm_init_local(self, group, nranks, rank):
        self._ck(self.L.gv_comm_init_local(self.h, group, nranks, rank))

    def comm_init_callback(self, nranks, rank, allreduce):
        """gv_comm_init_callback: `allreduce(a)` sums the float64 numpy array `a` in place over the ranks (e.g.
        torch.distributed.all_reduce over gloo on torch.from_numpy(a))."""
        def _cb(_user, buf, n):
            try:
                allreduce(np.ctypeslib.as_array(buf, shape=(n,)))
                return 0
            except Exception:          # no exception may cross the C ABI
                import traceback
                traceback.print_exc()
                return 1
        self._cb_keep = ALLREDUCE_FN(_cb)          # keep the trampoline alive as long as the context
        self._ck(self.L.gv_comm_init_callback(self.h, nranks, rank, self._cb_keep, None))

    def comm_share(self, owner):
        """gv_comm_share: join the communicator another Shard of this process already holds"""
        self._ck(self.L.gv_comm_share(self.h, owner.h))
        if getattr(owner, "_cb_keep", None) is not None:
            self._cb_keep = owner._cb_keep

    def set_overlap(self, tiles):
        self._ck(self.L.gv_set_overlap(self.h, tiles))

    def force_multi(self, transport=1, delay_us=0):
        """gv_debug_force_multi (test hook): this one-rank shard takes the sharded branches over an in-stream exchange
        (1 loop-back with poisoning, 2 the 1-rank RCCL communicator, 3 both; 0 = off)."""
        self._ck(self.L.gv_debug_force_multi(self.h, transport, delay_us))

    def set_timing(self, on):
        self._ck(self.L.gv_set_timing(self.h, int(on)))

    def counters(self, reset=False):
        c = Counters()
        self._ck(self.L.gv_get_counters(self.h, C.byref(c)))
        if reset:
            self._ck(self.L.gv_reset_counters(self.h))
        return dict(n_ax=c.n_ax, n_atx=c.n_atx, ms_ax=c.ms_ax, ms_atx=c.ms_atx, ms_allreduce=c.ms_allreduce,
                    n_ax_kernel=c.n_ax_kernel, n_atx_kernel=c.n_atx_kernel, ms_ax_kernel=c.ms_ax_kernel,
                    ms_atx_kernel=c.ms_atx_kernel, n_ax_pass=c.n_ax_pass, n_atx_pass=c.n_atx_pass,
                    n_allreduce=c.n_allreduce)

    def ingest_info(self):
        """(seconds allocating the resident layouts, seconds filling them) of the last ingest"""
        a, f = C.c_double(), C.c_double()
        self._ck(self.L.gv_ingest_info(self.h, C.byref(a), C.byref(f)))
        return a.value, f.value

    def ingest_stats(self):
        """gv_ingest_info2 of the last ingest as a dict"""
        st = IngestStats()
        self._ck(self.L.gv_ingest_info2(self.h, C.byref(st)))
        return {"alloc_s": st.alloc_seconds, "fill_s": st.fill_seconds, "overlap_s": st.overlap_seconds,
                "resident_GB": st.resident_bytes / 1e9, "layout": st.layout, "expected_passes": st.expected_passes}

    def set_expected_passes(self, passes):
        self._ck(self.L.gv_set_expected_passes(self.h, int(passes)))

    def tune_info(self):
        """(seconds spent picking the decompositions, source: 'pending' / 'model' / 'measured' / 'cache' / 'fixed' / 'builtin')"""
        sec, src = C.c_double(), C.c_int()
        self._ck(self.L.gv_tune_info(self.h, C.byref(sec), C.byref(src)))
        return sec.value, {-1: "pending", 0: "model", 1: "measured", 2: "cache", 3: "fixed", 4: "builtin"}[src.value]

    def decomp(self):
        """work decomposition per streaming-kernel class (gv_get_decomp)"""
        d = (DecompInfo * 4)()
        self._ck(self.L.gv_get_decomp(self.h, d))
        names = ("atx", "atx2", "ax", "ax2")
        return {n: (({"balanced_cells": int(x.balanced_cells)} | ({"whole_quads": int(x.whole_quads)} if x.whole_quads > 0 else {}))
                    if x.balanced_cells > 0 else ({"ks": x.ks, "taper": round(float(x.taper), 2)} | ({"geo": round(float(x.geo), 2)} if x.geo > 0 else {})))
                | ({"wgs_per_cu": 2} if x.wgs_per_cu == 2 else {}) | ({"xcd_skew": round(float(x.xcd_skew), 3)} if x.xcd_skew != 0 else {})
                | {"prio": x.prio, "tuned": bool(x.tuned)}
                for n, x in zip(names, d)}

    def set_decomp(self, cls, ks=1, balanced_cells=0, whole_quads=0, prio=0, taper=0.0, geo=0.0, wgs_per_cu=0, xcd_skew=0.0):
        """pins the decomposition of one streaming-kernel class (gv_set_decomp); cls: 0 atx, 1 atx2, 2 ax, 3 ax2 or its name"""
        if isinstance(cls, str):
            cls = ("atx", "atx2", "ax", "ax2").index(cls)
        d = DecompInfo(ks, balanced_cells, prio, taper, 0, whole_quads, geo, wgs_per_cu, xcd_skew)
        self._ck(self.L.gv_set_decomp(self.h, cls, C.byref(d)))

    def synchronize(self):
        self._ck(self.L.gv_synchronize(self.h))

    def copy_bandwidth(self, nbytes=1 << 30, reps=10):
        out = C.c_double()
        self._ck(self.L.gv_copy_bandwidth(self.h, nbytes, reps, C.byref(out)))
        return out.value

    def read_bandwidth(self, nbytes=1 << 30, reps=5):
        """read-only stream probe over the resident stripes (or a scratch buffer): GB/s"""
        out = C.c_double()
        self._ck(self.L.gv_read_bandwidth(self.h, nbytes, reps, C.byref(out)))
        return out.value
