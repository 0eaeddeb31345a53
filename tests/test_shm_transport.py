"""The host transport behind GVAMP_COMM=host (gvamp_amd/csrc/host/shm_comm.cpp, exported by libgvamp_host.so): ranks are
processes of one node that meet in a POSIX shared-memory segment and sum in rank order.  No GPU involved: this is the piece that
stands in for MPI_Allreduce (data.cpp:928/:995, utilities.cpp:203) when the sharded drivers run as processes sharing one GPU."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from gvamp_amd import hostapi
L = hostapi.load()
L.gvh_shm_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
L.gvh_shm_allreduce.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
L.gvh_shm_close.argtypes = [C.c_void_p]
rank, n, cap, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
h = C.c_void_p()
if L.gvh_shm_open(sys.argv[5].encode() if sys.argv[5] != "-" else None, n, rank, cap, C.byref(h)):
    sys.exit("open failed: " + L.gvh_last_error().decode())
res = []
for k, size in enumerate((1, cap - 1, cap, cap + 1, 3 * cap + 7, 5)):
    a = np.random.default_rng(1000 * k + rank).standard_normal(size) * 10.0 ** (rank - 2)
    if L.gvh_shm_allreduce(h, a.ctypes.data_as(C.POINTER(C.c_double)), size):
        sys.exit("allreduce failed")
    res.append(a)
L.gvh_shm_close(h)
np.savez(out, *res)
"""


def _run(n, cap, name, tmp_path, env=None):
    procs = [subprocess.Popen([sys.executable, "-c", WORKER % {"root": ROOT}, str(r), str(n), str(cap), str(tmp_path / ("r%d.npz" % r)), name],
                              env=env) for r in range(n)]
    for p in procs:
        assert p.wait(timeout=180) == 0
    return [np.load(tmp_path / ("r%d.npz" % r)) for r in range(n)]


def test_shm_allreduce_sums_in_rank_order_and_is_identical_on_every_rank(tmp_path):
    n, cap = 4, 1000
    got = _run(n, cap, "/gvamp_test_%d" % os.getpid(), tmp_path)
    for k, size in enumerate((1, cap - 1, cap, cap + 1, 3 * cap + 7, 5)):
        want = np.zeros(size)
        for r in range(n):                                      # rank order, as every rank adds them
            a = np.random.default_rng(1000 * k + r).standard_normal(size) * 10.0 ** (r - 2)
            want = a.copy() if r == 0 else want + a
        for r in range(n):
            assert np.array_equal(got[r]["arr_%d" % k], want), (k, r)


def test_shm_default_name_comes_from_the_launcher(tmp_path):
    """no name given: the ranks find one another through $GVAMP_RENDEZVOUS (scripts/run_sharded.py hands every job a fresh one)"""
    env = dict(os.environ, GVAMP_RENDEZVOUS=str(tmp_path / "rdv"), MASTER_PORT="29655")
    got = _run(2, 64, "-", tmp_path, env=env)
    assert np.array_equal(got[0]["arr_4"], got[1]["arr_4"])
