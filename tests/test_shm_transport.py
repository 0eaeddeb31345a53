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


def _wrapped(code, args, env):
    """a rank started THROUGH a wrapper shell, as a per-rank launch script would: its parent is the wrapper, not the launcher"""
    import shlex
    inner = " ".join(shlex.quote(str(x)) for x in [sys.executable, "-c", code] + list(args))
    return subprocess.Popen(["bash", "-c", inner + " ; exit $?"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def test_eight_ranks_behind_wrapper_scripts_meet_through_the_launchers_job_id(tmp_path):
    """First-contact shape of an 8-GPU node: 8 ranks whose parents are 8 different wrapper processes.  With a launcher job id in
    the environment (torchrun's TORCHELASTIC_RUN_ID here; Slurm / PMIx ids work the same) they derive one key, find one segment and
    sum identically -- the parent's pid, which round 3 used, would have given eight different keys."""
    env = {k: v for k, v in os.environ.items() if k not in ("GVAMP_RENDEZVOUS", "SLURM_JOB_ID", "PMIX_NAMESPACE", "PMI_JOBID")}
    env.update(TORCHELASTIC_RUN_ID="job_%d" % os.getpid(), MASTER_ADDR="127.0.0.1", MASTER_PORT="29656")
    n, cap = 8, 256
    procs = [_wrapped(WORKER % {"root": ROOT}, [r, n, cap, tmp_path / ("w%d.npz" % r), "-"], env) for r in range(n)]
    for p in procs:
        out, err = p.communicate(timeout=180)
        assert p.returncode == 0, err.decode()[-400:]
    got = [np.load(tmp_path / ("w%d.npz" % r)) for r in range(n)]
    for r in range(1, n):
        for k in range(6):
            assert np.array_equal(got[r]["arr_%d" % k], got[0]["arr_%d" % k])


ID_WORKER = r"""
import ctypes as C, os, sys
sys.path.insert(0, %(root)r)
from gvamp_amd import hostapi
L = hostapi.load()
L.gvh_exchange_id.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_double]
L.gvh_job_key_c.restype = C.c_char_p
rank, timeout = int(sys.argv[1]), float(sys.argv[2])
buf = C.create_string_buffer(bytes((7 * i + 3) %% 256 for i in range(128)) if rank == 0 else bytes(128), 128)
rc = L.gvh_exchange_id(None, rank, buf, timeout)
if rc:
    sys.stderr.write(L.gvh_last_error().decode())
    sys.exit(3)
print(L.gvh_job_key_c().decode(), buf.raw.hex())
"""


def test_rccl_id_file_reaches_eight_wrapped_ranks_and_a_lost_rank_says_where_it_looked(tmp_path):
    """The id-file rendezvous of the RCCL drivers (host/data.cpp: gv_host_world) without RCCL: rank 0 publishes 128 bytes, seven
    ranks behind wrapper shells receive them through the file named after the launcher's job id (Slurm's here).  Without any job id
    and without $GVAMP_RENDEZVOUS a wrapped rank derives a key from ITS parent: it times out -- and the message names the file, the
    key and the remedy instead of a bare 'cannot read'."""
    base = {k: v for k, v in os.environ.items() if k not in ("GVAMP_RENDEZVOUS", "TORCHELASTIC_RUN_ID", "PMIX_NAMESPACE", "PMI_JOBID",
                                                              "SLURM_JOB_ID", "SLURM_STEP_ID", "OMPI_MCA_ess_base_jobid")}
    env = dict(base, SLURM_JOB_ID="4242%d" % os.getpid(), SLURM_STEP_ID="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29657")
    procs = [_wrapped(ID_WORKER % {"root": ROOT}, [r, 60], env) for r in range(8)]
    outs = []
    for p in procs:
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, err.decode()[-400:]
        outs.append(out.decode().split())
    want = bytes((7 * i + 3) % 256 for i in range(128)).hex()
    assert all(o[1] == want for o in outs) and len({o[0] for o in outs}) == 1 and outs[0][0].startswith("slurm.")
    import glob          # (rank 0 of a driver removes its file after the first collective; here the test does)
    for f in glob.glob("/tmp/gvamp_rccl_id.*"):
        if os.path.getsize(f) == 128 and open(f, "rb").read().hex() == want:
            os.remove(f)
    lost = _wrapped(ID_WORKER % {"root": ROOT}, [1, 1.5], dict(base, MASTER_ADDR="127.0.0.1", MASTER_PORT="29658"))
    out, err = lost.communicate(timeout=60)
    assert lost.returncode == 3
    msg = err.decode()
    assert "/tmp/gvamp_rccl_id." in msg and "ppid." in msg and "GVAMP_RENDEZVOUS" in msg, msg
