"""N > 1 path on CPU: two processes over torch.distributed gloo, one marker shard each, all-reduce through the
communicator callback -- the same sharded algorithm the GPU build runs over RCCL."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from gvamp_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    import torch, torch.distributed as dist
    from oracle import gvoracle as go
    from gvamp_amd import synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    N, Mt = 400, 1000
    bed = synth.synth_bed(N, Mt, seed=77, miss_ppm=5000)
    beta, y = go.sim_phen(bed, N, Mt, 0.5, 50, 5)
    def allreduce(a):
        t = torch.from_numpy(a)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    r = go.infere(bed, N, Mt, y, [0.9, 0.07, 0.03], [0, 0.001, 0.01], nshards=world, shard_rank=rank, iterations=3,
                  CG_max_iter=15, rho=0.5, seed=5, true_signal=beta, allreduce=allreduce)
    M, S = go.divide_work(Mt, world, rank)
    np.save(os.path.join(%(out)r, "x_rank%%d.npy" %% rank), r.x_est[S:S + M])
    np.save(os.path.join(%(out)r, "trace_rank%%d.npy" %% rank), np.array([[t["gamw"], t["alpha2"], t["cg_iters"]] for t in r.trace]))
    dist.barrier()
    dist.destroy_process_group()
""")


def test_two_process_gloo_matches_in_process_shards(oracle, tmp_path):
    N, Mt = 400, 1000
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": str(tmp_path)})
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    bed = synth.synth_bed(N, Mt, seed=77, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, Mt, 0.5, 50, 5)
    ref2 = oracle.infere(bed, N, Mt, y, [0.9, 0.07, 0.03], [0, 0.001, 0.01], nshards=2, iterations=3, CG_max_iter=15,
                         rho=0.5, seed=5, true_signal=beta)
    x = np.concatenate([np.load(tmp_path / ("x_rank%d.npy" % r)) for r in range(2)])
    assert np.linalg.norm(x - ref2.x_est) / np.linalg.norm(ref2.x_est) < 1e-12
    t0, t1 = np.load(tmp_path / "trace_rank0.npy"), np.load(tmp_path / "trace_rank1.npy")
    assert np.array_equal(t0, t1)                                   # every rank sees the same scalars
    assert np.allclose(t0[:, 0], [t["gamw"] for t in ref2.trace], rtol=1e-12)
    # sharding changes the Hutchinson probes (seed + S): close to, but not equal to, the single-shard run
    ref1 = oracle.infere(bed, N, Mt, y, [0.9, 0.07, 0.03], [0, 0.001, 0.01], nshards=1, iterations=3, CG_max_iter=15,
                         rho=0.5, seed=5, true_signal=beta)
    d = np.linalg.norm(ref2.x_est - ref1.x_est) / np.linalg.norm(ref1.x_est)
    assert 1e-9 < d < 0.2


SHM_WORKER = textwrap.dedent("""
    import ctypes as C, os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    from oracle import gvoracle as go
    from gvamp_amd import synth, hostapi
    rank, world = int(sys.argv[1]), int(sys.argv[2])
    L = hostapi.load()                       # libgvamp_host.so: the transport is product code (host/shm_comm.cpp), no GPU in it
    L.gvh_shm_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
    L.gvh_shm_allreduce.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_size_t]
    L.gvh_shm_close.argtypes = [C.c_void_p]
    h = C.c_void_p()
    if L.gvh_shm_open(sys.argv[3].encode(), world, rank, 256, C.byref(h)):      # (capacity below N: the messages travel in pieces)
        sys.exit("open failed: " + L.gvh_last_error().decode())
    N, Mt = 400, 1000
    bed = synth.synth_bed(N, Mt, seed=77, miss_ppm=5000)
    beta, y = go.sim_phen(bed, N, Mt, 0.5, 50, 5)
    def allreduce(a):
        assert a.dtype == np.float64 and a.flags.c_contiguous
        if L.gvh_shm_allreduce(h, a.ctypes.data_as(C.POINTER(C.c_double)), a.size):
            sys.exit("allreduce failed")
    r = go.infere(bed, N, Mt, y, [0.9, 0.07, 0.03], [0, 0.001, 0.01], nshards=world, shard_rank=rank, iterations=3,
                  CG_max_iter=15, rho=0.5, seed=5, true_signal=beta, allreduce=allreduce)
    M, S = go.divide_work(Mt, world, rank)
    np.save(os.path.join(%(out)r, "sx_rank%%d.npy" %% rank), r.x_est[S:S + M])
    np.save(os.path.join(%(out)r, "strace_rank%%d.npy" %% rank), np.array([[t["gamw"], t["alpha2"], t["cg_iters"]] for t in r.trace]))
    L.gvh_shm_close(h)
""")


def test_three_processes_over_the_products_host_transport(oracle, tmp_path):
    """The same sharded run with the PRODUCT's transport in place of gloo: three processes meet in the shared-memory segment of
    GVAMP_COMM=host (gvh_shm_open / gvh_shm_allreduce of libgvamp_host.so, what host/data.cpp hands to gv_comm_init_callback) and
    every all-reduce of the sharded algorithm -- N-vectors in pieces, scalars, the E-step sums -- goes through it.  Sums in rank
    order, as the oracle's in-process shards add them: the same bits."""
    N, Mt, world = 400, 1000, 3
    script = tmp_path / "shm_worker.py"
    script.write_text(SHM_WORKER % {"root": ROOT, "out": str(tmp_path)})
    name = "/gvamp_gloo_test_%d" % os.getpid()
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), name], env=dict(os.environ, OMP_NUM_THREADS="1"),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    bed = synth.synth_bed(N, Mt, seed=77, miss_ppm=5000)
    beta, y = oracle.sim_phen(bed, N, Mt, 0.5, 50, 5)
    ref = oracle.infere(bed, N, Mt, y, [0.9, 0.07, 0.03], [0, 0.001, 0.01], nshards=world, iterations=3, CG_max_iter=15,
                        rho=0.5, seed=5, true_signal=beta)
    x = np.concatenate([np.load(tmp_path / ("sx_rank%d.npy" % r)) for r in range(world)])
    assert np.linalg.norm(x - ref.x_est) / np.linalg.norm(ref.x_est) < 1e-12
    traces = [np.load(tmp_path / ("strace_rank%d.npy" % r)) for r in range(world)]
    assert all(np.array_equal(traces[0], t) for t in traces[1:])              # every rank sees the same scalars
    assert np.allclose(traces[0][:, 0], [t["gamw"] for t in ref.trace], rtol=1e-12)
    assert [int(v) for v in traces[0][:, 2]] == [t["cg_iters"] for t in ref.trace]
