"""Can RCCL form a 2-rank communicator from two processes on ONE GPU?  (development probe: if it can, the RCCL paths with
nranks > 1 are testable on a 1-GPU box)   python scripts/rccl_same_gpu_probe.py"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import os, sys, time
sys.path.insert(0, %r)
import numpy as np
from gvamp_amd import capi, synth
rank = int(sys.argv[1]); path = sys.argv[2]
if rank == 0:
    uid = capi.comm_unique_id()
    open(path + ".tmp", "wb").write(uid); os.rename(path + ".tmp", path)
else:
    while not os.path.exists(path): time.sleep(0.05)
    uid = open(path, "rb").read()
N, Mt = 2000, 4000
bed = synth.synth_bed(N, Mt, seed=1)
mb = N // 4
M, S = Mt // 2, rank * (Mt // 2)
with capi.Shard(N, M, Mt=Mt, S=S, device=0) as sh:
    sh.upload_bed(bed[S * mb:(S + M) * mb])
    sh.set_kernel_mode(1)
    sh.comm_init(2, rank, uid)
    sh.compute_markers_statistics()
    z = sh.Ax(np.ones(M))
    print("rank", rank, "Ax norm", float(np.linalg.norm(z)), flush=True)
""" % ROOT
d = tempfile.mkdtemp()
open(os.path.join(d, "w.py"), "w").write(WORKER)
env = dict(os.environ, NCCL_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
ps = [subprocess.Popen([sys.executable, os.path.join(d, "w.py"), str(r), os.path.join(d, "id")], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True) for r in range(2)]
t0 = time.time()
for p in ps:
    try:
        out = p.communicate(timeout=max(1, 90 - (time.time() - t0)))[0]
    except subprocess.TimeoutExpired:
        p.kill(); out = "TIMEOUT\n" + p.communicate()[0]
    print("rc", p.returncode, out[-1500:])
