"""Contexts created, used and destroyed from several threads at once (development; run on a GPU box):
   python scripts/stress_lifecycle.py [seconds] [threads]      (GV_DBG_LIB=<path> picks an experimental library)
The in-process rank groups of the tests do exactly this; a crash here is a lifecycle race, in this library or below it."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gvamp_amd import capi, synth
if os.environ.get("GV_DBG_LIB"):
    capi.LIB_PATH = os.environ["GV_DBG_LIB"]

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N, M = 515, 300
bed = synth.synth_bed(N, M, seed=1)
counts = [0] * (nthreads + 1)
errors = []
t_end = time.time() + secs


def work(k):
    rng = np.random.default_rng(k)
    try:
        while time.time() < t_end:
            with capi.Shard(N, M) as sh:
                sh.set_layout(False, 1 + (counts[k] & 1))
                sh.set_kernel_mode(1)
                sh.upload_bed(bed)
                sh.compute_markers_statistics()
                v, mu = sh.vecM(rng.standard_normal(M)), sh.vecM()
                sh.cg_solve(v, None, 2.0, 0.7, 1, 5, mu)
                if counts[k] % 3 == 0:
                    sh.set_overlap(2)
                    sh.Ax(rng.standard_normal(M))
            counts[k] += 1
    except Exception as e:   # noqa: BLE001
        errors.append((k, repr(e)))


th = [threading.Thread(target=work, args=(k,), daemon=True) for k in range(nthreads)]
for t in th:
    t.start()
work(nthreads)
for t in th:
    t.join(timeout=60)
print("cycles per thread", counts, "errors", errors, flush=True)
sys.exit(1 if errors or any(t.is_alive() for t in th) else 0)
