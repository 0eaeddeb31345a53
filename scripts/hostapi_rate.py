import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from gvamp_amd import capi
N, M = 400000, 1000000
with capi.Shard(N, M) as sh:
    sh.set_layout(False, True); sh.set_kernel_mode(1); sh.synth_bed(1234, 5000); sh.compute_markers_statistics()
    rng = np.random.default_rng(0)
    x = rng.standard_normal(M); p = np.zeros(4 * ((N + 3) // 4)); p[:N] = rng.standard_normal(N)
    sh.Ax(x); sh.ATx(p)
    mb = (N + 3) // 4; nbytes = M * mb + 24 * M + 32 * mb
    t = time.perf_counter()
    for _ in range(5): z = sh.Ax(x)
    ta = (time.perf_counter() - t) / 5
    t = time.perf_counter()
    for _ in range(5): w = sh.ATx(p)
    tb = (time.perf_counter() - t) / 5
    print("host-pointer gv_ax  %.2f ms = %.0f GB/s ; gv_atx %.2f ms = %.0f GB/s (vectors cross PCIe both ways, pageable numpy buffers)" % (ta * 1e3, nbytes / ta / 1e9, tb * 1e3, nbytes / tb / 1e9))
