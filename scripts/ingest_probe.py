import sys, time, os
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0)
from gvamp_amd import capi
capi.load()
for rep in range(2):
    t0 = time.perf_counter()
    sh = capi.Shard(400000, 1000000)
    sh.set_layout(False, 1)
    sh.set_kernel_mode(1)
    t1 = time.perf_counter()
    sh.synth_bed(1234, 5000)
    t2 = time.perf_counter()
    sh.synth_bed(1234, 5000)
    t3 = time.perf_counter()
    sh.compute_markers_statistics()
    t4 = time.perf_counter()
    sh.close()
    print(sys.argv[1:], "create %.3f first synth %.3f second synth %.3f stats %.3f close %.3f" % (t1 - t0, t2 - t1, t3 - t2, t4 - t3, time.perf_counter() - t4), flush=True)
