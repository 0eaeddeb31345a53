"""Read-only stream ceiling for the three lane -> piece patterns of the streaming kernels (development)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gvamp_amd import capi
with capi.Shard(400000, 250000) as sh:
    sh.set_layout(False, 2)
    sh.set_kernel_mode(1)
    sh.synth_bed(1, 5000)
    for perm, name in ((0, "lane-linear (stripe kernels)"), (1, "tile layout, ATx side"), (2, "tile layout, Ax side")):
        os.environ["GV_READ_PERM"] = str(perm)
        print(name, [round(sh.read_bandwidth(1 << 30, 5), 1) for _ in range(3)], flush=True)
