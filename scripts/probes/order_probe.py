import sys
sys.path.insert(0,'/root/repo')
order=sys.argv[1]
if order=='torch_first':
    import torch; print('torch cuda', torch.cuda.is_available()); torch.cuda.set_device(0); torch.cuda.synchronize()
from gvamp_amd import capi
capi.load()
if order=='lib_first':
    sh=capi.Shard(100,10); print('shard ok before torch'); sh.close()
    import torch; print('torch cuda', torch.cuda.is_available())
sh=capi.Shard(2000,100); sh.synth_bed(1,5000); sh.compute_markers_statistics(); print('ok', order, sh.marker_stats()[0][:3]); sh.close()
