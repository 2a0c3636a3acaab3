"""Per block of a C3 fit in 'halo' mode (one nccl rank): largest nearest distance, drift since the shares were cut, movement over the
block, and whether new shares are cut before the next block."""
import os, sys
sys.path.insert(0, '.')
os.environ.setdefault('TORCH_NCCL_CUDA_EVENT_CACHE', '0')
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29544')
import numpy as np, torch, torch.distributed as dist
from ch_shrinkwrap_amd import synth, parallel
from ch_shrinkwrap_amd.trimesh import TriMesh
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
cfg = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else 'c3', seed=0)
mesh = TriMesh(cfg['vertices'], cfg['faces'])
ts = torch.cuda.Stream()
scene = parallel.HaloScene(mesh, cfg['points'], dist, halo=float(sys.argv[2]) if len(sys.argv) > 2 else 100.0, torch_stream=ts)
s = 1.0 / cfg['sigma'].ravel()
prev = 0
for b in range(10):
    scene.search(cfg['lams'], 5, s)
    print('block %d: worst %.1f drift %.1f last step %.1f -> %s (partitions so far %d)' % (b, scene.max_dist, scene.drift, scene._last_step,
          'NEW SHARES' if scene.last_partition is None else 'keep', scene.repartitions), flush=True)
dist.destroy_process_group()
