"""Host-side cost of a block in 'halo' mode, measured where it can be on a one-GPU box: ONE rank over RCCL (world size 1, so the collectives are
real RCCL calls on the kernels' stream but carry nothing between GPUs) runs HaloScene on the headline workload; reports, per block of 5
iterations: wall time, the time the host spends enqueueing the phases and collectives (before it starts waiting), and the block tail.
usage: python tools/halo_host_time.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from ch_shrinkwrap_amd import synth, parallel
from ch_shrinkwrap_amd.trimesh import TriMesh

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
cfg = synth.make_config('c3', scale=scale, seed=0)
mesh = TriMesh(cfg['vertices'], cfg['faces'])
ts = torch.cuda.Stream()
scene = parallel.HaloScene(mesh, cfg['points'], dist, halo=100.0, torch_stream=ts)
s = 1.0 / cfg['sigma'].ravel()
orig = parallel.run_search
enq = []


def timed_run_search(*a, **k):
    t = time.perf_counter()
    r = orig(*a, **k)
    enq.append(time.perf_counter() - t)
    return r


parallel.run_search = timed_run_search
for b in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    scene.search(cfg['lams'], 5, s)
    dt = time.perf_counter() - t0
    if b == 1:
        scene.ex.cg.optimize_layout()
    if b >= 4:
        print('block %2d: wall %.3f ms; run_search (enqueue of 5 iterations + wait for the logs) %.3f ms; tail: collectives + copy %.3f ms, host mesh %.3f ms' % (
            b, dt * 1e3, enq[-1] * 1e3, scene.host_ms['block_tail_collectives_and_copy'], scene.host_ms['block_tail_host_mesh']), flush=True)
print('setup (partition + upload, once per topology): %.1f ms' % scene.host_ms['setup'])
dist.destroy_process_group()
