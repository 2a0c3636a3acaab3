"""One rank, 'halo' mode, a long fit: growth of the nearest distances, drift, step and margin after every block, and when shares are cut again.
usage: python tools/experiments/r04_halo_margin_trace.py [blocks]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, parallel
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import NativeContext


class One(object):
    get_rank = staticmethod(lambda: 0)
    get_world_size = staticmethod(lambda: 1)


nb = int(sys.argv[1]) if len(sys.argv) > 1 else 80
cfg = synth.make_config('c3')
pts, s_inv = cfg['points'], 1.0 / cfg['sigma'].ravel()
mesh = TriMesh(cfg['vertices'], cfg['faces'])
native = NativeContext(0)
comm = parallel.NativeComm(native, One)
scene = parallel.HaloScene(mesh, pts, One, halo=60.0, native=native, comm=comm)
t0 = time.perf_counter()
for b in range(nb):
    r0 = scene.repartitions
    t = time.perf_counter()
    scene.search(cfg['lams'], 5, s_inv)
    print('block %3d: %.1f ms  growth %.2f drift %.2f step %.2f | cut margin %.2f, next margin %.2f%s%s' % (
        b, (time.perf_counter() - t) * 1e3, scene.max_dist, scene.drift, scene._last_step, scene._cut_margin, scene.margin,
        '  [shares cut before this block]' if scene.repartitions != r0 else '', '  [new shares wanted]' if scene.last_partition is None else ''), flush=True)
print('%d blocks, %d cuts, %.2f s' % (nb, scene.repartitions, time.perf_counter() - t0))
comm.close()
