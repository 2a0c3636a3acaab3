import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from ch_shrinkwrap_amd import synth, parallel
from ch_shrinkwrap_amd import _lib as nw
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import NativeContext
class One(object):
    get_rank = staticmethod(lambda: 0)
    get_world_size = staticmethod(lambda: 1)
cfg = synth.make_config('c3')
pts, s_inv = cfg['points'], 1.0 / cfg['sigma'].ravel()
mesh = TriMesh(cfg['vertices'], cfg['faces'])
native = NativeContext(0)
comm = parallel.NativeComm(native, One)
scene = parallel.HaloScene(mesh, pts, One, halo=60.0, native=native, comm=comm)
for _ in range(3): scene.search(cfg['lams'], 5, s_inv)
scene.optimize_layout(); scene.search(cfg['lams'], 5, s_inv); scene.optimize_layout(); scene.search(cfg['lams'], 5, s_inv)
import cProfile, pstats, io
nb = 60
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
for _ in range(nb): scene.search(cfg['lams'], 5, s_inv)
pr.disable()
dt = time.perf_counter() - t0
print('per block: total %.1f us; repartitions %d' % (dt / nb * 1e6, scene.repartitions))
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(14)
print('\n'.join(l[:150] for l in out.getvalue().splitlines() if l.strip() and ('{' in l or '.py' in l)))
comm.close()
