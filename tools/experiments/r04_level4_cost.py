import sys, os, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
c = synth.make_config('c3')
pts, s = c['points'], 1.0 / c['sigma'].ravel()
cg = ShrinkwrapMeshConjGrad(TriMesh(c['vertices'].copy(), c['faces']), pts)
for b in range(3):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.optimize_layout()
cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
level = int(sys.argv[1])
cg.set_profiling(level)
for b in range(4):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
t0 = time.perf_counter()
nb = 40
for b in range(nb):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
print('level %d%s: %.1f us per block' % (level, ' (no events)' if os.environ.get('NW_L4_NOEVENTS') else '', (time.perf_counter() - t0) / nb * 1e6))
