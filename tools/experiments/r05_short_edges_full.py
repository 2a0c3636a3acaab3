"""Where do the edges of no length at the full C4 scale come from?  One block of the fit, then the block's mesh through the host remesher
(partitioned and serial) and the device remesher: edges below 1e-6 L in input and results.  python3 tools/experiments/r05_short_edges_full.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, remesh as R
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad


def short(v, f, L):
    e = np.linalg.norm(v[f].astype('f8') - v[np.roll(f, -1, 1)].astype('f8'), axis=2)
    return int((e <= 1e-6 * L).sum()), float(e.min()), int((e <= 1e-3 * L).sum())


c = synth.make_config('c4', scale=1.0, seed=0)
v, f = c['vertices'], c['faces']
pts, s = c['points'], 1.0 / c['sigma'].ravel()
L0 = float(TriMesh(v, f)._mean_edge_length)
print('start mesh: (edges <= 1e-6 L, shortest, edges <= 1e-3 L) =', short(v, f, L0), flush=True)
cg = ShrinkwrapMeshConjGrad(TriMesh(v.copy(), f), pts)
out = cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s).copy()
del cg
target = 1.13 * L0
print('after one block:', short(out, f, target), flush=True)
for name, fn in (('host, partitioned', lambda: R.remesh(out, f, 5, target, 0.5, 0)), ('host, serial', lambda: R.remesh(out, f, 5, target, 0.5, 0, serial=True)),
                 ('device', lambda: R.remesh_device(out, f, 5, target))):
    t0 = time.perf_counter()
    rv, rf = fn()[:2]
    print('%-18s %8.1f ms -> %d vertices: %s' % (name, (time.perf_counter() - t0) * 1e3, rv.shape[0], short(rv, rf, target)), flush=True)
