"""Developer tool: steady-state cost of the NN query against the absolute cell size, for a config with the number of
localizations overridden (how does the best cell size move with the point density?).
usage: python tools/cell_sweep.py <config> <n_points or 0> h1 h2 ...      (runs on the GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

name, n_over = sys.argv[1], int(sys.argv[2])
hs = [float(x) for x in sys.argv[3:]]
c = synth.make_config(name, scale=1.0, seed=0)
pts = c['points']
if n_over:
    rng = np.random.default_rng(1)
    if n_over <= pts.shape[0]:
        pts = pts[rng.choice(pts.shape[0], n_over, replace=False)]
    else:                                   # denser cloud: resample the surface with fresh noise
        reps = int(np.ceil(n_over / pts.shape[0]))
        base = np.concatenate([pts] * reps)[:n_over]
        pts = (base + rng.normal(scale=3.0, size=base.shape)).astype('f4')
    pts = np.ascontiguousarray(pts)
s = np.full(pts.size, 0.1, 'f4')
v0, f = c['vertices'], c['faces']
print('%s: N=%d M=%d F=%d' % (name, pts.shape[0], v0.shape[0], f.shape[0]))
for h in hs:
    if h > 0:
        os.environ['NW_CELL_SIZE'] = str(h)
    else:
        os.environ.pop('NW_CELL_SIZE', None)
    mesh = TriMesh(v0.copy(), f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    for _ in range(4):
        cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    cg.set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(2):
        cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    wall = (time.perf_counter() - t0) / 10
    nn = cg.stage_ms_total['nn']
    cg.set_profiling(2)
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    st = cg.stage_ms_total
    line = 'h=%5.1f  wall %.4f ms/iter  nn %.4f  attract %.4f  grid %.4f fixup %.4f  md %.2f' % (
        h, wall * 1e3, nn[0] / nn[1], st['attract'][0] / st['attract'][1], st['grid'][0] / st['grid'][1], st['fixup'][0] / st['fixup'][1], cg.mean_dist)
    print(line, flush=True)
    del cg
