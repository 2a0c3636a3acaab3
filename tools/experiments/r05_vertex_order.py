"""What the order of the vertices costs an iteration: a mesh as seven remeshing steps leave it (new vertices appended at the end), the same mesh
with its vertices in Morton order of their positions, and with the vertices shuffled.  python3 tools/experiments/r05_vertex_order.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, remesh as R
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad


def morton(v):
    lo, hi = v.min(0), v.max(0)
    q = np.clip(((v - lo) / (hi - lo).max() * 1023.0), 0, 1023).astype(np.uint64)
    def spread(x):
        x = (x | (x << 16)) & np.uint64(0x030000FF); x = (x | (x << 8)) & np.uint64(0x0300F00F)
        x = (x | (x << 4)) & np.uint64(0x030C30C3); x = (x | (x << 2)) & np.uint64(0x09249249)
        return x
    return spread(q[:, 0]) | (spread(q[:, 1]) << np.uint64(1)) | (spread(q[:, 2]) << np.uint64(2))


scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
c = synth.make_config('c4', scale=scale, seed=0)
v, f = c['vertices'], c['faces']
L = float(TriMesh(v, f)._mean_edge_length)
rng = np.random.default_rng(0)
for k in range(7):
    v, f = R.remesh_device((v + rng.normal(0, 0.1, v.shape)).astype('f4'), f, 5, L * (0.99 - 0.01 * k))
print('mesh after 7 remeshing steps: %d vertices' % v.shape[0])
pts, s = c['points'], 1.0 / c['sigma'].ravel()
order = np.argsort(morton(v), kind='stable')
inv = np.empty_like(order); inv[order] = np.arange(order.size)
shuf = rng.permutation(v.shape[0]); sinv = np.empty_like(shuf); sinv[shuf] = np.arange(shuf.size)
for name, vv, ff in (('as the remesher left it', v, f), ('vertices in Morton order', v[order], inv[f].astype('i4')), ('vertices shuffled', v[shuf], sinv[f].astype('i4')),
                     ('as the remesher left it', v, f)):
    cg = ShrinkwrapMeshConjGrad(TriMesh(vv.copy(), ff), pts)
    for b in range(3):
        cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
        if b == 1:
            cg.optimize_layout()
    cg.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for b in range(20):
            cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
        cg.synchronize()
        best = min(best, (time.perf_counter() - t0) / 100)
    print('%-28s %.4f ms per iteration' % (name, best * 1e3), flush=True)
