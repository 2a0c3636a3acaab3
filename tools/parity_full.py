"""One-off evidence run: HIP path vs the CPU oracle on a FULL-size workload (default C3: 1M localizations, 198 812 vertices;
`python tools/parity_full.py c4` for the 5M-localization network),
one block of 5 iterations + a second block; prints vertex RMS (relative to the bbox diagonal and in nm) and the number of
nearest-face disagreements per iteration."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
from oracle import nanowrap_oracle as O

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
c = synth.make_config(name, scale=1.0, seed=0)
print('%s: N=%d M=%d F=%d' % (name, c['points'].shape[0], c['vertices'].shape[0], c['faces'].shape[0]), flush=True)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
mo, mg = TriMesh(c['vertices'], c['faces']), TriMesh(c['vertices'], c['faces'])
cg = ShrinkwrapMeshConjGrad(mg, pts)
diag = np.linalg.norm(c['vertices'].max(0) - c['vertices'].min(0))
for block in range(2):
    trace = []
    t0 = time.perf_counter()
    r = O.search(mo.vertices.copy(), mo.vertex_normals.copy(), mo.neighbor_vertex_table(), mo.faces, pts, c['lams'], 5, s, trace=trace)
    t1 = time.perf_counter()
    mo._vertices['position'][:] = r.positions
    out = cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    t2 = time.perf_counter()
    d = np.sqrt(((out.astype('f8') - r.positions) ** 2).sum(1))
    mism = int((cg.nearest_face != trace[-1]['face']).sum())
    print('block %d: oracle %.2f s, hip %.4f s | vertex RMS %.3e nm = %.3e of bbox diagonal (max %.3e nm) | NN disagreements in the last iteration: %d of %d | ress rel diff %.2e'
          % (block, t1 - t0, t2 - t1, np.sqrt((d ** 2).mean()), np.sqrt((d ** 2).mean()) / diag, d.max(), mism, pts.shape[0],
             abs(float(cg.ress[-1]) - float(r.ress[-1])) / float(r.ress[-1])), flush=True)
