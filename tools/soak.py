"""Developer tool: soak run on the GPU box -- many blocks on the headline workload (stability of the converged state, no
status ever raised), then repeated context creation / destruction at several sizes with the free device memory watched."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

c = synth.make_config('c3', scale=1.0, seed=1)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
cg = ShrinkwrapMeshConjGrad(TriMesh(c['vertices'], c['faces']), pts)
t0 = time.time()
nblocks = int(sys.argv[1]) if len(sys.argv) > 1 else 400
level = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # profiling level of the long run (4 = what bench.py times at)
cg.set_profiling(level)
for b in range(nblocks):
    out = cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    if b == 1:
        cg.optimize_layout()
    if b % 100 == 0 or b == nblocks - 1:
        r = np.linalg.norm(out - out.mean(0), axis=1)
        print('block %4d  iterations %5d  ress %.6e  mean_dist %.4f  finite %s  extent %.2f' % (b, len(cg.tests), float(cg.ress[-1]), cg.mean_dist, bool(np.isfinite(out).all()), r.max()), flush=True)
print('%d iterations in %.2f s (%.3f ms each)' % (5 * nblocks, time.time() - t0, (time.time() - t0) / (5 * nblocks) * 1e3))
del cg
free0 = torch.cuda.mem_get_info()[0]
rng = np.random.default_rng(0)
for k in range(60):
    v, f = icosphere(int(rng.integers(2, 6)), 100.0)
    p = (v[rng.integers(0, v.shape[0], size=int(rng.integers(100, 200000)))] * rng.uniform(0.8, 1.1) + rng.normal(scale=5.0, size=(1, 3))).astype('f4')
    g = ShrinkwrapMeshConjGrad(TriMesh(v, f), p)
    g.search(p, lams=[10.0], num_iters=int(rng.integers(1, 4)), sigma_inv=0.1)
    del g
free1 = torch.cuda.mem_get_info()[0]
print('free device memory before / after 60 contexts: %.1f / %.1f MB (difference %.1f MB)' % (free0 / 2**20, free1 / 2**20, (free0 - free1) / 2**20))
