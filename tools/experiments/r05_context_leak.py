"""Does creating and destroying contexts leak device memory?  Fixed-size contexts (so that the allocator's caching does not blur the trend), free
memory after every hundred.  python3 tools/experiments/r05_context_leak.py"""
import os, sys, gc
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
from ch_shrinkwrap_amd import remesh as R
rng = np.random.default_rng(0)
v, f = icosphere(5, 100.0)
p = (v[rng.integers(0, v.shape[0], size=100000)] * 0.9 + rng.normal(scale=5.0, size=(100000, 3))).astype('f4')


def cycle(n):
    for k in range(n):
        g = ShrinkwrapMeshConjGrad(TriMesh(v, f), p)
        g.search(p, lams=[10.0], num_iters=2, sigma_inv=0.1)
        g.synchronize()
        del g
    gc.collect()


cycle(5)
R.remesh_device(v, f, 2, 8.0)
f0 = torch.cuda.mem_get_info()[0]
for rep in range(4):
    cycle(100)
    R.remesh_device(v, f, 2, 8.0)
    print('after %d contexts: free memory down by %.1f MB in all' % (100 * (rep + 1), (f0 - torch.cuda.mem_get_info()[0]) / 2**20), flush=True)
