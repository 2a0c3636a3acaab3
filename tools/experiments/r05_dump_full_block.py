import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
import numpy as np
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
c = synth.make_config('c4', scale=1.0, seed=0)
v, f = c['vertices'], c['faces']
pts, s = c['points'], 1.0 / c['sigma'].ravel()
cg = ShrinkwrapMeshConjGrad(TriMesh(v.copy(), f), pts)
out = cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s).copy()
np.savez_compressed('gpurun_out/full_block1.npz', v=out, f=f, L0=float(TriMesh(v, f)._mean_edge_length))
print('saved', out.shape, f.shape)
