import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np
from ch_shrinkwrap_amd import remesh as R
from ch_shrinkwrap_amd.trimesh import icosphere, TriMesh
v, f = icosphere(4, 100.0)
L = float(TriMesh(v, f)._mean_edge_length) * 0.7
R.remesh_device(v, f, 5, L)
for k in range(3):
    sys.stderr.write('RUN %d\n' % k); sys.stderr.flush()
    dv, df = R.remesh_device(v, f, 5, L)
    sys.stderr.write('RESULT %d %d %d\n' % (dv.shape[0], df.shape[0], int(np.abs(dv).sum() * 1000)))
