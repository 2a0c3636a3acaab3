import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import remesh as R
from ch_shrinkwrap_amd.trimesh import icosphere, TriMesh
v, f = icosphere(3, 100.0)
L = float(TriMesh(v, f)._mean_edge_length) * 0.3
a = R.remesh_device(v, f, 5, L)
print('reference', a[0].shape, f.shape, flush=True)
os.environ['NW_REMESH_ROOM'] = sys.argv[1] if len(sys.argv) > 1 else '0.3'
b = R.remesh_device(v, f, 5, L)
print(b[0].shape, np.array_equal(a[0], b[0]))
