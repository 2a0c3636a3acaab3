"""Five chained calls of the device remesher on the C4 network at fit_network.py 0.2's size, for a kernel trace:
rocprofv3 --kernel-trace --stats -d <dir> -- python3 tools/experiments/r05_remesh_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, remesh as R
from ch_shrinkwrap_amd.trimesh import TriMesh
c = synth.make_config('c4', scale=0.2, seed=0)
v, f = c['vertices'], c['faces']
L = float(TriMesh(v, f)._mean_edge_length) * 0.99
rng = np.random.default_rng(0)
for k in range(5):
    v, f, st = R.remesh_device((v + rng.normal(0, 0.1, v.shape)).astype('f4'), f, 5, L, return_stats=True)
    print(k, v.shape[0], st)
