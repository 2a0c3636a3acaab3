"""The host remesher at the full C4 scale with the driver's first targets (coarsening: the recipe's minimum edge length is above the start mesh's
edges): which path it takes and how long.  python3 tools/experiments/r05_host_remesh_full.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, remesh as R
from ch_shrinkwrap_amd.trimesh import TriMesh
c = synth.make_config('c4', scale=1.0, seed=0)
v, f = c['vertices'], c['faces']
L0 = float(TriMesh(v, f)._mean_edge_length)
print('start mesh %d vertices, mean edge %.3f' % (v.shape[0], L0), flush=True)
for rel in (1.13, 1.25, 0.99):
    os.environ['NWR_VERBOSE'] = '1'
    t0 = time.perf_counter()
    hv, hf = R.remesh(v, f, 5, rel * L0, 0.5, 0)
    th = time.perf_counter() - t0
    os.environ.pop('NWR_VERBOSE')
    t0 = time.perf_counter()
    dv, df = R.remesh_device(v, f, 5, rel * L0)
    td = time.perf_counter() - t0
    print('target %.2f x mean: host %.1f ms -> %d vertices; device %.1f ms -> %d vertices' % (rel, th * 1e3, hv.shape[0], td * 1e3, dv.shape[0]), flush=True)
