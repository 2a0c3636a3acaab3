"""A/B inside ONE process (box-to-box and run-to-run spread of bench.py is +-5 %): C3, every block replayed as a graph, alternating
passes of 8 blocks with the vertex records written behind the block (NW_FLAG_ROWS_ASYNC) and with the synchronous write-back.
usage: python3 tools/experiments/r05_tail_ab.py [config] [scale]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ch_shrinkwrap_amd import synth, mesh_conj_grad
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = synth.make_config(name, scale=scale, seed=0)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
mesh = TriMesh(c['vertices'], c['faces'])
cg = ShrinkwrapMeshConjGrad(mesh, pts)


def blocks(n):
    for _ in range(n):
        cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)


blocks(2)
cg.optimize_layout()
blocks(3)
cg.synchronize()
res = {True: [], False: []}
for rep in range(8):
    for mode in (True, False):
        mesh_conj_grad._ROWS_ASYNC = mode
        blocks(2)
        cg.synchronize()
        t0 = time.perf_counter()
        blocks(8)
        cg.synchronize()
        res[mode].append((time.perf_counter() - t0) / 40 * 1e3)
for mode in (True, False):
    a = np.array(res[mode])
    print('%s rows_async=%d: ms per step median %.4f min %.4f max %.4f  (%s)' % (name, mode, np.median(a), a.min(), a.max(), ' '.join('%.4f' % x for x in a)))
