"""What a block of a FIT costs beyond its five iterations: a new optimiser on a new topology per block (cold query, set-up), as the driver runs them.
python3 tools/experiments/r05_block_cold.py [scale]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, remesh as R
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
c = synth.make_config('c4', scale=scale, seed=0)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
v, f = c['vertices'], c['faces']
L = float(TriMesh(v, f)._mean_edge_length)
nat = NativeContext(0)
for blk in range(6):
    mesh = TriMesh(v, f, vertex_normals=False, lazy_topology=True, all_referenced=True)
    t0 = time.perf_counter()
    cg = ShrinkwrapMeshConjGrad(mesh, pts, native=nat, reuse_device_mesh=True, device_tables=True, shield_sigma=L / 2)
    t1 = time.perf_counter()
    prof = blk >= 3
    if prof:
        cg.set_profiling(2)
    out = cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    cg.synchronize()
    t2 = time.perf_counter()
    extra = ''
    if prof:
        extra = ' | stages (ms, launches): ' + ', '.join('%s %.3f/%d' % (k, ms, n) for k, (ms, n) in cg.stage_ms_total.items() if n)
    print('block %d: optimiser %.2f ms, search %.2f ms%s' % (blk, (t1 - t0) * 1e3, (t2 - t1) * 1e3, extra), flush=True)
    v, f = R.remesh_device(out, f, 5, L * (0.99 - 0.01 * blk))
