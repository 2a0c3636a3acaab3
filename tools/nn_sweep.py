"""Developer tool: steady-state cost of the NN query against the cell size and the Morton block of the work list.
usage: python tools/nn_sweep.py <config> h[:block_cells[:nn_map]] ...      (runs on the GPU box; h = 0: the built-in rule)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

name = sys.argv[1]
c = synth.make_config(name, scale=float(os.environ.get('SWEEP_SCALE', '1.0')), seed=0)
pts = c['points']
s = 1.0 / c['sigma'].ravel()
v0, f = c['vertices'], c['faces']
print('%s: N=%d M=%d F=%d' % (name, pts.shape[0], v0.shape[0], f.shape[0]))
for spec in sys.argv[2:]:
    parts = spec.split(':')
    h = float(parts[0])
    if h > 0:
        os.environ['NW_CELL_SIZE'] = str(h)
    else:
        os.environ.pop('NW_CELL_SIZE', None)
    if len(parts) > 1 and parts[1]:
        os.environ['NW_ITEM_BLOCK_CELLS'] = parts[1]
    else:
        os.environ.pop('NW_ITEM_BLOCK_CELLS', None)
    if len(parts) > 2 and parts[2]:
        os.environ['NW_NN_MAP'] = parts[2]               # bit 1 warm start, 2 round-robin over XCDs, 4 interleaved runs
    else:
        os.environ.pop('NW_NN_MAP', None)
    mesh = TriMesh(v0.copy(), f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    cg.set_profiling(1)
    cg.search(pts, lams=c['lams'], num_iters=1, sigma_inv=s)
    cold = cg.stage_ms_total['nn']
    cg.set_profiling(0)
    for _ in range(3):
        cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    cg.set_profiling(1)
    t0 = time.perf_counter()
    for _ in range(2):
        cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    wall = (time.perf_counter() - t0) / 10
    nn = cg.stage_ms_total['nn']
    cg.set_profiling(2)
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    st = cg.stage_ms_total
    cg.nn_stats()
    cg.search(pts, lams=c['lams'], num_iters=1, sigma_inv=s)
    ns = cg.nn_stats()
    w = float(max(ns['items'], 1))
    print('   per wave: %.0f candidates, box rows %.0f, rows listed %.1f visited %.1f, cells tested %.1f visited %.1f, rounds %.2f; %d waves'
          % (ns['candidates'] / w, ns['box_rows'] / w, ns['rows_nonempty'] / w, ns['rows_visited'] / w, ns['cells_tested'] / w, ns['cells_visited'] / w,
             ns['rounds'] / w, ns['items']))
    print('   per wave: %.0f ticks in all (slowest wave %.0f), %.0f in the stream stage' % (16 * ns['wave_cycles_16'] / w, 16.0 * ns['max_wave_cycles_16'], 16 * ns['stream_cycles_16'] / w))
    print('%-10s wall %.4f ms/iter  nn %.4f (cold first query %.4f)  grid %.4f fixup %.4f attract %.4f  md %.2f' % (
        spec, wall * 1e3, nn[0] / nn[1], cold[0] / max(cold[1], 1), st['grid'][0] / st['grid'][1], st['fixup'][0] / max(st['fixup'][1], 1),
        st['attract'][0] / st['attract'][1], cg.mean_dist), flush=True)
    del cg
