"""Developer probe of the nearest-face query on the headline workload: per-iteration query time (HIP events, profiling level 1) and the
walk counters and phase shares of nw_debug (what = 0).
usage: python tools/nn_probe.py [config] [scale] [blocks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 8
c = synth.make_config(name, scale=scale, seed=0)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
mesh = TriMesh(c['vertices'].copy(), c['faces'])
cg = ShrinkwrapMeshConjGrad(mesh, pts)
cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.optimize_layout()
cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.set_profiling(1)
for b in range(blocks):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    ms, n = cg.stage_ms_total['nn']
    cg.set_profiling(1)
    print('block %d: nn %.1f us/query (production kernel)' % (b, ms / max(n, 1) * 1e3), flush=True)
cg.nn_stats()
cg.set_profiling(1)
for b in range(min(blocks, 3)):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    st = cg.nn_stats()
    ms, n = cg.stage_ms_total['nn']
    cg.set_profiling(1)
    items = max(st['items'], 1)
    wc = max(st['wave_cycles_16'], 1)
    print('block %d: nn %.1f us/query (counting kernel) | per wave: candidates %.0f, rows visited %.1f, cells tested %.1f; wave time %.0f ticks16: prologue %.2f, stream %.2f, tail %.2f, walk and the rest %.2f; slowest wave %.0f' % (
        b, ms / max(n, 1) * 1e3, st['candidates'] / (5.0 * items), st['rows_visited'] / (5.0 * items), st['cells_tested'] / (5.0 * items), wc / (5.0 * items),
        st['prologue_cycles_16'] / wc, st['stream_cycles_16'] / wc, st['tail_cycles_16'] / wc,
        1.0 - (st['prologue_cycles_16'] + st['stream_cycles_16'] + st['tail_cycles_16']) / wc, st['max_wave_cycles_16']), flush=True)
    print('         per wave: segments with centroids %.1f, of them a ball reaches %.1f; cells tested %.1f, reached %.1f; box rows %.1f, rounds %.2f' % (
        st.get('rows_nonempty', 0) / (5.0 * items), st.get('rows_visited', 0) / (5.0 * items), st['cells_tested'] / (5.0 * items), st.get('cells_visited', 0) / (5.0 * items),
        st.get('box_rows', 0) / (5.0 * items), st.get('rounds', 0) / (5.0 * items)), flush=True)
    print('         per LANE (its own ball against the cells): cells reached %.2f (largest lane of a wave %.1f), their candidates %.1f (largest lane %.1f)' % (
        st.get('lane_cells', 0) / (5.0 * items * 59.0), st.get('lane_cells_max', 0) / (5.0 * items), st.get('lane_candidates', 0) / (5.0 * items * 59.0),
        st.get('lane_candidates_max', 0) / (5.0 * items)), flush=True)
