"""Developer probe: distribution of the per-item duration of the nearest-face query (s_memtime ticks of the production kernel,
nw_debug what = 1) a few warm queries into a fit.  NW_ITEM_ORDER=0 keeps the timing on.  usage: python tools/nn_costs.py [config] [scale]"""
import os, sys, ctypes
os.environ.setdefault('NW_ITEM_TIMES', '1')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
c = synth.make_config(name, scale=scale, seed=0)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
mesh = TriMesh(c['vertices'].copy(), c['faces'])
cg = ShrinkwrapMeshConjGrad(mesh, pts)
for b in range(3):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    if b == 1:
        cg.optimize_layout()
cg.set_profiling(1)
for b in range(3):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
    ms, n = cg.stage_ms_total['nn']
    cg.set_profiling(1)
    cap = 400000
    items = np.zeros((cap, 2), np.int32)
    cost = np.zeros(2 * cap, np.uint32)
    n_items = ctypes.c_int(0)
    cg._native.check(cg._L.nw_debug(cg._h, 1, items.ctypes.data_as(ctypes.c_void_p), cost.ctypes.data_as(ctypes.c_void_p), cap, ctypes.byref(n_items)))
    k = n_items.value
    cst = cost[:k].astype(np.float64)
    print('block %d: nn %.1f us/query; %d items; item ticks: mean %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f; share of the total in the slowest 1%%: %.2f, slowest 10%%: %.2f' % (
        b, ms / max(n, 1) * 1e3, k, cst.mean(), np.percentile(cst, 50), np.percentile(cst, 90), np.percentile(cst, 99), np.percentile(cst, 99.9), cst.max(),
        np.sort(cst)[-max(k // 100, 1):].sum() / cst.sum(), np.sort(cst)[-max(k // 10, 1):].sum() / cst.sum()), flush=True)
    st = cost[cap:cap + k].astype(np.int64)
    st = (st - st.min()) & 0xffffffff
    en = st + cost[:k]
    span = en.max()
    # how many waves are running at a few points of the launch
    prof = []
    for fr in (0.1, 0.3, 0.5, 0.7, 0.8, 0.9, 0.95):
        t = fr * span
        prof.append('%d%%:%d' % (int(fr * 100), int(((st <= t) & (en > t)).sum())))
    order = np.argsort(st)
    print('         span %d ticks (10 ns) (sum of durations / span = %.0f waves on average); last start at %.2f of the span; running waves at %s' % (
        span, cst.sum() / span, st.max() / span, ' '.join(prof)), flush=True)
    late = np.argsort(en)[-5:]
    print('         last five to finish: ' + ', '.join('start %.2f dur %d' % (st[i] / span, cost[i]) for i in late), flush=True)
