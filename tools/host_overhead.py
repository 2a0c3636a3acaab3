"""Developer tool: where does the host time of one search() block go?  (C3 workload, 5-iteration blocks)"""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ch_shrinkwrap_amd import synth, _lib as nw
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
c = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else 'c3', scale=1.0, seed=0)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
mesh = TriMesh(c['vertices'], c['faces'])
cg = ShrinkwrapMeshConjGrad(mesh, pts)
for _ in range(4):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.set_profiling(int(os.environ.get('NW_PROF', '2')))
T = dict(upload=0.0, search=0.0, logs=0.0, stage=0.0, finish=0.0, total=0.0)
lams_a = np.ascontiguousarray(c['lams'], dtype=np.float32)
n = 10
for _ in range(n):
    t0 = time.perf_counter()
    cg._upload_points(s, None)
    t1 = time.perf_counter()
    logs = (nw.IterLog * 5)(); lc = ctypes.c_int(0)
    cg._cache = {}
    cg._native.check(cg._L.nw_search(cg._h, nw.ptr(lams_a), 1, 5, 0, None, logs, ctypes.byref(lc)))
    t2 = time.perf_counter()
    cg._consume_logs(logs, lc.value)
    t3 = time.perf_counter()
    cg._accumulate_stage_ms()
    t4 = time.perf_counter()
    cg._finish()
    t5 = time.perf_counter()
    for k, a, b in (('upload', t0, t1), ('search', t1, t2), ('logs', t2, t3), ('stage', t3, t4), ('finish', t4, t5), ('total', t0, t5)):
        T[k] += (b - a) / n
dev = cg.stage_ms_total['total'][0] / n
print({k: round(v * 1e3, 3) for k, v in T.items()}, 'device ms per block', round(dev, 3))
