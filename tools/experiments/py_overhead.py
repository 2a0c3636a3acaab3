"""Python-side cost of a search() block (cProfile over 400 blocks of 5 on C1, where the host matters most), and block times after idle
pauses.  NOTE: after a few hundred blocks the fit has converged and the device-side stop condition ends blocks early (their kernels
return at once): the wall times printed here are those of such blocks -- the Python share is what this tool is for."""
import sys, time, cProfile, pstats
sys.path.insert(0, '.')
import numpy as np
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
cfg = synth.make_config(sys.argv[1] if len(sys.argv) > 1 else 'c1', seed=0)
mesh = TriMesh(cfg['vertices'], cfg['faces'])
pts, s = cfg['points'], 1.0 / cfg['sigma'].ravel()
cg = ShrinkwrapMeshConjGrad(mesh, pts)
for _ in range(4):
    cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
cg.optimize_layout()
for _ in range(4):
    cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
n = 400
t0 = time.perf_counter()
for _ in range(n):
    cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
dt = time.perf_counter() - t0
print('block of 5: %.1f us wall' % (dt / n * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(14)

# block-by-block after an idle pause: does the first block after a pause cost more?
for pause in (0.0, 0.002, 0.05):
    time.sleep(pause)
    ts = []
    for _ in range(8):
        t = time.perf_counter()
        cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
        ts.append((time.perf_counter() - t) * 1e6)
    print('after %.0f ms idle: blocks of 5 take' % (pause * 1e3), ' '.join('%.0f' % x for x in ts), 'us')
cg.set_profiling(4)
for _ in range(3):
    cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
ts = []
for _ in range(8):
    t = time.perf_counter()
    cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
    ts.append((time.perf_counter() - t) * 1e6)
print('profiling level 4: blocks of 5 take', ' '.join('%.0f' % x for x in ts), 'us')
