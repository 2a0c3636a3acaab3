"""Where a C3 block's wall time goes on the host: cProfile over 200 replayed blocks, by own time.  python3 tools/experiments/r05_block_profile.py"""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
c = synth.make_config('c3', scale=1.0, seed=0)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
cg = ShrinkwrapMeshConjGrad(TriMesh(c['vertices'], c['faces']), pts)
for _ in range(3):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.optimize_layout()
for _ in range(5):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.synchronize()
n = 200
t0 = time.perf_counter()
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
cg.synchronize()
pr.disable()
dt = time.perf_counter() - t0
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats('tottime').print_stats(12)
print('%.1f us per block (%.4f ms per step)' % (dt / n * 1e6, dt / n / 5 * 1e3))
print('\n'.join(l for l in out.getvalue().splitlines() if l.strip() and ('{' in l or '.py' in l or 'ncalls' in l))[:2500])
