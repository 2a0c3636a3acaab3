"""Does the device remesher breed coincident vertices on the needles of the full-scale mesh (the host remesher's pieces did)?  Needs
gpurun_out/full_block1.npz (tools/experiments/r05_dump_full_block.py) -- regenerated here if absent."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import remesh as R
p = 'gpurun_out/full_block1.npz'
if not os.path.exists(p):
    import subprocess
    subprocess.check_call([sys.executable, os.path.join(os.path.dirname(__file__), 'r05_dump_full_block.py')])
d = np.load(p)
v, f, L = d['v'], d['f'], 1.13 * float(d['L0'])
for it in (2, 5):
    rv, rf = R.remesh_device(v, f, it, L)
    u, cnt = np.unique(rv, axis=0, return_counts=True)
    e = np.linalg.norm(rv[rf].astype('f8') - rv[np.roll(rf, -1, 1)].astype('f8'), axis=2)
    a = rv[rf[:, 1]].astype('f8') - rv[rf[:, 0]].astype('f8'); b = rv[rf[:, 2]].astype('f8') - rv[rf[:, 0]].astype('f8')
    ar = 0.5 * np.linalg.norm(np.cross(a, b), axis=1)
    print('device, %d iterations: %d vertices, positions held by several vertices %d (max %d), edges <= 1e-6 L %d, shortest %.4f, faces of area < 1e-6 L^2: %d' % (
        it, rv.shape[0], int((cnt > 1).sum()), int(cnt.max()), int((e <= 1e-6 * L).sum()), e.min(), int((ar < 1e-6 * L * L).sum())), flush=True)
a = v[f[:, 1]].astype('f8') - v[f[:, 0]].astype('f8'); b = v[f[:, 2]].astype('f8') - v[f[:, 0]].astype('f8')
print('input: faces of area < 1e-6 L^2: %d' % int((0.5 * np.linalg.norm(np.cross(a, b), axis=1) < 1e-6 * L * L).sum()))
