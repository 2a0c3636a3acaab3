"""The crafted needles of tests/test_remesh.py (faces of no area, long edges) through the device remesher: positions held by several vertices?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd.trimesh import icosphere
from ch_shrinkwrap_amd import remesh as R
v, f = icosphere(6, 100.0)
v = (v * np.array([1.0, 0.7, 1.4], 'f4')).astype('f4')
rng = np.random.default_rng(3)
taken = np.zeros(v.shape[0], bool); n = 0
for face in rng.permutation(f.shape[0]):
    a, b, c = f[face]
    if taken[[a, b, c]].any():
        continue
    v[c] = 0.5 * (v[a] + v[b]); taken[[a, b, c]] = True; n += 1
    if n == 1000:
        break
for it in (1, 3, 5):
    ov, of = R.remesh_device(v, f, it, 1.1)
    u, cnt = np.unique(ov, axis=0, return_counts=True)
    e = np.sort(np.concatenate([of[:, [0, 1]], of[:, [1, 2]], of[:, [2, 0]]]), 1)
    _, ec = np.unique(e, axis=0, return_counts=True)
    el = np.linalg.norm(ov[of].astype('f8') - ov[np.roll(of, -1, 1)].astype('f8'), axis=2)
    print('%d iterations: %d vertices, positions held twice %d, closed %s, edges of no length %d, shortest %.2e' % (it, ov.shape[0], int((cnt > 1).sum()), bool((ec == 2).all()), int((el <= 1e-6).sum()), el.min()))
hv, hf = R.remesh(v, f, 5, 1.1, 0.5, 0)
u, cnt = np.unique(hv, axis=0, return_counts=True)
print('host: %d vertices, positions held twice %d' % (hv.shape[0], int((cnt > 1).sum())))
