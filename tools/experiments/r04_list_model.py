"""Offline model for per-localization candidate lists (round 4): how long would a list recorded at a block's first query stay
certifiable, and how long would it be?

Runs the CPU oracle on a configuration (C3 by default, full size) in blocks of 5 and, for every block, takes the centroids at the block's
first query as the record state: list_i(s) = centroids within d0_i + s of localization i.  For the block's later queries it reports the
fraction of localizations whose true nearest distance d_k satisfies d_k + D <= d0_i + s for D = the largest centroid displacement since
the record (global) and for D = the largest displacement in the 3x3x3 neighbourhood of coarse cells around the localization (local).

  python tools/experiments/r04_list_model.py [config] [scale] [blocks]
"""
import sys
import os
import time
import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nanowrap_oracle as O          # noqa: E402
from ch_shrinkwrap_amd import synth              # noqa: E402
from ch_shrinkwrap_amd.trimesh import TriMesh    # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 6
SKINS = (0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0)
COARSE = 32.0

cfg = synth.make_config(name, scale=scale)
mesh = TriMesh(cfg['vertices'], cfg['faces'])
pts = cfg['points']
s = 1.0 / cfg['sigma'].ravel()
nbr = mesh.neighbor_vertex_table()
faces = mesh.faces
print('N %d M %d F %d' % (pts.shape[0], mesh.vertices.shape[0], faces.shape[0]), flush=True)
tests = []
pos = mesh.vertices.copy()
lo = pts.min(0) - 64.0


def cell_of(x):
    return np.floor((x - lo) / COARSE).astype(np.int64)


dims = cell_of(pts.max(0) + 64.0) + 1
pc = cell_of(pts)

for b in range(blocks):
    mesh._vertices['position'][:] = pos
    mesh.invalidate() if hasattr(mesh, 'invalidate') else None
    nrm = TriMesh(pos, faces).vertex_normals.copy()
    trace = []
    t0 = time.time()
    r = O.search(pos, nrm, nbr, faces, pts, cfg['lams'], 5, s, tests=tests, trace=trace)
    # centroids the k-th query of the block saw: positions BEFORE the k-th update
    P = [pos] + [t['fnew'].reshape(-1, 3) for t in trace[:-1]]
    C = [O.face_centroids(p, faces).astype('f8') for p in P]
    D0 = trace[0]['dmean']
    print('block %d (%.0f s): mean d %.2f  p99 %.2f; vertex movement over the block: max %.2f mean %.3f' % (
        b, time.time() - t0, D0.mean(), np.percentile(D0, 99),
        np.linalg.norm(r.positions - pos, axis=1).max(), np.linalg.norm(r.positions - pos, axis=1).mean()), flush=True)
    tree = cKDTree(C[0])
    for sk in SKINS:
        cnt = tree.query_ball_point(pts.astype('f8'), D0 + sk, workers=-1, return_length=True)
        line = '  skin %.1f: list mean %.1f p50 %d p90 %d p99 %d max %d |' % (
            sk, cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 90), np.percentile(cnt, 99), cnt.max())
        for k in range(1, 5):
            disp = np.linalg.norm(C[k] - C[0], axis=1)
            dk = trace[k]['dmean']
            # local bound: max displacement of centroids per coarse cell, dilated by one cell
            cc = cell_of(C[0])
            grid = np.zeros(dims, 'f8')
            np.maximum.at(grid, (cc[:, 0], cc[:, 1], cc[:, 2]), disp)
            dil = grid.copy()
            for ax in range(3):
                a = dil
                dil = np.maximum(a, np.maximum(np.roll(a, 1, ax), np.roll(a, -1, ax)))
            dloc = dil[pc[:, 0], pc[:, 1], pc[:, 2]]
            okg = (dk + disp.max() <= D0 + sk).mean()
            okl = (dk + dloc <= D0 + sk).mean()
            line += ' k=%d Dmax %.2f glob %.4f loc %.4f |' % (k, disp.max(), okg, okl)
        print(line, flush=True)
    pos = r.positions.copy()
