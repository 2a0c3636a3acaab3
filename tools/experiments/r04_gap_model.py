"""Offline model of the RUNNER-UP GAP certificate: how many localizations could skip the nearest-face query?

A localization whose last query evaluated every centroid within d1 + g knows L = min(d2, d1 + g) <= the distance of every OTHER centroid.
If every centroid has moved by at most D since (sum of the iterations' largest vertex movements), its nearest face is unchanged while
d1 + D < L - D.  Runs the CPU oracle on a configuration in blocks of 5 (the bench's schedule, fixed topology) and reports per iteration the
largest vertex movement, the fraction of localizations that would skip (global D, and D taken per coarse cell of 32 nm + neighbours),
and -- for runs of 64 localizations in the order of their feet -- the fraction of runs in which every localization skips and the mean
number that do not.
  python tools/experiments/r04_gap_model.py [config] [scale] [blocks] [skin_nm]"""
import sys, os, time
import numpy as np
from scipy.spatial import cKDTree

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nanowrap_oracle as O          # noqa: E402
from ch_shrinkwrap_amd import synth              # noqa: E402
from ch_shrinkwrap_amd.trimesh import TriMesh    # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 5
skin = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
COARSE = 32.0

cfg = synth.make_config(name, scale=scale)
mesh = TriMesh(cfg['vertices'], cfg['faces'])
pts = cfg['points'].astype('f8')
s = 1.0 / cfg['sigma'].ravel()
nbr = mesh.neighbor_vertex_table()
faces = mesh.faces
N = pts.shape[0]
print('N %d M %d F %d, skin %.2f nm' % (N, mesh.vertices.shape[0], faces.shape[0], skin), flush=True)
lo = pts.min(0) - 64.0
cell_of = lambda x: np.floor((x - lo) / COARSE).astype(np.int64)
dims = cell_of(pts.max(0) + 64.0) + 1
pc = cell_of(pts)
pos = mesh.vertices.copy()
tests = []
# per-localization state of the certificate
d1_ub = np.full(N, np.inf)
L = np.zeros(N)
d1l_ub = np.full(N, np.inf)
Ll = np.zeros(N)
order = None
it = 0
for b in range(blocks):
    nrm = TriMesh(pos, faces).vertex_normals.copy()
    trace = []
    t0 = time.time()
    r = O.search(pos, nrm, nbr, faces, cfg['points'], cfg['lams'], 5, s, tests=tests, trace=trace)
    P = [pos] + [t['fnew'].reshape(-1, 3) for t in trace]          # positions before query k (k = 0..4) and after the block
    for k in range(5):
        cent = O.face_centroids(P[k], faces).astype('f8')
        if k > 0 or b > 0:
            mv = np.linalg.norm(P[k] - Pprev, axis=1)               # vertex movement of the update before this query
            D = float(mv.max())
            grid = np.zeros(dims)
            vc = cell_of(Pprev)
            np.maximum.at(grid, (vc[:, 0], vc[:, 1], vc[:, 2]), mv)
            dil = grid
            for ax in range(3):
                dil = np.maximum(dil, np.maximum(np.roll(dil, 1, ax), np.roll(dil, -1, ax)))
            Dl = dil[pc[:, 0], pc[:, 1], pc[:, 2]]
            d1_ub += D; L -= D
            d1l_ub += Dl; Ll -= Dl
        else:
            D = 0.0
        skip_g = d1_ub < L
        skip_l = d1l_ub < Ll
        dd, ii = cKDTree(cent).query(pts, k=2, workers=-1)
        if order is None and (b > 0 or k > 0):
            pass
        if order is None:
            # runs of 64 in the order of the feet (Morton code of the nearest centroid), as the library sorts them after its first query
            q = np.clip(((cent[ii[:, 0]] - lo) / 4.0).astype(np.int64), 0, 1023)
            def spread(v):
                v = (v | (v << 16)) & 0x030000FF; v = (v | (v << 8)) & 0x0300F00F; v = (v | (v << 4)) & 0x030C30C3; v = (v | (v << 2)) & 0x09249249
                return v
            order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind='stable')
        # localizations that did NOT skip ran the query: fresh bounds
        for sk, du, LL in ((skip_g, d1_ub, L), (skip_l, d1l_ub, Ll)):
            fresh = ~sk
            du[fresh] = dd[fresh, 0]
            LL[fresh] = np.minimum(dd[fresh, 1], dd[fresh, 0] + skin)
        nrun = N // 64
        runs_g = (~skip_g)[order][:nrun * 64].reshape(nrun, 64).sum(1)
        runs_l = (~skip_l)[order][:nrun * 64].reshape(nrun, 64).sum(1)
        # (a skipped localization whose face would have changed = the certificate is wrong: must never happen)
        prev_face_ok = True
        print('iteration %2d: largest movement %.3f nm | skip: global D %.3f (runs with nobody left %.3f, lanes left per run %.1f) | local D %.3f (runs %.3f, lanes left %.1f) | gap d2-d1 median %.2f nm' % (
            it, D, skip_g.mean(), (runs_g == 0).mean(), runs_g.mean(), skip_l.mean(), (runs_l == 0).mean(), runs_l.mean(), float(np.median(dd[:, 1] - dd[:, 0]))), flush=True)
        Pprev = P[k]
        it += 1
    # the update after the block's last query moves the mesh once more before the next block's first query
    pos = r.positions.copy()
    Pprev_after = P[5]
    print('block %d done (%.0f s)' % (b, time.time() - t0), flush=True)
    Pprev = P[4]
