"""
Fit quality in the reference's own terms: the evaluation recipes of the reference compare
a fitted mesh with the true object by

    PointsFromMesh           recipe_modules/surface_feature_extraction.py:76-105  -> evaluation_utils.points_from_mesh  (:35-150)
    AverageSquaredDistance   recipe_modules/surface_feature_extraction.py:107-138 -> evaluation_utils.average_squared_distance (:153-180)

i.e. a regular grid of points laid over every triangle of the mesh (spacing dx_min in the triangle's own plane) against a cloud of
points on the true surface, nearest-neighbour squared distances both ways: mse01, mse10 and mse_rms = sqrt((mse01 + mse10) / 2).
This module restates the two functions (paths relative to /root/reference/ch_shrinkwrap/) with the per-triangle loop vectorised; the
restatement is pinned by tests/golden/fit_quality.npz, produced by the reference's own functions (tests/golden/make_golden.py).
Host code: the metric is a block-boundary / end-of-fit diagnostic, not part of the iteration.
"""
import numpy as np
import scipy.spatial


def points_from_mesh(mesh, dx_min=5.0, p=1.0, rng=None):
    """evaluation_utils.points_from_mesh (:35-150) without the normals: every triangle gets the points of a regular grid in its own
    plane (axes e0 = the first edge, e1 = normal x e0; origin at the grid offset the reference uses) that fall inside it.
    `mesh` needs `_vertices['position']` and `faces` like the reference's.  p < 1 keeps a random share (the reference draws it with
    the global numpy state, `np.random.choice`: pass `rng` to reproduce a draw)."""
    tris = np.asarray(mesh._vertices['position'])[np.asarray(mesh.faces)]                       # (F, 3, 3)
    norms = np.cross(tris[:, 2, :] - tris[:, 1, :], tris[:, 0, :] - tris[:, 1, :])               # :56
    nn = np.linalg.norm(norms, axis=1)
    ok = nn != 0                                                                                 # :63 degenerate triangles are left out
    norms = norms[ok] / nn[ok, None]
    tris = tris[ok]
    v0 = tris[:, 1, :] - tris[:, 0, :]
    e0 = v0 / np.linalg.norm(v0, axis=1)[:, None]
    e1 = np.cross(norms, e0, axis=1)
    x0, y0 = (tris[:, 0, :] * e0).sum(1), (tris[:, 0, :] * e1).sum(1)                            # :82-87
    x1, y1 = (tris[:, 1, :] * e0).sum(1), (tris[:, 1, :] * e1).sum(1)
    x2, y2 = (tris[:, 2, :] * e0).sum(1), (tris[:, 2, :] * e1).sum(1)
    xs, ys = np.vstack([x0, x1, x2]).T, np.vstack([y0, y1, y2]).T
    xl, xu, yl, yu = xs.min(1), xs.max(1), ys.min(1), ys.max(1)
    with np.errstate(divide='ignore', invalid='ignore'):
        x1x0, x2x1, x0x2 = x1 - x0, x2 - x1, x0 - x2
        m0 = (y1 - y0) / x1x0
        m0[x1x0 == 0] = 0
        m1 = (y2 - y1) / x2x1
        m1[x2x1 == 0] = 0
        m2 = (y0 - y2) / x0x2
        m2[x0x2 == 0] = 0
    s1, s2 = np.sign(m1), np.sign(m2)
    # the grid of triangle i: x = arange(xl - x0 - dx/2, xu - x0, dx), y likewise (:117-118), coordinates relative to vertex 0
    xa, xb = xl - x0 - dx_min / 2, xu - x0
    ya, yb = yl - y0 - dx_min / 2, yu - y0
    nx = np.maximum(np.ceil((xb - xa) / dx_min), 0).astype(np.int64)                            # numpy.arange's length
    ny = np.maximum(np.ceil((yb - ya) / dx_min), 0).astype(np.int64)
    per = nx * ny
    tot = int(per.sum())
    if tot == 0:
        return np.zeros((0, 3), tris.dtype)
    t = np.repeat(np.arange(tris.shape[0]), per)                                                # triangle of every grid node
    k = np.arange(tot) - np.repeat(np.cumsum(per) - per, per)                                   # node number inside its grid (row-major: y outer)
    X = xa[t] + (k % nx[t]) * dx_min
    Y = ya[t] + (k // nx[t]) * dx_min
    inside = (Y > X * m0[t]) & (s1[t] * Y > s1[t] * (y1[t] - y0[t] + (X - x1[t] + x0[t]) * m1[t])) \
        & (s2[t] * Y < s2[t] * (y2[t] - y0[t] + (X - x2[t] + x0[t]) * m2[t]))                    # :123
    t, X, Y = t[inside], X[inside], Y[inside]
    d = X[:, None] * e0[t] + Y[:, None] * e1[t] + tris[t, 0, :]                                 # :126
    if p < 1.0:
        rng = np.random.default_rng() if rng is None else rng
        d = d[rng.choice(d.shape[0], size=int(p * d.shape[0]), replace=False)]
    return d


def average_squared_distance(points0, points1):
    """evaluation_utils.average_squared_distance (:153-180): (mean squared distance of points1 from their nearest neighbours in
    points0, the same of points0 from points1)."""
    e0, _ = scipy.spatial.cKDTree(points0).query(points1, k=1)
    e1, _ = scipy.spatial.cKDTree(points1).query(points0, k=1)
    return np.nansum(e0 ** 2) / len(e0), np.nansum(e1 ** 2) / len(e1)


def fit_quality(mesh, surface_points, dx_min=5.0):
    """What the reference's evaluation recipe records for a fit (recipe_modules/surface_feature_extraction.py:133-138):
    dict(mse01, mse10, mse_rms) between the grid points of the fitted mesh and points on the true surface (nm^2, nm^2, nm)."""
    m = points_from_mesh(mesh, dx_min=dx_min, p=1.0)
    mse0, mse1 = average_squared_distance(m, np.asarray(surface_points, m.dtype))
    return dict(mse01=float(mse0), mse10=float(mse1), mse_rms=float(np.sqrt((mse0 + mse1) / 2)), n_mesh_points=int(m.shape[0]))
