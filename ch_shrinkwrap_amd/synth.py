"""
Synthetic localization clouds and start meshes for the BASELINE.json configurations.

Own restatement of the signed-distance primitives needed (sphere: /root/reference/ch_shrinkwrap/sdf.py:39-46,
capsule: sdf.py:60-80, round_box: sdf.py:250-269, sheet: sdf.py:271-292, smooth union / difference / rotation:
shape.py:369-376, 403-410, 446-480, the `ThreeWayJunction` and `ERSim2` networks: shape.py:252-261, 288-313; all
pinned against the reference's own values in tests/golden/sdf_shapes.npz), a sparse surface-nets mesher for
shapes that are not star-shaped (the reference obtains its start mesh from PYME's dual marching cubes, which is
not available), plus a surface sampler that replaces
`PYME.simulation.locify.points_from_sdf` (shape.py:75, PYME is not available): points are drawn uniformly on
the start mesh's faces, projected onto the zero level set by Newton steps along the SDF gradient, and jittered
with isotropic Gaussian localization error.  All float32, all seeded.
"""
import numpy as np

from .trimesh import icosphere, geodesic_sphere, TriMesh


def sphere_cloud(n, radius, sigma, seed, background=0.0, dtype='f4'):
    """C1: uniform directions * R + N(0, sigma^2) per axis."""
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    pts = d * radius + rng.normal(scale=sigma, size=(n, 3))
    nb = int(round(background * n))
    if nb:
        pts[:nb] = rng.uniform(-1.6 * radius, 1.6 * radius, size=(nb, 3))
    return pts.astype(dtype)


def _pmap(fn, items, workers=None):
    """thread-pool map (NumPy releases the GIL inside its loops); keeps order"""
    import os
    from concurrent.futures import ThreadPoolExecutor
    items = list(items)
    workers = workers or min(16, os.cpu_count() or 1)
    if workers <= 1 or len(items) <= 1:
        return [fn(x) for x in items]
    with ThreadPoolExecutor(workers) as ex:
        return list(ex.map(fn, items))


# ---- signed distance functions (points are (N,3) float64 arrays) ---------------------------------------------
def sdf_sphere(p, radius, center=(0, 0, 0)):
    return np.linalg.norm(p - np.asarray(center, 'f8')[None, :], axis=1) - radius


def sdf_capsule(p, a, b, radius):
    a = np.asarray(a, 'f8')
    b = np.asarray(b, 'f8')
    pa = p - a[None, :]
    ba = (b - a)[None, :]
    h = np.clip((pa * ba).sum(1) / (ba * ba).sum(), 0.0, 1.0)
    return np.linalg.norm(pa - ba * h[:, None], axis=1) - radius


def smooth_min(d1, d2, k):
    """polynomial smooth minimum used by the reference's UnionShape (shape.py:369-376)"""
    h = np.clip(0.5 + 0.5 * (d2 - d1) / k, 0.0, 1.0)
    return d2 * (1 - h) + d1 * h - k * h * (1 - h)


def sdf_two_lobe(p, radius=300.0, offset=250.0, k=50.0):
    """C3 headline shape: smooth union of two spheres at x = +-offset."""
    return smooth_min(sdf_sphere(p, radius, (-offset, 0, 0)), sdf_sphere(p, radius, (offset, 0, 0)), k)


def smooth_union(d0, d1, k):
    """UnionShape.sdf as the reference writes it (shape.py:369-376); algebraically the same polynomial as `smooth_min`."""
    res = np.minimum(d0, d1)
    if k > 0:
        h = np.maximum(k - np.abs(d0 - d1), 0.0)
        return res - h * h * 0.25 / k
    return res


def smooth_difference(d0, d1, k):
    """DifferenceShape(s0, s1).sdf: s1 with s0 carved out (shape.py:403-410)."""
    res = np.maximum(-d0, d1)
    if k > 0:
        h = np.maximum(k - np.abs(-d0 - d1), 0.0)
        return res + h * h * 0.25 / k
    return res


def sdf_round_box(p, halfwidth, r):
    q = np.abs(p) - np.asarray(halfwidth, 'f8')[None, :]
    return np.linalg.norm(np.maximum(q, 0.0), axis=1) + np.minimum(q.max(axis=1), 0.0) - r


def sdf_sheet(p, halfwidth, r):
    """box with a dumbbell (rounded, thickened) rim, sdf.py:271-292"""
    w = np.asarray(halfwidth, 'f8')
    q = np.abs(p) - w[None, :]
    rim = np.hypot(np.maximum(q[:, 0], q[:, 1]) + r, q[:, 2] + w[2]) - r
    return np.minimum(rim, q.max(axis=1))


def rotate_z_inverse(p, rz):
    """coordinates of p in the frame of a shape rotated by rz about z (RotationShape with rx = ry = 0, shape.py:446-480)"""
    # p @ inv(R).T = p @ R, written out element by element: a matrix product goes through BLAS, whose threaded kernels round the last
    # bits differently from host to host (and from run to run on a many-core one) -- the generator has to give the same mesh everywhere
    c, s = float(np.cos(rz)), float(np.sin(rz))
    p = np.asarray(p, 'f8')
    out = np.empty_like(p)
    out[:, 0] = p[:, 0] * c + p[:, 1] * s
    out[:, 1] = p[:, 1] * c - p[:, 0] * s
    out[:, 2] = p[:, 2]
    return out


def sdf_three_way_junction(p, h, r, k=0.0, centroid=(0.0, 0.0, 0.0)):
    """three tubes of length h, radius r meeting at `centroid` (shape.py:252-261): the two upper arms are joined with
    smoothing k, the lower arm with a sharp union."""
    c = np.asarray(centroid, 'f8')
    q = h / np.sqrt(2.0)
    upper = smooth_union(sdf_capsule(p, c, c + [-q, q, 0], r), sdf_capsule(p, c, c + [q, q, 0], r), k)
    return smooth_union(sdf_capsule(p, c, c + [0, -h, 0], r), upper, 0.0)


def sdf_er_sim2(p):
    """The reference's synthetic endoplasmic-reticulum network `ERSim2` (shape.py:288-313): three sheets and five tubes
    joined by smooth unions (k = 25 nm), with a 50 nm fenestration punched through the central sheet.  As in the
    reference, the `centroid` given to the third (rotated) sheet is discarded by RotationShape, so it sits at the origin."""
    sh = 100.0
    a, b = np.array([0.0, 0, 0]), np.array([400.0, -50, 0])
    c, d = np.array([500.0, 250, 0]), np.array([0.0, 240, 0])
    e, f = np.array([0.0, -600, 0]), np.array([-600.0, 0, 0])
    g, h = np.array([-40.0, 0, -100]), np.array([-40.0, 0, 100])
    k = sh / 4
    sheet0 = sdf_sheet(rotate_z_inverse(p, np.pi / 4), (226, 200, sh / 3), sh / 3)
    sheet1 = sdf_sheet(p - np.array([0.0, 133, 0])[None, :], (50, 50, sh / 3), 1)
    sheet2 = sdf_sheet(rotate_z_inverse(p, 7 * np.pi / 3), (33, 33, sh / 3), sh / 2)
    rt = sh // 2
    cap0, cap1, cap2 = sdf_capsule(p, a, b, rt), sdf_capsule(p, b, c, rt), sdf_capsule(p, c, d, rt)
    cap3, cap4, cap5 = sdf_capsule(p, a, e, rt), sdf_capsule(p, a, f, rt), sdf_capsule(p, g, h, 50)
    u = smooth_union(cap1, smooth_union(sheet2, cap2, k), k)
    u = smooth_union(sheet0, smooth_union(cap0, u, k), k)
    u = smooth_union(smooth_union(smooth_union(u, sheet1, k), cap3, k), cap4, k)
    return smooth_difference(cap5, u, k)


def _sheet_labels():
    """SHEET[cfg, e]: for the 8-bit inside/outside pattern `cfg` of a cell's corners (bit k = corner k inside,
    k = dz*4 + dy*2 + dx) and cube edge e = axis*4 + a + 2*b (a, b = offsets of the edge along the two other axes u, v with
    (u, v, axis) cyclic), the sheet the crossing on that edge belongs to (smallest edge index of its cycle), -1 if the edge
    is not crossed.  Crossings are linked face by face: a face with two crossed edges links them; an ambiguous face (four
    crossed edges, inside corners on a diagonal) links the two edges around each INSIDE corner -- a rule that depends only
    on the face's own corners, so the two cells sharing the face agree and every mesh edge is used exactly twice."""
    uv = [(1, 2), (2, 0), (0, 1)]
    ends = []
    for axis in range(3):
        ua, va = uv[axis]
        for b in range(2):
            for a_ in range(2):
                k0 = (a_ << ua) | (b << va)
                ends.append((k0, k0 | (1 << axis)))
    order = [[axis * 4 + a_ + 2 * b for b in range(2) for a_ in range(2)] for axis in range(3)]
    assert [e for ax in order for e in sorted(ax)] == list(range(12))
    ends = [ends[[axis * 4 + b * 2 + a_ for axis in range(3) for b in range(2) for a_ in range(2)].index(e)] for e in range(12)]
    # rebuild `ends` directly indexed by e = axis*4 + a + 2*b
    ends = []
    for e in range(12):
        axis, r = divmod(e, 4)
        a_, b = r & 1, r >> 1
        ua, va = uv[axis]
        k0 = (a_ << ua) | (b << va)
        ends.append((k0, k0 | (1 << axis)))
    tab = np.full((256, 12), -1, np.int64)
    for cfg in range(256):
        crossed = [((cfg >> k0) & 1) != ((cfg >> k1) & 1) for k0, k1 in ends]
        parent = list(range(12))

        def find(i):
            while parent[i] != i:
                i = parent[i]
            return i

        def link(i, j):
            ri, rj = find(i), find(j)
            if ri != rj:
                parent[max(ri, rj)] = min(ri, rj)
        for n in range(3):
            for side in range(2):
                fe = [e for e in range(12) if e // 4 != n and ((ends[e][0] >> n) & 1) == side and ((ends[e][1] >> n) & 1) == side]
                ce = [e for e in fe if crossed[e]]
                if len(ce) == 2:
                    link(ce[0], ce[1])
                elif len(ce) == 4:
                    for k in range(8):
                        if ((k >> n) & 1) == side and ((cfg >> k) & 1):
                            inc = [e for e in fe if k in ends[e]]
                            link(inc[0], inc[1])
        for e in range(12):
            if crossed[e]:
                tab[cfg, e] = find(e)
    return tab


_SHEET = _sheet_labels()


def isosurface_mesh(sdf, lo, hi, cell, level=0.0, block=32, lipschitz=1.5, slack=0.0, project=2):
    """Closed, oriented, MANIFOLD triangle mesh of `sdf == level` inside the box [lo, hi] by sparse surface nets.

    One quad per sign-changing grid edge (split along its shorter diagonal, outward orientation: sdf < level is inside),
    joining the vertices of the four cells around the edge.  A cell gets one vertex per SHEET that crosses it (`_sheet_labels`:
    the crossed edges of a cell are chained face by face into cycles), so two sheets that pass through the same cell (thin
    gaps, touching features) never share a vertex or an edge -- plain one-vertex-per-cell surface nets produce edges used
    four times there.  A vertex sits at the mean of its crossing
    points; `project` Newton steps pull it onto the level set.  Only blocks of `block`^3 cells whose centre is within
    lipschitz * half-diagonal + slack of the surface are evaluated.  Returns (vertices f4, faces i4)."""
    lo = np.asarray(lo, 'f8')
    hi = np.asarray(hi, 'f8')
    ncell = np.maximum(np.ceil((hi - lo) / cell).astype(np.int64), 1)
    NX, NY, NZ = (int(x) for x in ncell)
    nblk = (ncell + block - 1) // block
    bz, by, bx = np.meshgrid(np.arange(nblk[2]), np.arange(nblk[1]), np.arange(nblk[0]), indexing='ij')
    corners = np.stack([bx.ravel(), by.ravel(), bz.ravel()], 1) * block
    centres = lo[None, :] + (corners + 0.5 * block) * cell
    half_diag = 0.5 * (block + 2) * cell * np.sqrt(3.0)
    active = np.abs(sdf(centres) - level) <= lipschitz * half_diag + slack
    B = block
    ar = np.arange(-1, B + 1)                     # local node n <-> global node c0 - 1 + n, n in [0, B + 2)
    dims = np.array([NX, NY, NZ], np.int64)

    def do_block(c0):
        empty = (np.zeros(0, np.int64), np.zeros((0, 3)), np.zeros((0, 4), np.int64))
        gz, gy, gx = np.meshgrid(c0[2] + ar, c0[1] + ar, c0[0] + ar, indexing='ij')          # global node coordinates
        d = (sdf(lo[None, :] + np.stack([gx.ravel(), gy.ravel(), gz.ravel()], 1) * cell) - level).reshape(B + 2, B + 2, B + 2)
        ins = d < 0                                                                           # [z, y, x]
        if ins.all() or not ins.any():
            return empty
        # 8-bit corner pattern of every evaluated cell (local cell c <-> global cell c0 - 1 + c, c in [0, B + 1))
        cfg = np.zeros((B + 1, B + 1, B + 1), np.int64)
        for k in range(8):
            dz, dy, dx = (k >> 2) & 1, (k >> 1) & 1, k & 1
            cfg |= ins[dz:dz + B + 1, dy:dy + B + 1, dx:dx + B + 1].astype(np.int64) << k
        keys_acc, pos_acc, quads = [], [], []
        for axis in range(3):                       # 0: x edges, 1: y edges, 2: z edges; array axes are [z, y, x]
            ax = 2 - axis
            sl0 = [slice(None)] * 3
            sl1 = [slice(None)] * 3
            sl0[ax] = slice(0, B + 1)
            sl1[ax] = slice(1, B + 2)
            cross = ins[tuple(sl0)] != ins[tuple(sl1)]
            if not cross.any():
                continue
            ez, ey, ex = np.nonzero(cross)          # local coordinates of the LOWER node of each crossing edge
            low = [ex, ey, ez]
            d0 = d[tuple(sl0)][ez, ey, ex]
            d1 = d[tuple(sl1)][ez, ey, ex]
            t = d0 / (d0 - d1)
            inside0 = ins[tuple(sl0)][ez, ey, ex]
            p = np.stack([c0[0] - 1 + ex, c0[1] - 1 + ey, c0[2] - 1 + ez], 1).astype('f8')
            p[:, axis] += t                          # crossing point in global grid units
            ua, va = [(1, 2), (2, 0), (0, 1)][axis]  # the two other axes, (u, v, axis) cyclic -> u x v = axis
            gk = []
            okq = np.ones(ex.shape[0], bool)
            for a, b in ((1, 1), (0, 1), (0, 0), (1, 0)):            # counter-clockwise about +axis
                cl = [low[0].copy(), low[1].copy(), low[2].copy()]   # local cell coordinates (x, y, z)
                cl[ua] = cl[ua] - a
                cl[va] = cl[va] - b
                inb = (cl[ua] >= 0) & (cl[va] >= 0) & (cl[ua] <= B) & (cl[va] <= B) & (cl[axis] <= B)
                cx, cy, cz = (np.clip(c, 0, B) for c in cl)
                cf = cfg[cz, cy, cx]
                sheet = _SHEET[cf, axis * 4 + a + 2 * b]             # this grid edge is cube edge (axis, a, b) of that cell
                g = [c0[i] - 1 + cl[i] for i in range(3)]
                ing = inb & (g[0] >= 0) & (g[1] >= 0) & (g[2] >= 0) & (g[0] < NX) & (g[1] < NY) & (g[2] < NZ)
                key = ((g[2] * NY + g[1]) * NX + g[0]) * 16 + sheet
                gk.append(np.where(ing, key, -1))
                okq &= ing
                own_cell = ing & (cl[0] >= 1) & (cl[1] >= 1) & (cl[2] >= 1)      # cells [c0, c0 + B): this block places their vertices
                if own_cell.any():
                    keys_acc.append(key[own_cell])
                    pos_acc.append(p[own_cell])
            own_edge = (ex >= 1) & (ey >= 1) & (ez >= 1) & (low[axis] <= B) & (low[ua] <= B) & (low[va] <= B) & okq
            if own_edge.any():
                q = np.stack(gk, 1)[own_edge]
                q = np.where(inside0[own_edge][:, None], q, q[:, ::-1])          # inside at the lower node -> normal +axis
                quads.append(q)
        if not keys_acc:
            return empty
        k_all = np.concatenate(keys_acc)
        p_all = np.concatenate(pos_acc)
        uk, inv = np.unique(k_all, return_inverse=True)
        acc = np.zeros((uk.shape[0], 3))
        np.add.at(acc, inv, p_all)
        cnt = np.bincount(inv, minlength=uk.shape[0])
        return uk, lo[None, :] + (acc / cnt[:, None]) * cell, (np.concatenate(quads) if quads else np.zeros((0, 4), np.int64))

    res = _pmap(do_block, list(corners[active]))
    vid = np.concatenate([r[0] for r in res])
    if vid.size == 0:
        raise ValueError('isosurface_mesh: the level set does not cross the box')
    pos = np.concatenate([r[1] for r in res])
    q = np.concatenate([r[2] for r in res])
    order = np.argsort(vid)
    vid, pos = vid[order], pos[order]
    if (q < 0).any():
        raise RuntimeError('isosurface_mesh: the surface touches the box')
    qi = np.searchsorted(vid, q)
    if (qi >= vid.shape[0]).any() or (vid[np.minimum(qi, vid.shape[0] - 1)] != q).any():
        raise RuntimeError('isosurface_mesh: a surface cell was not evaluated (raise `lipschitz` / `slack`)')
    if project:
        pos0 = pos
        pos = project_to_level(sdf, pos, level, iters=project)
        # a vertex already sits inside a crossed cell: a projection that carries it further than two cells (or to a non-finite place:
        # a Newton step across a crease of the distance field) is undone
        off = np.abs(pos - pos0).max(1)
        bad = ~(off < 2.0 * cell)
        if bad.any():
            pos = np.where(bad[:, None], pos0, pos)
    d02 = np.linalg.norm(pos[qi[:, 0]] - pos[qi[:, 2]], axis=1)
    d13 = np.linalg.norm(pos[qi[:, 1]] - pos[qi[:, 3]], axis=1)
    s_ = (d02 <= d13)[:, None]
    t1 = np.where(s_, qi[:, [0, 1, 2]], qi[:, [0, 1, 3]])
    t2 = np.where(s_, qi[:, [0, 2, 3]], qi[:, [1, 2, 3]])
    return pos.astype('f4'), np.concatenate([t1, t2]).astype('i4')


def sdf_gradient(sdf, p, eps=1e-3):
    g = np.empty_like(p)
    for k in range(3):
        d = np.zeros(3)
        d[k] = eps
        g[:, k] = (sdf(p + d[None, :]) - sdf(p - d[None, :])) / (2 * eps)
    n = np.linalg.norm(g, axis=1)
    n[n == 0] = 1
    return g / n[:, None]


def project_to_level(sdf, p, level=0.0, iters=8, chunk=1 << 17):
    p = np.array(p, 'f8')

    def run(q):
        for _ in range(iters):
            d = sdf(q) - level
            q -= sdf_gradient(sdf, q) * d[:, None]
        return q
    if p.shape[0] <= chunk:
        return run(p)
    parts = _pmap(run, [p[i:i + chunk] for i in range(0, p.shape[0], chunk)])
    return np.concatenate(parts)


def star_mesh(sdf, freq, level=0.0, rmax=None, relax=10):
    """Closed genus-0 mesh of the level set `sdf == level` for a shape that is star-shaped about the origin:
    geodesic-sphere directions (frequency `freq`, 10 freq^2 + 2 vertices) are ray-marched to the level set (bisection), then tangentially relaxed (umbrella
    smoothing + re-projection) so that triangle sizes even out on elongated shapes."""
    v, f = geodesic_sphere(freq, 1.0, dtype='f8')
    if rmax is None:
        rmax = 1.0
        while (sdf(v * rmax) - level).min() < 0:
            rmax *= 2
    lo = np.zeros(v.shape[0])
    hi = np.full(v.shape[0], float(rmax))
    for _ in range(48):
        mid = 0.5 * (lo + hi)
        inside = (sdf(v * mid[:, None]) - level) < 0
        lo = np.where(inside, mid, lo)
        hi = np.where(inside, hi, mid)
    p = v * (0.5 * (lo + hi))[:, None]
    if relax:
        mesh = TriMesh(p.astype('f4'), f)
        nb = mesh.neighbor_vertex_table()
        mask = nb >= 0
        cnt = np.maximum(mask.sum(1), 1)
        for _ in range(relax):
            c = (p[np.where(mask, nb, 0)] * mask[:, :, None]).sum(1) / cnt[:, None]
            p = project_to_level(sdf, 0.5 * (p + c), level, iters=3)
    return p.astype('f4'), f


def sample_surface(sdf, verts, faces, n, sigma, seed, dtype='f4', iters=4):
    """n localizations: area-weighted uniform samples on the faces of a mesh of the zero level set, projected
    onto the level set, plus N(0, sigma^2) per axis."""
    rng = np.random.default_rng(seed)
    v = np.asarray(verts, 'f8')
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    fi = rng.choice(faces.shape[0], size=n, p=area / area.sum())
    r1 = np.sqrt(rng.random(n))
    r2 = rng.random(n)
    p = (1 - r1)[:, None] * a[fi] + (r1 * (1 - r2))[:, None] * b[fi] + (r1 * r2)[:, None] * c[fi]
    p = project_to_level(sdf, p, 0.0, iters=iters)
    p += rng.normal(scale=sigma, size=p.shape)
    return p.astype(dtype)


def _c4_start_mesh(sdf, cell):
    """Start mesh of config C4: sparse surface nets at +20 nm, then three passes of the isotropic remesher at the mesh's own mean
    edge length (surface nets leave slivers where neighbouring cell vertices project to almost the same point; the real pipeline
    remeshes its isosurface as well).  Every step is fixed-order arithmetic (no BLAS: rotate_z_inverse), so the mesh is the same
    on every host; the result is still checked -- a remesh at the mean edge length changes the face count by ~20 %."""
    from . import remesh as _remesh
    v, f = isosurface_mesh(sdf, (-1500, -1500, -420), (1400, 900, 420), cell, level=20.0, slack=60.0)
    e = np.concatenate([v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 1]], v[f[:, 0]] - v[f[:, 2]]]).astype('f8')
    target = float(np.sqrt((e * e).sum(1)).mean())
    v2, f2 = _remesh.remesh(v, f, 3, target, 0.5, 0, serial=True)      # (serial: the benchmark's start mesh must not change with the library's threading)
    if not 0.5 * f.shape[0] < f2.shape[0] < 2 * f.shape[0]:
        raise RuntimeError('synth c4: remeshing the start surface at its mean edge length %.3f turned %d faces into %d' % (target, f.shape[0], f2.shape[0]))
    return v2, f2


C5_LATTICE = np.array([1400.0, 900.0, 900.0], 'f4')     # pitch of the 2x2x2 lattice of vesicles in BASELINE configs[4] (nm)


def make_config(name, scale=1.0, seed=0):
    """BASELINE.json configs -> dict(points, sigma, mesh vertices, faces, lams, block, iters).
    `scale` < 1 shrinks N and the mesh resolution together (parity-test sizes)."""
    if name == 'c1':      # sphere R=100, 10k localizations, icosphere nsub=4 at 1.2 R, 20 iterations
        v, f = icosphere(4, 120.0)
        pts = sphere_cloud(10000, 100.0, 10.0, seed)
        return dict(points=pts, sigma=np.full(pts.shape, 10.0, 'f4'), vertices=v, faces=f, lams=[10.0], iters=20, block=20,
                    sdf=lambda p: sdf_sphere(p, 100.0), surface=(icosphere(4, 100.0)[0], f))
    if name == 'c2':      # capped tube r=50, L=1000 along y: 200k localizations, 39 692 vertices, 50 iterations in blocks of 5
        sdf = lambda p: sdf_capsule(p, (0, -500, 0), (0, 500, 0), 50.0)
        freq = max(4, int(round(63 * np.sqrt(scale))))
        n = int(200000 * scale)
        v0, f = star_mesh(sdf, freq, level=0.0)
        pts = sample_surface(sdf, v0, f, n, 10.0, seed)
        v, _ = star_mesh(sdf, freq, level=20.0)
        return dict(points=pts, sigma=np.full(pts.shape, 10.0, 'f4'), vertices=v, faces=f, lams=[10.0], iters=50, block=5, sdf=sdf, surface=(v0, f))
    if name == 'c3':      # two-lobe vesicle (headline): 1M localizations, 198 812 vertices, remesh_frequency=5 -> blocks of 5
        sdf = sdf_two_lobe
        freq = max(4, int(round(141 * np.sqrt(scale))))
        n = int(1000000 * scale)
        v0, f = star_mesh(sdf, freq, level=0.0, relax=4)
        pts = sample_surface(sdf, v0, f, n, 10.0, seed)
        v, _ = star_mesh(sdf, freq, level=20.0, relax=4)
        return dict(points=pts, sigma=np.full(pts.shape, 10.0, 'f4'), vertices=v, faces=f, lams=[10.0], iters=5, block=5, sdf=sdf, surface=(v0, f))
    if name == 'c4':      # ER-like tube/sheet network with a fenestration (ERSim2, twice life size): 5M localizations, ~800k vertices
        sdf = lambda p: 2.0 * sdf_er_sim2(np.asarray(p, 'f8') * 0.5)
        n = int(5000000 * scale)
        cell = 2.96 / np.sqrt(scale)
        v0, f = _c4_start_mesh(sdf, cell)
        v = project_to_level(sdf, v0, 20.0, iters=2).astype('f4')
        pts = sample_surface(sdf, v, f, n, 10.0, seed, iters=6)
        return dict(points=pts, sigma=np.full(pts.shape, 10.0, 'f4'), vertices=v, faces=f, lams=[10.0], iters=5, block=5, sdf=sdf, surface=(v0, f))
    if name == 'c5':      # BASELINE configs[4] as ONE scene: 8 two-lobe vesicles on a 2x2x2 lattice (the scene 8 ranks share out, tile = vesicle)
        parts = [make_config('c3', scale=scale, seed=seed + k) for k in range(8)]
        nv = parts[0]['vertices'].shape[0]
        off = [np.array([(k & 1), (k >> 1) & 1, (k >> 2) & 1], 'f4') * C5_LATTICE for k in range(8)]
        return dict(points=np.concatenate([p['points'] + o[None, :] for p, o in zip(parts, off)]).astype('f4'),
                    sigma=np.concatenate([p['sigma'] for p in parts]),
                    vertices=np.concatenate([p['vertices'] + o[None, :] for p, o in zip(parts, off)]).astype('f4'),
                    faces=np.concatenate([p['faces'] + k * nv for k, p in enumerate(parts)]).astype(parts[0]['faces'].dtype),
                    lams=[10.0], iters=5, block=5)
    raise ValueError(name)


def truth_cloud(cfg, density=0.04, seed=12345):
    """Points ON the true surface of a configuration (no localization error), `density` per nm^2: what the reference's evaluation recipe
    compares a fitted mesh with (`PointcloudFromShape(no_jitter=True, p=1.0)`, test_evaluation_recipe.yaml; the sampler itself is PYME's,
    this is the generator's own: area-weighted samples of the zero-level mesh projected onto the level set)."""
    v0, f = cfg['surface']
    v = np.asarray(v0, 'f8')
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    area = float((0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)).sum())
    n = max(1000, int(area * density))
    return sample_surface(cfg['sdf'], v0, f, n, 0.0, seed, iters=6)
