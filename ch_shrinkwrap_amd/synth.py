"""
Synthetic localization clouds and start meshes for the BASELINE.json configurations.

Own restatement of the few signed-distance primitives needed (sphere: /root/reference/ch_shrinkwrap/sdf.py:39-46,
capsule: sdf.py:60-80, smooth-min union: shape.py:369-376) plus a surface sampler that replaces
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


def sdf_gradient(sdf, p, eps=1e-3):
    g = np.empty_like(p)
    for k in range(3):
        d = np.zeros(3)
        d[k] = eps
        g[:, k] = (sdf(p + d[None, :]) - sdf(p - d[None, :])) / (2 * eps)
    n = np.linalg.norm(g, axis=1)
    n[n == 0] = 1
    return g / n[:, None]


def project_to_level(sdf, p, level=0.0, iters=8):
    p = np.array(p, 'f8')
    for _ in range(iters):
        d = sdf(p) - level
        p -= sdf_gradient(sdf, p) * d[:, None]
    return p


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


def sample_surface(sdf, verts, faces, n, sigma, seed, dtype='f4'):
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
    p = project_to_level(sdf, p, 0.0, iters=4)
    p += rng.normal(scale=sigma, size=p.shape)
    return p.astype(dtype)


def make_config(name, scale=1.0, seed=0):
    """BASELINE.json configs -> dict(points, sigma, mesh vertices, faces, lams, block, iters).
    `scale` < 1 shrinks N and the mesh resolution together (parity-test sizes)."""
    if name == 'c1':      # sphere R=100, 10k localizations, icosphere nsub=4 at 1.2 R, 20 iterations
        v, f = icosphere(4, 120.0)
        pts = sphere_cloud(10000, 100.0, 10.0, seed)
        return dict(points=pts, sigma=np.full(pts.shape, 10.0, 'f4'), vertices=v, faces=f, lams=[10.0], iters=20, block=20)
    if name == 'c2':      # capped tube r=50, L=1000 along y: 200k localizations, 39 692 vertices, 50 iterations in blocks of 5
        sdf = lambda p: sdf_capsule(p, (0, -500, 0), (0, 500, 0), 50.0)
        freq = max(4, int(round(63 * np.sqrt(scale))))
        n = int(200000 * scale)
        v0, f = star_mesh(sdf, freq, level=0.0)
        pts = sample_surface(sdf, v0, f, n, 10.0, seed)
        v, _ = star_mesh(sdf, freq, level=20.0)
        return dict(points=pts, sigma=np.full(pts.shape, 10.0, 'f4'), vertices=v, faces=f, lams=[10.0], iters=50, block=5)
    if name == 'c3':      # two-lobe vesicle (headline): 1M localizations, 198 812 vertices, remesh_frequency=5 -> blocks of 5
        sdf = sdf_two_lobe
        freq = max(4, int(round(141 * np.sqrt(scale))))
        n = int(1000000 * scale)
        v0, f = star_mesh(sdf, freq, level=0.0, relax=4)
        pts = sample_surface(sdf, v0, f, n, 10.0, seed)
        v, _ = star_mesh(sdf, freq, level=20.0, relax=4)
        return dict(points=pts, sigma=np.full(pts.shape, 10.0, 'f4'), vertices=v, faces=f, lams=[10.0], iters=5, block=5)
    raise ValueError(name)
