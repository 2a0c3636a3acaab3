"""Fuzz of the device remesher: random surfaces (sphere, ellipsoid, bumpy sphere, open cap, the genus-2 network), noise, targets from 0.45 to
2.2 x the mean edge, 1-6 iterations, with and without relaxation, chained calls: every result must be a valid mesh of the same topology, and
a second run must give the same arrays.  python3 tools/experiments/r05_remesh_fuzz.py [cases] [seed]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import remesh as R
from ch_shrinkwrap_amd.trimesh import icosphere, TriMesh


def check(v, f, euler, n_bnd):
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    assert np.isfinite(v).all(), 'non-finite vertex'
    assert set(np.unique(cnt)) <= {1, 2}, 'an edge with %d faces' % cnt.max()
    assert int((cnt == 1).sum()) == n_bnd, 'boundary edges %d != %d' % ((cnt == 1).sum(), n_bnd)
    assert v.shape[0] - ue.shape[0] + f.shape[0] == euler, 'Euler characteristic %d != %d' % (v.shape[0] - ue.shape[0] + f.shape[0], euler)
    # every directed edge once (orientation kept)
    d = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]).astype(np.int64)
    key = d[:, 0] * (v.shape[0] + 1) + d[:, 1]
    assert np.unique(key).size == key.size, 'a directed edge twice'
    deg = np.bincount(f.ravel(), minlength=v.shape[0])
    assert deg.min() >= (1 if n_bnd else 3) and f.min() == 0        # (an open mesh may end with an ear at its frozen boundary) and f.max() == v.shape[0] - 1, 'degrees %d..%d, face ids %d..%d of %d vertices' % (deg.min(), deg.max(), f.min(), f.max(), v.shape[0])
    a = v[f[:, 1]] - v[f[:, 0]]; b = v[f[:, 2]] - v[f[:, 0]]
    return float(np.linalg.norm(np.cross(a, b), axis=1).min())


def topo(v, f):
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    ue, cnt = np.unique(e, axis=0, return_counts=True)
    return v.shape[0] - ue.shape[0] + f.shape[0], int((cnt == 1).sum())


n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
net = None
t0 = time.time()
for case in range(n_cases):
    kind = rng.choice(['sphere', 'ellipsoid', 'bumpy', 'cap', 'network'], p=[0.25, 0.25, 0.25, 0.15, 0.10])
    if kind == 'network':
        if net is None:
            from ch_shrinkwrap_amd import synth
            sdf = lambda p: 2.0 * synth.sdf_er_sim2(np.asarray(p, 'f8') * 0.5)
            net = synth._c4_start_mesh(sdf, 2.96 / np.sqrt(0.06))
        v, f = net
    else:
        v, f = icosphere(int(rng.integers(2, 6)), 100.0)
        if kind == 'ellipsoid':
            v = v * rng.uniform(0.4, 1.6, 3)
        if kind == 'bumpy':
            v = v * (1.0 + 0.25 * np.sin(v[:, :1] * rng.uniform(0.02, 0.1)) * np.cos(v[:, 1:2] * rng.uniform(0.02, 0.1)))
        if kind == 'cap':
            keep = v[f].mean(1)[:, 2] < rng.uniform(-20, 70)
            g = f[keep]; used = np.unique(g); remap = np.full(v.shape[0], -1, 'i4'); remap[used] = np.arange(used.size, dtype='i4')
            v, f = v[used], remap[g]
    v = np.ascontiguousarray(v, 'f4'); f = np.ascontiguousarray(f, 'i4')
    L0 = float(TriMesh(v, f)._mean_edge_length)
    v = (v + rng.normal(0, 0.08 * L0, v.shape)).astype('f4')
    euler, n_bnd = topo(v, f)
    for chain in range(int(rng.integers(1, 4))):
        rel = float(rng.uniform(0.45, 2.2)); it = int(rng.integers(1, 7)); relax = int(rng.choice([0, 0, 3, 10]))
        try:
            dv, df, st = R.remesh_device(v, f, it, rel * L0, 0.5, relax, return_stats=True)
            dv2, df2 = R.remesh_device(v, f, it, rel * L0, 0.5, relax)
        except RuntimeError as e:
            print('case %d (%s, %d vertices, target %.2f x, %d iterations, relax %d): %s' % (case, kind, v.shape[0], rel, it, relax, e)); raise
        assert np.array_equal(dv, dv2) and np.array_equal(df, df2), 'two runs differ'
        try:
            amin = check(dv, df, euler, n_bnd)
        except AssertionError as e:
            print('case %d (%s, %d vertices, target %.2f x, %d iterations, relax %d): %s' % (case, kind, v.shape[0], rel, it, relax, e)); raise
        print('case %2d.%d %-9s %7d -> %7d vertices, target %.2f x mean, %d iterations, relax %2d: ops %d / %d / %d, rounds %s, smallest face area %.2e' % (
            case, chain, kind, v.shape[0], dv.shape[0], rel, it, relax, st['n_split'], st['n_collapse'], st['n_flip'], st['rounds'], 0.5 * amin), flush=True)
        v, f = (dv + rng.normal(0, 0.03 * L0, dv.shape)).astype('f4'), df
print('%d cases in %.1f s: all valid, all reproducible' % (n_cases, time.time() - t0))
