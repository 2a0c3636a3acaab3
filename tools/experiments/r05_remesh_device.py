"""The device remesher against the host one on a sphere, an ellipsoid above the partition size and the C4 network at fit_network's size:
validity (closed, genus, degrees), edge statistics, determinism, time.  python3 tools/experiments/r05_remesh_device.py [network scale]"""
import os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import remesh as R, synth
from ch_shrinkwrap_amd.trimesh import icosphere, TriMesh


def describe(v, f):
    m = TriMesh(v, f)
    closed = bool((m._halfedges['twin'] >= 0).all())
    e = np.linalg.norm(v[f] - v[np.roll(f, -1, 1)], axis=2)
    deg = np.bincount(f.ravel())
    return dict(nv=v.shape[0], nf=f.shape[0], closed=closed, euler=v.shape[0] - f.shape[0] // 2, mean=float(e.mean()), mn=float(e.min()), mx=float(e.max()),
                deg_max=int(deg.max()), deg_min=int(deg[deg > 0].min()), crc=(zlib.crc32(v.tobytes()), zlib.crc32(f.tobytes())))


cases = []
v, f = icosphere(4, 100.0)
cases.append(('sphere 2562, target 0.7 x mean', v, f, 0.7))
cases.append(('sphere 2562, target 1.6 x mean', v, f, 1.6))
v, f = icosphere(6, 100.0)
v = (v * np.array([1.0, 0.7, 1.4], 'f4')).astype('f4')
cases.append(('ellipsoid 40962, target 0.7 x mean', v, f, 0.7))
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 0.2
c = synth.make_config('c4', scale=scale, seed=0)
cases.append(('network %d, target 0.99 x mean' % c['vertices'].shape[0], c['vertices'], c['faces'], 0.99))
for name, v, f, rel in cases:
    L = float(TriMesh(v, f)._mean_edge_length) * rel
    t0 = time.perf_counter(); hv, hf = R.remesh(v, f, 5, L, 0.5, 0); th = time.perf_counter() - t0
    R.remesh_device(v, f, 5, L)
    t0 = time.perf_counter(); dv, df, st = R.remesh_device(v, f, 5, L, return_stats=True); td = time.perf_counter() - t0
    dv2, df2 = R.remesh_device(v, f, 5, L)
    same = np.array_equal(dv, dv2) and np.array_equal(df, df2)
    print('%s (L = %.3f)' % (name, L))
    print('   host   %7.1f ms: %s' % (th * 1e3, describe(hv, hf)))
    print('   device %7.1f ms: %s' % (td * 1e3, describe(dv, df)))
    print('          ops %d / %d / %d in %s rounds; mean edge (stats) %.4f, max degree %d; two runs identical: %s' % (
        st['n_split'], st['n_collapse'], st['n_flip'], st['rounds'], st['mean_edge_length'], st['max_valence'], same), flush=True)
    # chained: the device's own output, moved a little, again (what the fit does)
    if 'network' in name:
        rng = np.random.default_rng(0)
        for k in range(3):
            v2 = (dv + rng.normal(0, 0.1, dv.shape)).astype('f4')
            t0 = time.perf_counter(); hv, hf = R.remesh(v2, df, 5, L, 0.5, 0); th = time.perf_counter() - t0
            t0 = time.perf_counter(); dv, df, st = R.remesh_device(v2, df, 5, L, return_stats=True); td = time.perf_counter() - t0
            print('   chained call %d: device %.1f ms -> %s, ops %d / %d / %d, rounds %s (host on the same input: %.1f ms -> %d vertices)' % (
                k, td * 1e3, {k_: v_ for k_, v_ in describe(dv, df).items() if k_ != 'crc'}, st['n_split'], st['n_collapse'], st['n_flip'], st['rounds'], th * 1e3, hv.shape[0]), flush=True)
