"""
Block-boundary isotropic remesher (include/nw_remesh.h, ch_shrinkwrap_amd/csrc/remesh.cpp).
PYME's TriangleMesh.remesh -- what the reference calls at `_membrane_mesh.pyx:1546` -- is not in the reference tree, so
there is nothing to pin against; these tests check the properties the optimiser relies on: the result is a closed oriented
2-manifold of the same genus, edge lengths gather around the target, vertex degrees around six and within the
neighbour-table width, the surface stays where it was.  CPU only.
"""
import os
import re
import ctypes

import numpy as np
import pytest

from ch_shrinkwrap_amd import remesh as R
from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _edges(f):
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    return np.unique(e, axis=0, return_counts=True)


def _volume(v, f):
    a, b, c = (v[f[:, k]].astype('f8') for k in range(3))
    return np.einsum('ij,ij->i', a, np.cross(b, c)).sum() / 6


def test_library_exports_header_symbols():
    from ch_shrinkwrap_amd import build
    build.build_host_library()
    txt = re.sub(r'/\*.*?\*/', '', open(os.path.join(ROOT, 'include', 'nw_remesh.h')).read(), flags=re.S)
    names = sorted(set(re.findall(r'\b(nwr_[a-z_]+)\s*\(', txt)))
    L = ctypes.CDLL(R.LIB_PATH)
    assert names == sorted(R.SYMBOLS)
    for n in names:
        assert hasattr(L, n)
    assert R.load().nwr_abi_version() == 4
    assert ctypes.sizeof(R.Stats) == 3 * 8 + 8 + 2 * 4


@pytest.mark.parametrize('target', [30.0, 12.0, 6.0])
def test_sphere_to_target_edge_length(target):
    v, f = icosphere(3, 100.0)                      # 642 vertices, edges ~15
    nv, nf, st = R.remesh(v, f, 5, target, 0.5, 0, return_stats=True)
    ue, cn = _edges(nf)
    assert (cn == 2).all()                                               # closed 2-manifold
    assert nv.shape[0] - ue.shape[0] + nf.shape[0] == 2                  # still a sphere
    assert 0.95 < _volume(nv, nf) / (4 / 3 * np.pi * 1e6) <= 1.0         # outward, vertices stay on / inside the sphere (chords)
    L = np.linalg.norm(nv[ue[:, 0]] - nv[ue[:, 1]], axis=1)
    assert 0.85 * target < L.mean() < 1.15 * target
    assert L.min() > 0.45 * target and L.max() < 2.0 * target
    assert np.mean((L > 0.8 * target) & (L < 4 / 3 * target)) > 0.6     # without relaxation many edges sit just below 4/5 L (blocked collapses)
    deg = np.bincount(ue.ravel(), minlength=nv.shape[0])
    assert deg.min() >= 3 and deg.max() <= 9 and np.mean(np.abs(deg - 6) <= 1) > 0.9
    assert st['max_valence'] == deg.max() and abs(st['mean_edge_length'] - L.mean()) < 1e-3 * target
    assert np.linalg.norm(nv, axis=1).max() <= 100.0 + 1e-3
    TriMesh(nv, nf)                                                      # the optimiser's substrate accepts it
    # n = 0 is the identity (up to dropping unreferenced vertices)
    zv, zf = R.remesh(v, f, 0, target, 0.5, 0)
    assert np.array_equal(zv, v) and np.array_equal(zf, f)


def test_relaxation_evens_out_edge_lengths_and_keeps_the_surface():
    v, f = icosphere(3, 100.0)
    a_v, a_f = R.remesh(v, f, 5, 12.0, 0.5, 0)
    b_v, b_f = R.remesh(v, f, 5, 12.0, 0.5, 10)
    sd = []
    for xv, xf in ((a_v, a_f), (b_v, b_f)):
        ue, _ = _edges(xf)
        L = np.linalg.norm(xv[ue[:, 0]] - xv[ue[:, 1]], axis=1)
        sd.append(L.std() / L.mean())
    assert sd[1] < sd[0]
    r = np.linalg.norm(b_v, axis=1)
    assert r.min() > 98.5 and r.max() < 100.5                            # tangential moves only


def test_genus_two_network_and_degenerate_input():
    from ch_shrinkwrap_amd import synth
    sdf = lambda p: 2.0 * synth.sdf_er_sim2(np.asarray(p, 'f8') * 0.5)
    v, f = synth.isosurface_mesh(sdf, (-1500, -1500, -420), (1400, 900, 420), 19.0, level=20.0, slack=60.0)
    ue, cn = _edges(f)
    assert (cn == 2).all() and v.shape[0] - ue.shape[0] + f.shape[0] == -2
    L0 = np.linalg.norm(v[ue[:, 0]] - v[ue[:, 1]], axis=1)
    assert L0.min() < 0.05 * L0.mean()                                   # surface nets leave slivers
    nv, nf = R.remesh(v, f, 3, -1, 0.5, 0)                               # target = mean edge length of the input
    ue, cn = _edges(nf)
    assert (cn == 2).all() and nv.shape[0] - ue.shape[0] + nf.shape[0] == -2
    L = np.linalg.norm(nv[ue[:, 0]] - nv[ue[:, 1]], axis=1)
    assert L.min() > 0.2 * L.mean()
    assert np.abs(sdf(nv) - 20.0).max() < 6.0                            # new midpoints lie on chords of the old surface
    # errors, not crashes
    with pytest.raises(RuntimeError):
        R.remesh(v, np.concatenate([f, f[:1]]), 1, -1, 0.5, 0)           # the same oriented face twice
    with pytest.raises(ValueError):
        R.remesh(v[:, :2], f)
    # an open patch: boundary vertices are left where they are
    keep = v[f].mean(1)[:, 2] > 0
    pv, pf = R.remesh(v, f[keep], 2, -1, 0.5, 0)
    ue, cn = _edges(pf)
    assert set(np.unique(cn)) <= {1, 2} and (cn == 1).any()


@pytest.mark.parametrize('case', ['icosphere', 'network', 'open_with_spare_slots', 'large_on_all_cores'])
def test_native_geometry_refresh_is_bit_identical_to_the_numpy_definition(case, monkeypatch):
    """nwr_mesh_geometry / nwr_halfedge_twins against the NumPy code they replace in trimesh.TriMesh (the definition the
    optimiser's golden inputs were produced with): face normals, areas, half-edge lengths, vertex normals, twins."""
    from ch_shrinkwrap_amd import trimesh as T
    if case == 'network':
        from ch_shrinkwrap_amd import synth
        c = synth.make_config('c4', scale=0.02, seed=5)
        v, f, extra = c['vertices'], c['faces'], 0
    else:
        # 'large_on_all_cores': 327 680 faces -- the native loops run on several threads (every thread its own range of faces / vertices,
        # the vertex normals summed in the serial order inside each range)
        v, f = icosphere(7 if case == 'large_on_all_cores' else 3, 73.0)
        if case == 'large_on_all_cores':
            monkeypatch.setenv('NW_REMESH_THREADS', '8')
        v = (v * np.array([1.0, 0.6, 1.7], 'f4') + np.array([1e3, -2e3, 5e2], 'f4')).astype('f4')
        extra = 0
        if case == 'open_with_spare_slots':
            f = f[v[f].mean(1)[:, 2] > 5e2]
            extra = 11
    a = TriMesh(v, f, max_vertices=v.shape[0] + extra)
    b = TriMesh(v, f, max_vertices=v.shape[0] + extra)
    b._numpy_geometry = True
    b.update_geometry()
    for name in ('normal', 'area'):
        assert np.array_equal(a._faces[name], b._faces[name]), name
    assert np.array_equal(a._halfedges['length'], b._halfedges['length'])
    assert np.array_equal(a._vertices['normal'], b._vertices['normal'])
    # the linear-time pairing against the sort-based one
    nv = np.int64(a._vertices.shape[0])
    origin, dest = a._origin.astype('i8'), a._halfedges['vertex'].astype('i8')
    key, rkey = origin * nv + dest, dest * nv + origin
    order = np.argsort(key, kind='stable')
    posn = np.minimum(np.searchsorted(key[order], rkey), key.shape[0] - 1)
    cand = order[posn]
    assert np.array_equal(a._halfedges['twin'], np.where(key[cand] == rkey, cand, -1))


@pytest.mark.parametrize('case', ['icosphere', 'network', 'open_with_spare_slots', 'remeshed', 'large_on_all_cores'])
def test_native_topology_is_identical_to_the_numpy_definition(case, monkeypatch):
    """nwr_build_topology / nwr_ring_tables against the NumPy code they replace (trimesh._build_halfedges, TriMesh._build_rings,
    neighbor_vertex_table, MembraneMesh._neighbor_tables): every half-edge field, ring start, ring order, valence, and the per-slot
    tables -- closed, open (boundary fans) and freshly remeshed (ids in creation order) meshes, with unused vertex slots."""
    from ch_shrinkwrap_amd.membrane_mesh import MembraneMesh
    extra = 0
    if case == 'network':
        from ch_shrinkwrap_amd import synth
        c = synth.make_config('c4', scale=0.02, seed=5)
        v, f = c['vertices'], c['faces']
    else:
        v, f = icosphere(7 if case == 'large_on_all_cores' else 3, 73.0)
        if case == 'large_on_all_cores':
            monkeypatch.setenv('NW_REMESH_THREADS', '8')
        v = (v * np.array([1.0, 0.6, 1.7], 'f4')).astype('f4')
        if case == 'open_with_spare_slots':
            f = f[v[f].mean(1)[:, 2] > 0]
            extra = 11
        if case == 'remeshed':
            v, f = R.remesh(v, f, 3, 6.0, 0.5, 2)
    a = MembraneMesh(vertices=v, faces=f) if extra == 0 else TriMesh(v, f, max_vertices=v.shape[0] + extra)
    TriMesh._numpy_topology = True
    try:
        b = MembraneMesh(vertices=v, faces=f) if extra == 0 else TriMesh(v, f, max_vertices=v.shape[0] + extra)
        nvt_b = b.neighbor_vertex_table()
        tabs_b = b._neighbor_tables() if extra == 0 else None
    finally:
        TriMesh._numpy_topology = False
    for name in ('vertex', 'face', 'twin', 'next', 'prev', 'length', 'component'):
        assert np.array_equal(a._halfedges[name], b._halfedges[name]), name
    assert np.array_equal(a._origin, b._origin)
    for name in ('halfedge', 'valence', 'neighbors', 'component', 'locally_manifold', 'normal'):
        assert np.array_equal(a._vertices[name], b._vertices[name]), name
    assert np.array_equal(a.neighbor_vertex_table(), nvt_b)
    if tabs_b is not None:
        nxt, area = a._neighbor_tables()
        assert np.array_equal(nxt, tabs_b[0]) and np.array_equal(area, tabs_b[1])


def test_degenerate_input_is_refused_not_refined_for_ever():
    """A non-finite vertex (or one flung far away) has edges that stay too long however often they are split: the remesher must say
    so instead of doubling the mesh in every pass (seen as a ten-fold face count / a generator that did not come back)."""
    import time
    v, f = icosphere(3, 50.0)
    bad = v.copy()
    bad[7, 1] = np.inf
    with pytest.raises(RuntimeError):
        R.remesh(bad, f, 5, -1, 0.5, 0)
    far = v.copy()
    far[7] *= 1e7
    t0 = time.time()
    with pytest.raises(RuntimeError):
        R.remesh(far, f, 5, float(np.median(np.linalg.norm(v[f[:, 0]] - v[f[:, 1]], axis=1))), 0.5, 0)
    assert time.time() - t0 < 20.0


@pytest.mark.parametrize('case', ['ellipsoid', 'network'])
def test_partitioned_remesher_is_independent_of_the_thread_count_and_as_good_as_the_serial_one(case):
    """Above 40 000 faces the remesher works on 16 Morton runs of the faces at once (frozen rims), then on the strips along their seams at
    once, then on the patches around the strips' ends: the cuts depend on the mesh alone, so one thread and eight must give the same
    arrays; the result is a closed 2-manifold of the input's genus with the serial algorithm's edge statistics (not its arrays: the order
    of the operations differs), and the serial algorithm never had to take over (no edge of no length).  'network': the genus-2 tube /
    sheet network of BASELINE configs[3] (thin tubes: runs and strips that wrap around)."""
    import subprocess, sys, json, textwrap
    code = textwrap.dedent('''
        import sys, json, zlib, numpy as np
        sys.path.insert(0, %r)
        from ch_shrinkwrap_amd.trimesh import icosphere, TriMesh
        from ch_shrinkwrap_amd import remesh, synth
        if %r == 'network':
            sdf = lambda p: 2.0 * synth.sdf_er_sim2(np.asarray(p, 'f8') * 0.5)
            v, f = synth._c4_start_mesh(sdf, 2.96 / np.sqrt(0.06))      # ~50 000 vertices, mean edge ~11
            target = 9.5
        else:
            v, f = icosphere(6, 100.0)                       # 40 962 vertices, 81 920 faces, edge ~3.1
            v = (v * np.array([1.0, 0.7, 1.4], 'f4')).astype('f4')
            target = 2.2
        ov, of = remesh.remesh(v, f, n=5, target_edge_length=target, l=0.5, n_relax=0)[:2]
        m = TriMesh(ov, of)
        assert (m._halfedges['twin'] >= 0).all()             # closed
        e = np.linalg.norm(ov[of] - ov[np.roll(of, -1, 1)], axis=2)
        print(json.dumps(dict(nv=int(ov.shape[0]), nf=int(of.shape[0]), nf_in=int(f.shape[0]), crc=[zlib.crc32(ov.tobytes()), zlib.crc32(of.tobytes())],
                              mean=float(e.mean()), mn=float(e.min()), mx=float(e.max()), deg=int(np.bincount(of.ravel()).max()))))
    ''') % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), case)
    res = {}
    for tag, env in (('t1', {'NW_REMESH_THREADS': '1'}), ('t8', {'NW_REMESH_THREADS': '8'}), ('serial', {'NW_REMESH_PARTITION': '0'})):
        e = dict(os.environ); e.update(env); e['NWR_VERBOSE'] = '1'
        p = subprocess.run([sys.executable, '-c', code], env=e, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        res[tag] = json.loads(p.stdout.strip().splitlines()[-1])
        assert 'takes over' not in p.stderr
        if tag != 'serial':
            assert 'seam pass' in p.stderr and 'strips' in p.stderr        # (the partitioned path did run)
    assert res['t1'] == res['t8']
    a, s = res['t8'], res['serial']
    assert a['nf_in'] >= 40000
    genus = {'ellipsoid': 0, 'network': 2}[case]
    assert a['nv'] - a['nf'] // 2 == 2 - 2 * genus and a['nf'] % 2 == 0          # closed, the input's genus
    assert s['nv'] - s['nf'] // 2 == 2 - 2 * genus
    assert abs(a['nv'] - s['nv']) <= 0.02 * s['nv']
    assert abs(a['mean'] - s['mean']) <= 0.02 * s['mean']
    assert a['mx'] <= 1.15 * s['mx'] and a['mn'] >= 0.5 * s['mn'] and a['deg'] <= 16


def test_partitioned_remesher_refuses_what_the_serial_one_refuses():
    """Above 40 000 faces the input is no longer checked by matching the twins of the whole mesh: a directed edge that occurs twice is found
    inside a run by the run's own twin matching, across runs among the edges between two rim vertices.  Faces with the wrong sense, an
    extra face on an edge (three faces on it) and a doubled face must be refused with the serial path's error wherever they lie -- 40
    random places each, so that both kinds of place are hit --, a face that names a vertex twice or a vertex that does not exist with
    'bad argument', and the mesh as it was goes through."""
    v, f = icosphere(6, 100.0)
    assert f.shape[0] >= 40000
    rng = np.random.default_rng(5)
    R.remesh(v, f, 1, 3.0, 0.5, 0)
    for kind in ('flipped', 'fin', 'doubled'):
        for i in rng.choice(f.shape[0], 40, replace=False):
            g = f.copy()
            if kind == 'flipped':
                g[i] = g[i, ::-1]
            elif kind == 'fin':
                far = int((g[i, 0] + v.shape[0] // 2) % v.shape[0])
                g = np.vstack([g, [[g[i, 0], g[i, 1], far]]]).astype('i4')
            else:
                g = np.vstack([g, g[i:i + 1]]).astype('i4')
            for serial in (False, True):
                with pytest.raises(RuntimeError, match='2-manifold'):
                    R.remesh(v, g, 1, 3.0, 0.5, 0, serial=serial)
    g = f.copy()
    g[123, 1] = g[123, 0]
    with pytest.raises(RuntimeError, match='bad argument'):
        R.remesh(v, g, 1, 3.0, 0.5, 0)
    g = f.copy()
    g[77, 2] = v.shape[0]
    with pytest.raises(RuntimeError, match='bad argument'):
        R.remesh(v, g, 1, 3.0, 0.5, 0)


def test_worker_pool_survives_a_fork():
    """The remesher's worker threads live as long as the process.  A forked child has none of them (only the forking thread survives a
    fork): its first parallel loop must start a pool of its own instead of waiting for workers that are not there (pthread_atfork handler in
    csrc/remesh.cpp) -- and give the parent's result."""
    v, f = icosphere(6, 100.0)
    v = (v * np.array([1.0, 0.7, 1.4], 'f4')).astype('f4')
    ref_v, ref_f = R.remesh(v, f, 2, 2.5, 0.5, 0)                       # (above the partition size: the pool is running now)
    pid = os.fork()
    if pid == 0:
        code = 1
        try:
            import signal
            signal.alarm(120)                                            # a child that hangs is killed, the test then fails on its status
            cv, cf = R.remesh(v, f, 2, 2.5, 0.5, 0)
            code = 0 if (np.array_equal(cv, ref_v) and np.array_equal(cf, ref_f)) else 2
        finally:
            os._exit(code)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status
    again_v, again_f = R.remesh(v, f, 2, 2.5, 0.5, 0)                    # the parent's pool is unharmed
    assert np.array_equal(again_v, ref_v) and np.array_equal(again_f, ref_f)


def test_two_threads_in_the_remesher_at_once():
    """ctypes releases the GIL: two Python threads can be inside nwr_remesh together.  The worker pool serves one caller at a time; the other
    runs its loops on its own thread -- both get the result a lone call gets (the cuts do not depend on who runs what)."""
    import threading
    v, f = icosphere(6, 100.0)
    ref = R.remesh(v, f, 2, 2.5, 0.5, 0)
    out = [None, None, None]

    def work(i):
        out[i] = R.remesh(v, f, 2, 2.5, 0.5, 0)
    th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    for o in out:
        assert o is not None and np.array_equal(o[0], ref[0]) and np.array_equal(o[1], ref[1])


def test_an_edge_of_no_length_in_the_input_does_not_send_the_partitioned_remesher_to_the_serial_one():
    """The partitioned pass falls back to the serial algorithm if its result has an edge of (nearly) no length -- a net for what the partition
    might do.  An input that already has such an edge (two vertices the optimiser pulled onto each other: seen in a fit at 8 10^5 vertices,
    where the fall-back cost 4 s a call and mended nothing) must not trigger it."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r)
        from ch_shrinkwrap_amd.trimesh import icosphere
        from ch_shrinkwrap_amd import remesh
        v, f = icosphere(6, 100.0)
        for face in (1234, 40000, 77777):
            a, b, c = f[face]
            v[b] = v[a]; v[c] = v[a]                   # a face drawn together in a point: three edges of no length, and around them
                                                       # faces of no area, across which no collapse is admitted -- they stay
        ov, of = remesh.remesh(v, f, 3, 3.0, 0.5, 0)
        e = np.sort(np.concatenate([of[:, [0, 1]], of[:, [1, 2]], of[:, [2, 0]]]), 1)
        _, cnt = np.unique(e, axis=0, return_counts=True)
        assert (cnt == 2).all() and np.isfinite(ov).all()
        print('OK', ov.shape[0])
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e['NWR_VERBOSE'] = '1'
    p = subprocess.run([sys.executable, '-c', code], env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and 'OK' in p.stdout, p.stderr[-2000:]
    assert 'seam pass' in p.stderr                                   # the partitioned path ran ...
    assert 'takes over' not in p.stderr and 'and in the input: kept' in p.stderr        # ... and stood by its result


def test_needles_do_not_breed_coincident_vertices_in_the_partitioned_remesher():
    """Faces of no area with a long edge (a corner lying on the opposite edge: what a fit at 8 10^5 vertices produced here and there) used to be
    split again and again inside the runs of a partitioned mesh -- a needle that touches a frozen rim cannot be collapsed there --, and on one
    line all split points are dyadic: vertices piled up on the same positions, edges of no length appeared, and the safety net sent the whole
    mesh to the serial algorithm (the round-4 library does on this input).  Pieces now leave such an edge alone: no position is held twice,
    no fall-back.  The needles are put where they hurt: on faces with two RIM vertices (the 16 Morton runs of the faces are recomputed here)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import sys, numpy as np
        sys.path.insert(0, %r)
        from ch_shrinkwrap_amd.trimesh import icosphere
        from ch_shrinkwrap_amd import remesh
        v, f = icosphere(6, 100.0)
        v = (v * np.array([1.0, 0.7, 1.4], 'f4')).astype('f4')
        def spread(x):
            x = x.astype(np.uint64) & 0x3ff
            x = (x | (x << 16)) & 0x030000ff; x = (x | (x << 8)) & 0x0300f00f; x = (x | (x << 4)) & 0x030c30c3; x = (x | (x << 2)) & 0x09249249
            return x
        lo = v.min(0).astype('f8'); ext = float((v.max(0).astype('f8') - lo).max())
        c = (v[f[:, 0]].astype('f8') + v[f[:, 1]] + v[f[:, 2]]) / 3.0
        q = np.clip((c - lo) / ext * 1024.0, 0, 1023).astype(np.uint64)
        order = np.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), kind='stable')
        run = np.empty(f.shape[0], int)
        for r in range(16):
            run[order[f.shape[0] * r // 16: f.shape[0] * (r + 1) // 16]] = r
        lo_run = np.full(v.shape[0], 99); hi_run = np.full(v.shape[0], -1)
        np.minimum.at(lo_run, f.ravel(), np.repeat(run, 3)); np.maximum.at(hi_run, f.ravel(), np.repeat(run, 3))
        rim = lo_run != hi_run
        rng = np.random.default_rng(3)
        taken = np.zeros(v.shape[0], bool)
        n = 0
        for face in rng.permutation(f.shape[0]):
            a, b, c = f[face]
            if taken[[a, b, c]].any() or not (rim[a] and rim[b]) or rim[c]:
                continue
            v[c] = 0.5 * (v[a] + v[b])                 # the corner onto the middle of the opposite edge: a face of no area at a rim
            taken[[a, b, c]] = True
            n += 1
            if n == 300:
                break
        assert n == 300
        ov, of = remesh.remesh(v, f, 5, 1.1, 0.5, 0)   # edges are ~1.9: everything is too long, the needles' edges too
        u, cnt = np.unique(ov, axis=0, return_counts=True)
        e = np.sort(np.concatenate([of[:, [0, 1]], of[:, [1, 2]], of[:, [2, 0]]]), 1)
        _, ec = np.unique(e, axis=0, return_counts=True)
        print('RESULT', int((cnt > 1).sum()), bool((ec == 2).all()), ov.shape[0])
    ''') % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e['NWR_VERBOSE'] = '1'
    p = subprocess.run([sys.executable, '-c', code], env=e, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = [l for l in p.stdout.splitlines() if l.startswith('RESULT')][-1].split()
    assert 'seam pass' in p.stderr and 'takes over' not in p.stderr
    assert res[1] == '0' and res[2] == 'True' and int(res[3]) > 100000, res          # no position held by two vertices; closed; refined
