"""
Host logic of the outer-loop driver (ch_shrinkwrap_amd.membrane_mesh.MembraneMesh.opt_conjugate_gradient), which mirrors
/root/reference/ch_shrinkwrap/_membrane_mesh.pyx:1427-1560.  CPU tests replace the optimiser class with a recorder;
the GPU test runs two real blocks against the oracle driven with the same block structure.
"""
import numpy as np
import pytest

from conftest import rel_rms
from ch_shrinkwrap_amd import membrane_mesh as mm
from ch_shrinkwrap_amd.trimesh import icosphere, geodesic_sphere, TriMesh


class _Recorder(object):
    calls = []
    refreshes = 0

    def __init__(self, mesh, points, **kw):
        self.mesh, self.kw = mesh, kw

    def search(self, points, lams=None, num_iters=None, sigma_inv=None, weights=None):
        _Recorder.calls.append(dict(lams=list(lams), num_iters=num_iters, sigma_inv=sigma_inv, weights=weights, kw=self.kw))
        return self.mesh.vertices

    def refresh_normals(self):
        _Recorder.refreshes += 1


@pytest.fixture
def recorder(monkeypatch):
    _Recorder.calls = []
    _Recorder.refreshes = 0
    monkeypatch.setattr(mm, 'ShrinkwrapMeshConjGrad', _Recorder)

    class _Native(object):
        mesh_key = None
    monkeypatch.setattr(mm, 'NativeContext', lambda device=0: _Native())
    return _Recorder


def _mesh(**kw):
    v, f = icosphere(2, 50.0)
    return mm.MembraneMesh(v, f, **kw)


def test_block_schedule_and_lambda(recorder):
    m = _mesh(kc=1.0, step_size=20.0, max_iter=39, remesh_frequency=5, delaunay_remesh_frequency=0, truncate_at=1000)
    pts = np.zeros((7, 3), 'f4')
    sigma = np.full((7, 3), 10.0, 'f4')
    n = m.shrink_wrap(pts, sigma, method='conjugate_gradient', minimum_edge_length=5.0)
    assert n == 39
    its = [c['num_iters'] for c in recorder.calls]
    assert its == [5] * 7 + [4]                                     # blocks of remesh_frequency, last one truncated
    assert all(c['lams'] == [10.0] for c in recorder.calls)         # step_size*kc/2  (_membrane_mesh.pyx:1486)
    assert recorder.refreshes == 8                                  # normals refreshed after every block (:1524-1527)
    assert all(c['kw']['reuse_device_mesh'] for c in recorder.calls)
    assert np.allclose(recorder.calls[0]['sigma_inv'], 0.1) and recorder.calls[0]['sigma_inv'].shape == (21,)
    assert recorder.calls[0]['kw']['shield_sigma'] == pytest.approx(m._mean_edge_length / 2.0)
    # remesh target-length schedule (:1443-1455, :1544): linear from the initial mean edge length to minimum_edge_length
    assert len(m.block_log) == 7
    L0 = m._mean_edge_length
    slope = (5.0 - L0) / (5 * np.ceil(39 / 5))
    assert m.block_log[0]['target_length'] == pytest.approx(L0 + slope * 6)


def test_sigma_forms_truncate_and_shrink_weight(recorder):
    pts = np.zeros((4, 3), 'f4')
    m = _mesh(kc=2.0, step_size=3.0, max_iter=10, remesh_frequency=0, delaunay_remesh_frequency=0, shrink_weight=0.5, truncate_at=6)
    m.shrink_wrap(pts, 12.5)                                        # scalar sigma is passed through UN-inverted (:1460-1461)
    assert recorder.calls[-1]['sigma_inv'] == 12.5 and recorder.calls[-1]['num_iters'] == 6
    assert recorder.calls[-1]['lams'] == [3.0, 0.5]
    m.shrink_wrap(pts, np.array([1.0, 2.0, 4.0, 8.0]))              # (N,) -> 1/repeat(sigma, 3)  (:1462-1466)
    assert np.allclose(recorder.calls[-1]['sigma_inv'], np.repeat([1.0, 0.5, 0.25, 0.125], 3))
    with pytest.raises(ValueError):
        m.shrink_wrap(pts, np.ones((3, 3)))
    # gcd of remesh and punch frequencies (:1433-1435)
    recorder.calls.clear()
    m2 = _mesh(max_iter=12, remesh_frequency=6, delaunay_remesh_frequency=4)
    m2.shrink_wrap(pts, 10.0)
    assert [c['num_iters'] for c in recorder.calls] == [2] * 6
    # continuing a fit re-uses the cached points/sigma (:1650-1667)
    recorder.calls.clear()
    m2.shrink_wrap(max_iter=2)
    assert len(recorder.calls) == 1 and recorder.calls[0]['sigma_inv'] == 10.0


def test_neck_removal_schedule(recorder, monkeypatch):
    """remove_necks runs at remesh boundaries once j > neck_first_iter (_membrane_mesh.pyx:1537-1540); the selection is
    Gaussian curvature outside [low, high] (:1201-1215) and the surgery itself goes to the hook."""
    m = _mesh(max_iter=20, remesh_frequency=5, delaunay_remesh_frequency=0, neck_first_iter=9, neck_threshold_low=-1e-3, neck_threshold_high=1e-2)
    K = np.zeros(m.vertices.shape[0], 'f4')
    K[[3, 17]] = [-0.5, 0.3]
    K[5] = 5e-3                                                      # inside the band
    monkeypatch.setattr(mm.MembraneMesh, '_populate_curvature_grad', lambda self: setattr(self, '_K', K))
    removed = []
    m.neck_remover = lambda mesh, verts: removed.append(verts.copy())
    m.shrink_wrap(np.zeros((4, 3), 'f4'), 10.0)                      # thresholds are mesh attributes (surface_fitting.py:66-68)
    assert [e['iteration'] for e in m.neck_log] == [10, 15, 20]     # j = 5 is not > neck_first_iter
    assert all(e['candidates'] == 2 for e in m.neck_log)
    assert len(removed) == 3 and all(np.array_equal(r, [3, 17]) for r in removed)


def test_builtin_remesher_in_the_block_schedule(recorder):
    """remesher='builtin' (host C++, runs without a GPU): every remesh boundary hands the next optimiser a new topology at
    the scheduled target edge length (_membrane_mesh.pyx:1443-1455, :1544-1546)."""
    m = _mesh(kc=1.0, step_size=20.0, max_iter=15, remesh_frequency=5, delaunay_remesh_frequency=0, remesher='builtin')
    sizes = []
    orig = recorder.search

    def search(self, *a, **k):
        sizes.append((self.mesh.vertices.shape[0], self.mesh.faces.shape[0]))
        return orig(self, *a, **k)
    recorder.search = search
    try:
        L0 = m._mean_edge_length
        m.shrink_wrap(np.zeros((5, 3), 'f4'), 10.0, minimum_edge_length=L0 / 3)
    finally:
        recorder.search = orig
    assert len(sizes) == 3 and sizes[0] == (162, 320)
    assert sizes[1][0] > sizes[0][0] and sizes[2][0] > sizes[1][0]             # edges shrink -> more vertices every block
    assert all(f == 2 * v - 4 for v, f in sizes)                                # still closed, genus 0
    for b in m.block_log:
        assert abs(b['mean_length'] - b['target_length']) < 0.2 * b['target_length']
    assert m.block_log[0]['target_length'] > m.block_log[-1]['target_length']
    assert m._vertices['neighbors'].shape[0] == m.vertices.shape[0] and m.cg is None
    with pytest.raises(ValueError):
        _mesh(remesher='pyme').remesh()
    # 'device' = the same step as kernels: no silent fall-back to the host remesher where there is no GPU
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            _mesh(remesher='device').remesh(5, 3.0, 0.5, n_relax=0)


def test_result_buffers_are_recycled_only_when_the_caller_dropped_them():
    """search() returns a fresh (M,3) array every call as far as the caller can tell: an array is reused only when nothing
    outside the optimiser refers to it any more."""
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

    class Stub(object):
        M = 100
        _result_buffer = ShrinkwrapMeshConjGrad._result_buffer

        def __init__(self):
            self._fs_pool = []

        def step(self):                      # what _finish does with the buffer
            out = self._result_buffer()
            self.fs = out
            self.f = out.ravel()
            return np.real(self.fs)
    s = Stub()
    ids = []
    for _ in range(6):                       # results dropped immediately: two buffers alternate
        ids.append(s.step().__array_interface__['data'][0])
    assert len(set(ids)) == 2
    kept = s.step()
    kept[:] = 7.0
    others = [s.step() for _ in range(5)]
    assert all(o is not kept for o in others) and (kept == 7.0).all()      # a kept result is never handed out again
    view = s.step()[::2]                                                    # ... nor one that is only referenced through a view
    later = [s.step() for _ in range(4)]
    assert all(not np.shares_memory(view, o) for o in later)


def test_recipe_module_parameter_surface():
    r = mm.ShrinkwrapMembrane()
    # defaults of recipe_modules/surface_fitting.py:17-42
    assert (r.max_iters, r.curvature_weight, r.remesh_frequency, r.punch_frequency, r.kc) == (39, 20.0, 5, 0, 1.0)
    assert (r.neck_threshold_low, r.neck_threshold_high, r.neck_first_iter, r.truncate_at, r.minimum_edge_length) == (-1e-3, 1e-2, 9, 1000, 5.0)
    with pytest.raises(AttributeError):
        mm.ShrinkwrapMembrane(not_a_trait=1)

    class Few(object):
        faces = np.zeros((4, 3), 'i4')
        vertices = np.zeros((4, 3), 'f4')
    with pytest.raises(RuntimeError):
        r.execute({'surf': Few(), 'filtered_localizations': {}})


def test_ring_tables_of_open_meshes_cover_the_whole_fan():
    v, f = icosphere(1, 10.0)
    f = f[v[f].mean(1)[:, 2] > 0]                                   # a cap with a boundary loop
    m = TriMesh(v, f)
    val = (m.neighbor_vertex_table() >= 0).sum(1)
    e, cn = np.unique(np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1), axis=0, return_counts=True)
    deg = np.bincount(e.ravel(), minlength=v.shape[0])
    on_boundary = np.zeros(v.shape[0], bool)
    on_boundary[e[cn == 1].ravel()] = True
    assert (val[~on_boundary] == deg[~on_boundary]).all()           # closed fans: every neighbour once
    assert (val[on_boundary] == deg[on_boundary] - 1).all()         # open fans: their outgoing half-edges (documented convention)


def test_geometry_refresh_matches_definition():
    v, f = geodesic_sphere(6, 30.0)
    m = TriMesh(v, f)
    n = m.vertex_normals
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-5)
    assert np.allclose(n, v / 30.0, atol=2e-2)                       # sphere: normals are radial
    assert m.area() == pytest.approx(4 * np.pi * 900.0, rel=2e-2)


@pytest.mark.gpu
def test_two_blocks_against_oracle():
    from oracle import nanowrap_oracle as O
    from ch_shrinkwrap_amd import synth
    v, f = icosphere(4, 120.0)
    pts = synth.sphere_cloud(20000, 100.0, 10.0, seed=5)
    sigma = np.full(pts.shape, 10.0, 'f4')
    m = mm.MembraneMesh(v, f, kc=1.0, step_size=20.0, max_iter=8, remesh_frequency=4, delaunay_remesh_frequency=0)
    m.shrink_wrap(pts, sigma, minimum_edge_length=5.0)
    # oracle: same two blocks, normals refreshed between them with the same substrate
    ref = TriMesh(v, f)
    s = 1.0 / sigma.ravel()
    for _ in range(2):
        r = O.search(ref.vertices.copy(), ref.vertex_normals.copy(), ref.neighbor_vertex_table(), ref.faces, pts, [10.0], 4, s)
        ref._vertices['position'][:] = r.positions
        ref.update_geometry()
    assert rel_rms(m.vertices, ref.vertices) <= 1e-5
    dn = np.abs(m.vertex_normals - ref.vertex_normals).max(1)      # normals of sliver triangles amplify 1e-7 position noise
    assert np.mean(dn < 1e-4) > 0.99 and dn.max() < 2e-2
    assert m.S0.shape == m.vertices.shape and np.isfinite(m.point_dis).all() and np.isfinite(m.rms_point_sc).all()
    pi = m.point_influence
    assert np.allclose(pi, m.cg.point_influence, rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_device_normal_refresh_matches_host_definition():
    """Row f1: the block-boundary vertex-normal refresh on the device (nw_refresh_normals) against the host definition
    (trimesh.TriMesh.update_geometry), and a new optimiser re-using the resident mesh restarts its history."""
    from ch_shrinkwrap_amd import synth
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext
    v, f = geodesic_sphere(10, 100.0)
    pts = synth.sphere_cloud(8000, 90.0, 5.0, seed=1)
    s = 1.0 / np.full(pts.size, 5.0, 'f4')
    mesh = TriMesh(v, f)
    nat = NativeContext(0)
    cg = ShrinkwrapMeshConjGrad(mesh, pts, native=nat, reuse_device_mesh=True)
    cg.search(pts, lams=[10.0], num_iters=3, sigma_inv=s)
    dev = cg.refresh_normals().copy()
    assert np.array_equal(mesh.vertex_normals, dev)
    mesh.update_geometry()                                   # host definition on the same (written-back) positions
    assert np.abs(mesh.vertex_normals - dev).max() < 2e-5
    mesh._vertices['normal'][:] = dev
    cg2 = ShrinkwrapMeshConjGrad(mesh, pts, native=nat, reuse_device_mesh=True)     # no mesh upload, history restarted
    out = cg2.search(pts, lams=[10.0], num_iters=2, sigma_inv=s)
    assert len(cg2.tests) == 2 and cg2.loopcount == 2
    # same as a fresh context fed the same arrays
    mesh3 = TriMesh(v, f)
    mesh3._vertices['position'][:] = cg.fs
    mesh3._vertices['normal'][:] = dev
    out3 = ShrinkwrapMeshConjGrad(mesh3, pts).search(pts, lams=[10.0], num_iters=2, sigma_inv=s)
    assert rel_rms(out, out3) <= 1e-6


@pytest.mark.gpu
def test_neck_selection_on_network():
    """BASELINE.json configs[3] in small: two blocks on the ER-like network with the curvature kernel at the block
    boundary selecting neck candidates."""
    from ch_shrinkwrap_amd import synth
    c = synth.make_config('c4', scale=0.02, seed=4)
    m = mm.MembraneMesh(c['vertices'], c['faces'], kc=1.0, step_size=20.0, max_iter=10, remesh_frequency=5, delaunay_remesh_frequency=0,
                        neck_first_iter=4, neck_threshold_low=-1e-3, neck_threshold_high=1e-2)
    picked = []
    m.neck_remover = lambda mesh, verts: picked.append(verts.copy())
    m.shrink_wrap(c['points'], c['sigma'])
    assert [e['iteration'] for e in m.neck_log] == [5, 10]
    K = m.curvature_gaussian
    assert np.isfinite(K).all() and np.isfinite(m.curvature_mean).all()
    want = np.flatnonzero((K < -1e-3) | (K > 1e-2))
    assert np.array_equal(picked[-1], want)
    # tubes of radius ~100-120 nm: |K| of the fitted surface stays far below the high threshold nearly everywhere
    assert want.size < 0.05 * K.size


@pytest.mark.gpu
def test_shrink_wrap_with_builtin_remesher():
    """the outer loop with real remeshing between blocks (target edge length falling linearly to
    `minimum_edge_length`, _membrane_mesh.pyx:1443-1455, :1544): every block runs on a new topology, the fit converges to
    the sampled sphere and the final mesh is a closed manifold at the requested resolution."""
    from ch_shrinkwrap_amd import synth
    v, f = icosphere(3, 125.0)                                   # 642 vertices, edges ~19 nm
    pts = synth.sphere_cloud(40000, 100.0, 5.0, seed=2)
    sigma = np.full(pts.shape, 5.0, 'f4')
    m = mm.MembraneMesh(v, f, kc=1.0, step_size=20.0, max_iter=25, remesh_frequency=5, delaunay_remesh_frequency=0, remesher='builtin')
    n = m.shrink_wrap(pts, sigma, minimum_edge_length=7.0)
    assert n == 25 and len(m.block_log) == 5
    counts = [b['mean_length'] for b in m.block_log]
    assert counts[0] > counts[-1] and abs(counts[-1] - m.block_log[-1]['target_length']) < 0.2 * counts[-1]
    assert m.vertices.shape[0] > 4 * 642                          # refined: 19 nm -> 7 nm edges
    fcs = m.faces
    e = np.sort(np.concatenate([fcs[:, [0, 1]], fcs[:, [1, 2]], fcs[:, [2, 0]]]), 1)
    ue, cn = np.unique(e, axis=0, return_counts=True)
    assert (cn == 2).all() and m.vertices.shape[0] - ue.shape[0] + fcs.shape[0] == 2
    r = np.linalg.norm(m.vertices, axis=1)
    assert abs(r.mean() - 100.0) < 1.0 and r.std() < 1.5          # the cloud's sphere, noise averaged out
    assert ((m._vertices['neighbors'] != -1).sum(1) <= 12).all()


@pytest.mark.gpu
def test_recipe_module_end_to_end_on_the_network():
    """The recipe-module mirror end to end (examples/fit_network.py): the module's default 39 iterations in blocks of 5 on the
    genus-2 network with remeshing between blocks and neck selection after iteration 9; the surface moves from its 20 nm
    offset towards the true one (the fit is gradual at curvature_weight 20: 12 nm after 19 iterations, 7 after 39, 3 after 79,
    with or without remeshing) and keeps its topology."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('fit_network', os.path.join(os.path.dirname(os.path.dirname(__file__)), 'examples', 'fit_network.py'))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    mesh, d = ex.main(0.02)
    assert np.sqrt((d * d).mean()) < 9.0                        # started 20 nm off; localization error 10 nm
    f = mesh.faces
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    ue, cn = np.unique(e, axis=0, return_counts=True)
    assert (cn == 2).all() and mesh.vertices.shape[0] - ue.shape[0] + f.shape[0] == -2
    assert [b['iteration'] for b in mesh.block_log] == list(range(5, 40, 5)) and [b['iteration'] for b in mesh.neck_log] == list(range(10, 40, 5))
    assert mesh.block_log[-1]['mean_length'] < mesh.block_log[0]['mean_length']


def test_hook_that_edits_the_records_in_place_reaches_the_next_optimiser(recorder):
    """A surgery hook (hole puncher, edge cleaner, neck remover, a callable remesher driving PYME) may edit the half-edge / vertex
    records in place.  The 1-ring vertex table the optimiser uploads (mesh_conj_grad.py:50-54) is cached per topology: every hook must
    drop it, or the next block's curvature prior works on a stale ring."""
    tables = []

    class Rec(recorder):
        def __init__(self, mesh, points, **kw):
            recorder.__init__(self, mesh, points, **kw)
            tables.append(mesh.neighbor_vertex_table().copy())

    mm.ShrinkwrapMeshConjGrad = Rec            # (the fixture's monkeypatch restores the module attribute afterwards)
    m = _mesh(kc=1.0, step_size=20.0, max_iter=10, remesh_frequency=5, delaunay_remesh_frequency=0, remesher=None)
    victim = 17
    deg = int((m._vertices['neighbors'][victim] != -1).sum())

    def cleaner(mesh):
        mesh._vertices['neighbors'][victim, deg - 1] = -1          # in place: the last ring entry of one vertex goes

    m.edge_cleaner = cleaner
    m.remesh = lambda *a, **k: False                                # topology held otherwise
    pts = np.zeros((7, 3), 'f4')
    m.shrink_wrap(pts, np.full((7, 3), 10.0, 'f4'), method='conjugate_gradient', minimum_edge_length=5.0)
    assert len(tables) == 2
    assert tables[0][victim, deg - 1] >= 0 and tables[1][victim, deg - 1] == -1
    assert np.array_equal(np.delete(tables[0], victim, 0), np.delete(tables[1], victim, 0))


def test_topology_records_built_on_first_use_equal_the_eager_ones(recorder):
    """`lazy_topology` (what the driver asks for between two blocks): positions, geometry refreshes, the mean edge length and the test for a
    vertex slot in use work from the face array alone; the first access to the half-edge records, the vertex records or a ring table builds
    the topology, and everything then equals a mesh built the eager way -- with unused vertex slots as well."""
    v, f = geodesic_sphere(6, 30.0)
    for spare in (0, 5):
        eager = TriMesh(v, f, max_vertices=v.shape[0] + spare)
        lazy = TriMesh(v, f, max_vertices=v.shape[0] + spare, lazy_topology=True)
        assert lazy.__dict__['_topology_pending']
        assert np.array_equal(lazy.vertices, eager.vertices)
        assert lazy._mean_edge_length == eager._mean_edge_length
        assert np.array_equal(lazy.valid_vertex_mask(), eager._vertices['halfedge'] != -1)
        assert np.array_equal(lazy.vertex_normals, eager.vertex_normals)
        lazy.vertices[:] *= 1.01
        eager.vertices[:] *= 1.01
        lazy.update_geometry()
        eager.update_geometry()
        assert lazy.area() == eager.area()
        assert lazy.__dict__['_topology_pending'], 'none of the above needs the half-edge records'
        first = (lambda: lazy._halfedges, lambda: lazy._vertices, lambda: lazy.neighbor_vertex_table(), lambda: lazy.vertex_neighbors)[spare % 4]
        first()
        assert not lazy.__dict__['_topology_pending']
        assert np.array_equal(lazy._halfedges, eager._halfedges) and np.array_equal(lazy._origin, eager._origin)
        assert np.array_equal(lazy._vertices, eager._vertices)
        assert np.array_equal(lazy.neighbor_vertex_table(), eager.neighbor_vertex_table())
    # the driver: after a fit with the built-in remesher the records are there for whoever asks
    m = _mesh(kc=1.0, step_size=20.0, max_iter=10, remesh_frequency=5, delaunay_remesh_frequency=0, remesher='builtin')
    m.shrink_wrap(np.zeros((5, 3), 'f4'), 10.0, minimum_edge_length=m._mean_edge_length / 2)
    assert m.__dict__['_topology_pending'], 'nothing in the block loop asked for the host topology'
    ref = TriMesh(m.vertices.copy(), m.faces)
    assert np.array_equal(m._vertices['neighbors'], ref._vertices['neighbors']) and np.array_equal(m._halfedges['twin'], ref._halfedges['twin'])
    assert np.array_equal(m._halfedges['length'], ref._halfedges['length'])


def test_geometry_computed_on_first_use_equals_the_eager_one(recorder):
    """A mesh built with `mean_edge` (what the driver gives the mesh of a new topology inside a fit: the remesher's own figure) computes face
    normals, areas and edge lengths when somebody asks; until then `_mean_edge_length` is the given number.  Whatever is asked first --
    `_faces`, `face_normals`, `area()`, the half-edge records, an explicit `update_geometry` --, the arrays equal an eagerly built mesh's."""
    v, f = geodesic_sphere(6, 30.0)
    eager = TriMesh(v, f)
    given = float(eager._mean_edge_length) * 1.0000001
    for first in ('_faces', 'face_normals', 'area', '_halfedges', 'update_geometry', 'update_geometry_with_normals'):
        lazy = TriMesh(v, f, vertex_normals=False, lazy_topology=True, all_referenced=True, mean_edge=given)
        assert lazy.__dict__['_geometry_pending'] and lazy.__dict__['_topology_pending']
        assert lazy._mean_edge_length == np.float32(given) and lazy.__dict__['_geometry_pending']
        if first == 'area':
            assert lazy.area() == eager.area()
        elif first == 'update_geometry':
            lazy.update_geometry(vertex_normals=False)
        elif first == 'update_geometry_with_normals':
            lazy.update_geometry()
        else:
            getattr(lazy, first)
        assert not lazy.__dict__['_geometry_pending']
        assert np.array_equal(lazy._faces, eager._faces) and np.array_equal(lazy.face_normals, eager.face_normals)
        assert np.array_equal(lazy._halfedges, eager._halfedges)
        assert np.array_equal(lazy.vertex_normals, eager.vertex_normals)
        if first.startswith('update_geometry'):
            assert lazy._mean_edge_length == eager._mean_edge_length        # an explicit refresh recomputes it
    # the driver: a fit with the built-in remesher hands every new topology the remesher's mean edge length
    m = _mesh(kc=1.0, step_size=20.0, max_iter=10, remesh_frequency=5, delaunay_remesh_frequency=0, remesher='builtin')
    m.shrink_wrap(np.zeros((5, 3), 'f4'), 10.0, minimum_edge_length=m._mean_edge_length / 2)
    ref = TriMesh(m.vertices.copy(), m.faces)
    assert abs(m.block_log[-1]['mean_length'] - float(ref._mean_edge_length)) < 1e-5 * float(ref._mean_edge_length)
    assert np.array_equal(m.face_normals, ref.face_normals) and m.area() == ref.area()
