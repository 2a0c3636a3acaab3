"""
CPU tests: the oracle (oracle/nanowrap_oracle.py + oracle/nw_oracle.c) against the golden vectors generated
from the reference (tests/golden/make_golden.py), and -- in the build container -- against the live reference.
"""
import numpy as np
import pytest

from conftest import load_golden, rel_rms
from oracle import nanowrap_oracle as O
from oracle import ref_harness


def _mesh(g, prefix=''):
    return (g[prefix + 'vertices'], g[prefix + 'normals'], g[prefix + 'nbr'], g[prefix + 'faces'], g[prefix + 'valid'])


def test_stages_bit_exact():
    g = load_golden('stages_642')
    pos, nrm, nbr, faces, valid = _mesh(g)
    pts = g['points']
    s = 1.0 / g['sigma'].ravel()
    trace = []
    r = O.search(pos, nrm, nbr, faces, pts, [10.0], 3, s, valid=valid, trace=trace)
    assert r.loopcount == int(g['log_loopcount']) == 3
    for it, t in enumerate(trace):
        k = 'it%d_' % it
        assert np.array_equal(t['v_idx'], g[k + 'v_idx'])
        assert np.array_equal(t['w'], g[k + 'w'])
        assert np.array_equal(t['dmean'], g[k + 'dmean'])
        assert np.array_equal(t['pi'], g[k + 'pi'])
        assert np.array_equal(t['fdef'].reshape(-1, 3), g[k + 'fdef'])
        assert np.array_equal(t['S'][:, :t['n_search']], g[k + 'S'])
        assert np.array_equal(t['res'], g[k + 'res_masked'])          # all weights > 0: mask is all-true
        assert np.array_equal(t['fnew'], g[k + 'fnew'])
    assert np.array_equal(r.positions, g['positions'])
    assert np.array_equal(r.S, g['S_final'])
    assert np.array_equal(r.res, g['res_final'])
    assert np.allclose(r.tests, g['log_tests'], rtol=0, atol=0)
    assert np.allclose(r.ress, g['log_ress'], rtol=0, atol=0)
    assert float(r.cpred) == float(g['log_cpred'])
    assert float(r.wpreds[0]) == float(g['log_wpred'])


def test_brute_force_nn_matches_kdtree():
    g = load_golden('stages_642')
    pos, nrm, nbr, faces, valid = _mesh(g)
    cent = O.face_centroids(pos, faces)
    d0, i0 = O.nearest_faces(cent, g['points'], workers=1)
    d1, i1 = O.nearest_faces(cent, g['points'], brute=True)
    assert np.array_equal(i0, i1)
    assert np.allclose(d0, d1, rtol=1e-15, atol=0)


def test_c1_20_iterations():
    g = load_golden('c1_sphere_10k')
    pos, nrm, nbr, faces, valid = _mesh(g)
    s = 1.0 / g['sigma'].ravel()
    trace = []
    r = O.search(pos, nrm, nbr, faces, g['points'], [10.0], 20, s, valid=valid, trace=trace)
    assert np.array_equal(trace[0]['fnew'].reshape(-1, 3).astype('f4'), g['positions_1'])
    assert np.array_equal(trace[4]['fnew'].reshape(-1, 3).astype('f4'), g['positions_5'])
    assert np.array_equal(r.positions, g['positions_20'])
    assert rel_rms(r.positions, g['positions_20']) == 0.0
    assert np.array_equal(np.array(r.tests, 'f8'), g['log_tests'])
    assert np.array_equal(np.array(r.ress, 'f8'), g['log_ress'])


def test_variants():
    g = load_golden('variants_642')
    pos, nrm, nbr, faces, valid = (g['mesh_vertices'], g['mesh_normals'], g['mesh_nbr'], g['mesh_faces'], g['mesh_valid'])
    pts = g['points']
    # scalar sigma (passed through un-inverted, _membrane_mesh.pyx:1460-1461)
    r = O.search(pos, nrm, nbr, faces, pts, [10.0], 5, 10.0, valid=valid)
    assert np.array_equal(r.positions, g['scalar_positions'])
    assert np.array_equal(np.array(r.tests, 'f8'), g['scalar_log_tests'])
    # explicit weights with zeros -> mask
    r = O.search(pos, nrm, nbr, faces, pts, [10.0], 5, 1.0 / g['weights_sigma'].ravel(), weights=g['weights_weights'], valid=valid)
    assert np.array_equal(r.positions, g['weights_positions'])
    assert float(r.cpred) == float(g['weights_log_cpred'])
    # unused vertex slots + uniform background + an ignored second lambda
    hp = 'holes_mesh_'
    r = O.search(g[hp + 'vertices'], g[hp + 'normals'], g[hp + 'nbr'], g[hp + 'faces'], g['holes_points'], [10.0, 0.5], 5,
                 1.0 / np.full(3 * g['holes_points'].shape[0], 10.0, 'f4'), valid=g[hp + 'valid'])
    assert np.array_equal(r.positions, g['holes_positions'])
    assert np.array_equal(r.mesh_positions, g['holes_mesh_positions'])
    # two consecutive calls on one optimiser: the `tests` history is shared, positions restart from the mesh
    s = 1.0 / np.full(3 * pts.shape[0], 10.0, 'f4')
    hist = []
    r1 = O.search(pos, nrm, nbr, faces, pts, [10.0], 3, s, valid=valid, tests=hist)
    r2 = O.search(r1.mesh_positions, nrm, nbr, faces, pts, [10.0], 3, s, valid=valid, tests=hist)
    assert np.array_equal(r1.positions, g['twice_positions_a'])
    assert np.array_equal(r2.positions, g['twice_positions_b'])
    assert np.array_equal(np.array(hist, 'f8'), g['twice_log_tests'])


def test_native_helpers():
    import ctypes
    g = load_golden('native_helpers')
    L = O.lib()
    nbr = np.ascontiguousarray(g['nbr'])
    M, NB = nbr.shape
    x, f0 = g['x'], g['f0']
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    for name, call in (('l', lambda d: L.nwo_lfunc(P(x), P(nbr), M, NB, P(d))),
                       ('lh', lambda d: L.nwo_lhfunc(P(x), P(nbr), M, NB, P(d))),
                       ('lw', lambda d: L.nwo_lwfunc(P(x), P(f0), P(nbr), M, NB, P(d))),
                       ('lhw', lambda d: L.nwo_lhwfunc(P(x), P(f0), P(nbr), M, NB, P(d))),
                       ('vaw', lambda d: L.nwo_vertex_area_weights(P(f0), P(nbr), M, NB, P(d)))):
        d = np.zeros(3 * M, 'f4')
        call(d)
        assert np.array_equal(d, g['out_' + name]), name
    z = O.apply_At(g['ah_r'].ravel(), g['ah_v_idx'], g['ah_w'], M).reshape(M, 3)
    assert np.array_equal(z, g['ah_out'])


@pytest.mark.reference
@pytest.mark.skipif(not ref_harness.available(), reason='reference only exists in the build container')
def test_live_reference_random_case():
    """Fresh seed, not in the fixtures: oracle vs the reference run side by side."""
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    v, f = icosphere(3, 60.0)
    rng = np.random.default_rng(1234)
    d = rng.normal(size=(4000, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    pts = (d * 50 + rng.normal(scale=4, size=d.shape)).astype('f4')
    sigma = rng.uniform(2, 6, size=pts.shape).astype('f4')
    s = 1.0 / sigma.ravel()
    mesh = TriMesh(v, f)
    cg = ref_harness.new_reference_optimiser(mesh, pts, search_k=200, search_rad=100, shield_sigma=2.0)
    out = cg.search(pts, lams=[5.0], num_iters=7, sigma_inv=s, weights=None)
    m2 = TriMesh(v, f)
    r = O.search(m2.vertices.copy(), m2.vertex_normals.copy(), m2.neighbor_vertex_table(), m2.faces, pts, [5.0], 7, s)
    assert np.array_equal(r.positions, out)
    assert np.array_equal(np.array(r.tests), np.array(cg.tests))


def test_oracle_float64_points_and_wfunc_against_golden():
    """Round-2 fixture (tests/golden/make_golden.py::golden_f64_and_regulariser): the oracle follows the dtype of `points` like the
    reference (float64 residual / A f / Gc for float64 localizations) and restates the 'wfunc' regulariser; both bit-identical."""
    from ch_shrinkwrap_amd.trimesh import TriMesh
    g = load_golden('f64_and_wfunc')
    s = 1.0 / np.full(g['points_f32'].size, 10.0, 'f4')

    def mesh():
        return TriMesh(g['mesh_vertices'], g['mesh_faces'])

    for name, pts in (('f32', g['points_f32']), ('f64_same', g['points_f32'].astype('f8')), ('f64_raw', g['points_f64_raw'])):
        m = mesh()
        r = O.search(m.vertices.copy(), m.vertex_normals.copy(), m.neighbor_vertex_table(), m.faces, pts, [10.0], 5, s)
        assert np.array_equal(r.positions, g[name + '_positions']), name
        assert np.array_equal(np.array(r.tests, 'f8'), g[name + '_log_tests']), name
        assert str(r.res.dtype) == str(g[name + '_res_dtype'])
    m = mesh()
    r = O.search(m.vertices.copy(), m.vertex_normals.copy(), m.neighbor_vertex_table(), m.faces, g['points_f32'], [float(g['wfunc_lams'][0])], 4, s,
                 regulariser='wfunc')
    assert np.array_equal(r.positions, g['wfunc_positions'])
    assert np.array_equal(r.S, g['wfunc_S_final'])
    assert np.array_equal(np.array(r.tests, 'f8'), g['wfunc_log_tests'])


@pytest.mark.reference
@pytest.mark.skipif(not ref_harness.available(), reason='reference only exists in the build container')
def test_live_reference_alternate_regularisers_fail_upstream():
    """Why only ["I"] and ["wfunc"] are offered inside the loop: with the live default `_ncc()` (float64 because of its integer
    division, mesh_conj_grad.py:782-800) the names that go through conj_grad_utils.c -- "Lfunc", "Lfunc3" -- hand a float64 buffer
    to C code that reads float32 (conj_grad_utils.c:286-302, no dtype check): the reference raises in its first iteration."""
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    v, f = icosphere(2, 60.0)
    rng = np.random.default_rng(7)
    d = rng.normal(size=(800, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    pts = (d * 50 + rng.normal(scale=4, size=d.shape)).astype('f4')
    s = 1.0 / np.full(pts.size, 4.0, 'f4')
    for names in (["Lfunc", "Lhfunc"], ["Lfunc3", "Lhfunc3"]):
        mesh = TriMesh(v, f)
        cg = ref_harness.new_reference_optimiser(mesh, pts, search_k=200, search_rad=100, shield_sigma=2.0)
        cg.Lfuncs, cg.Lhfuncs = [names[0]], [names[1]]
        with np.errstate(all='ignore'), pytest.raises((AssertionError, ValueError, FloatingPointError)):
            cg.search(pts, lams=[1.0], num_iters=3, sigma_inv=s)
    mesh = TriMesh(v, f)
    cg = ref_harness.new_reference_optimiser(mesh, pts, search_k=200, search_rad=100, shield_sigma=2.0)
    cg.Lfuncs, cg.Lhfuncs = ["wfunc"], ["wfunc"]
    out = cg.search(pts, lams=[10.0], num_iters=3, sigma_inv=s)
    m2 = TriMesh(v, f)
    r = O.search(m2.vertices.copy(), m2.vertex_normals.copy(), m2.neighbor_vertex_table(), m2.faces, pts, [10.0], 3, s, regulariser='wfunc')
    assert np.array_equal(r.positions, out)


def test_oracle_data_other_than_the_localizations_against_golden():
    """search(data, ...) with `data` different from the optimiser's localizations (mesh_conj_grad.py:150: weight matrix from
    `self.points`, residual against `data`): fixture from the reference (tests/golden/make_golden.py::golden_data_target)."""
    g = load_golden('data_target')
    pos, nrm, nbr, faces, _ = _mesh(g, 'mesh_')
    s = 1.0 / np.full(g['points'].size, 10.0, 'f4')
    r = O.search(pos.copy(), nrm.copy(), nbr, faces, g['points'], [10.0], 5, s, data=g['data'])
    assert np.array_equal(r.positions.astype('f4'), g['positions'])
    assert np.array_equal(np.array(r.ress, 'f8'), g['log_ress'])
    assert np.array_equal(np.array(r.tests, 'f8'), g['log_tests'])
    # and it is not the fit to the localizations themselves
    r0 = O.search(pos.copy(), nrm.copy(), nbr, faces, g['points'], [10.0], 5, s)
    assert rel_rms(r0.positions, g['positions']) > 1e-5
