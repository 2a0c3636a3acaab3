"""
Fit quality in the reference's own metric (ch_shrinkwrap_amd/evaluation.py restates evaluation_utils.points_from_mesh :35-150 and
average_squared_distance :153-180 of /root/reference/ch_shrinkwrap/, surfaced upstream by recipe_modules/surface_feature_extraction.py:76-138).

CPU: the restatement against the fixture the reference's own functions produced (tests/golden/fit_quality.npz, make_golden.py) and, when
/root/reference is present, against the live functions.  GPU: complete fits asserted in that metric against points on the true surface --
the only handle on the quality of the block-boundary remesher, whose PYME counterpart is not in the reference tree.
"""
import os
import sys
import numpy as np
import pytest

from conftest import load_golden
from ch_shrinkwrap_amd import evaluation as E
from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere


def _sorted(p):
    return p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]


def test_restatement_equals_the_reference_fixture():
    g = load_golden('fit_quality')
    mesh = TriMesh(g['vertices'], g['faces'])
    for tag, dx in (('dx5', 5.0), ('dx11', 11.0)):
        pts = _sorted(E.points_from_mesh(mesh, dx_min=dx, p=1.0))
        ref = g['points_' + tag]
        assert pts.shape == ref.shape and pts.shape[0] > 100
        assert np.array_equal(pts, ref)                               # the same grid nodes, bit for bit
        m0, m1 = E.average_squared_distance(pts, g['truth'])
        assert np.allclose([m0, m1, np.sqrt((m0 + m1) / 2)], g['mse_' + tag], rtol=1e-13, atol=0)
    q = E.fit_quality(mesh, g['truth'])
    assert np.isclose(q['mse_rms'], g['mse_dx5'][2], rtol=1e-13) and q['n_mesh_points'] == g['points_dx5'].shape[0]


@pytest.mark.reference
def test_live_reference_points_and_distances():
    from oracle import ref_harness
    if not ref_harness.available():
        pytest.skip('reference not present (GPU box)')
    ev = ref_harness.load_evaluation_utils()
    v, f = icosphere(3, 75.0)
    v = (v * np.array([1.0, 0.6, 1.3], 'f4')).astype('f4')
    mesh = TriMesh(v, f)
    for dx in (3.0, 7.5):
        np.random.seed(1)
        ref = _sorted(np.asarray(ev.points_from_mesh(mesh, dx_min=dx, p=1.0)))
        mine = _sorted(E.points_from_mesh(mesh, dx_min=dx, p=1.0))
        assert np.array_equal(ref, mine)
    rng = np.random.default_rng(3)
    a, b = rng.normal(size=(500, 3)), rng.normal(size=(700, 3))
    assert E.average_squared_distance(a, b) == tuple(ev.average_squared_distance(a, b))


def test_a_mesh_of_the_surface_itself_scores_the_sampling_floor():
    """An icosphere ON the sphere against points on the sphere: the metric's floor is the spacing of the two samplings, not zero."""
    v, f = icosphere(4, 100.0)
    mesh = TriMesh(v, f)
    rng = np.random.default_rng(0)
    d = rng.normal(size=(5027, 3))                                    # 0.04 per nm^2 on 4 pi 100^2
    truth = (100.0 * d / np.linalg.norm(d, axis=1)[:, None]).astype('f4')
    q = E.fit_quality(mesh, truth)
    assert 1.5 < q['mse_rms'] < 3.5
    off = TriMesh((v * 1.1).astype('f4'), f)                           # 10 nm off: the metric says so
    assert 9.5 < E.fit_quality(off, truth)['mse_rms'] < 11.5
    sub = E.points_from_mesh(mesh, p=0.25, rng=np.random.default_rng(1))
    assert abs(sub.shape[0] - 0.25 * q['n_mesh_points']) <= 1


# ---- complete fits on the GPU ---------------------------------------------------------------------------------------------
def _recipe_fit(cfg, **kw):
    from ch_shrinkwrap_amd.membrane_mesh import ShrinkwrapMembrane

    class Surf(object):
        vertices, faces = cfg['vertices'], cfg['faces']
    pts = cfg['points']
    table = {'x': pts[:, 0], 'y': pts[:, 1], 'z': pts[:, 2], 'error_x': cfg['sigma'][:, 0], 'error_y': cfg['sigma'][:, 1], 'error_z': cfg['sigma'][:, 2]}
    mod = ShrinkwrapMembrane(**kw)
    return mod.execute({'surf': Surf, 'filtered_localizations': table})


@pytest.mark.gpu
def test_c1_fixed_topology_converges_to_the_sampling_floor():
    """BASELINE configs[0]: the start mesh is 20 nm off (mse_rms 20.1); 160 iterations on the fixed icosphere bring it to the sphere the
    10 000 localizations (sigma = 10 nm) were drawn from: within sigma / 2 in the reference's metric (observed 3.5 nm; sampling floor 2.8)."""
    from ch_shrinkwrap_amd import synth
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    cfg = synth.make_config('c1', seed=0)
    truth = synth.truth_cloud(cfg)
    mesh = TriMesh(cfg['vertices'].copy(), cfg['faces'])
    q0 = E.fit_quality(mesh, truth)
    assert 19.0 < q0['mse_rms'] < 21.0
    cg = ShrinkwrapMeshConjGrad(mesh, cfg['points'])
    s = 1.0 / cfg['sigma'].ravel()
    for block in range(32):
        cg.search(cfg['points'], lams=cfg['lams'], num_iters=5, sigma_inv=s)
        cg.refresh_normals()
    q = E.fit_quality(mesh, truth)
    print('c1 after 160 iterations:', q)
    assert q['mse_rms'] <= 5.0 and q['mse01'] <= 25.0 and q['mse10'] <= 25.0


def _recipe_fit_fixed(cfg, **kw):
    """the same fit with the topology held fixed (no remesher installed)"""
    from ch_shrinkwrap_amd.membrane_mesh import ShrinkwrapMembrane

    class Surf(object):
        vertices, faces = cfg['vertices'], cfg['faces']
    pts = cfg['points']
    table = {'x': pts[:, 0], 'y': pts[:, 1], 'z': pts[:, 2], 'error_x': cfg['sigma'][:, 0], 'error_y': cfg['sigma'][:, 1], 'error_z': cfg['sigma'][:, 2]}
    mod = ShrinkwrapMembrane(**kw)
    mod.remesher = None
    return mod.execute({'surf': Surf, 'filtered_localizations': table})


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize('name,scale,floor', [('c2', 1.0, None), ('c4', 0.02, 5.0)])
def test_recipe_fit_left_to_converge(name, scale, floor):
    """The recipe fit with this package's remesher (the module's default: on the device), given the iterations it needs (159 instead of the module's default 39; the start surface
    is 20 nm off, the reference's own recipes start from an isosurface of the cloud): the remesher -- the one component whose parity
    cannot be pinned (PYME's is not in the reference tree) -- must not cost the fit anything against the same fit on the FIXED start
    topology (within 1.3 x in the reference's metric), and where the fit converges it must reach sigma / 2.
    Observed (tools/experiments/r04_convergence.py): C4 x 0.02: 11.2 / 6.1 / 3.2 / 2.9 nm after 39 / 79 / 159 / 319 iterations (fixed topology:
    7.5 / 4.1 / 3.2 / 3.2); C2, a tube of 50 nm radius under curvature_weight 20: 8.1 / 7.8 / 7.8 / 7.9 nm (fixed: 6.8 / 6.6 / 6.8 / 7.1) --
    it does not get better with more iterations on either topology: the bias of the regulariser on a thin tube, not the remesher."""
    from ch_shrinkwrap_amd import synth
    cfg = synth.make_config(name, scale=scale, seed=0)
    truth = synth.truth_cloud(cfg)
    kw = dict(max_iters=159, remesh_frequency=5, curvature_weight=20.0, neck_first_iter=-1)
    q = E.fit_quality(_recipe_fit(cfg, **kw), truth)
    qf = E.fit_quality(_recipe_fit_fixed(cfg, **kw), truth)
    print(name, 'remeshed', q, 'fixed topology', qf)
    assert q['mse_rms'] <= 1.3 * qf['mse_rms']
    if floor is not None:
        assert q['mse_rms'] <= floor and qf['mse_rms'] <= floor


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize('remesher', ['device', 'builtin'])
@pytest.mark.parametrize('name,scale,limit', [('c2', 1.0, 9.0), ('c4', 0.02, 13.0)])
def test_recipe_fit_with_the_builtin_remesher(name, scale, limit, remesher):
    """The recipe module's default fit -- 39 iterations, remeshed every 5 by this package's own remesher, on the GPU (the default) or on the
    host: the same algorithm, the same fit quality (8.07 / 8.08 nm at C2, 11.25 / 11.24 at C4 x 0.02) -- from the +20 nm start surface:
    39 iterations do not converge a fit that starts 20 nm off (C2: 8.1 nm, C4 x 0.02: 11.2 nm observed: reported, not thresholds -- see
    test_recipe_fit_left_to_converge for the statement about quality); what IS asserted: the metric falls well below the start's and the
    surface after seven remeshing passes is a clean closed mesh."""
    from ch_shrinkwrap_amd import synth
    cfg = synth.make_config(name, scale=scale, seed=0)
    truth = synth.truth_cloud(cfg)
    q0 = E.fit_quality(type('M', (), {'_vertices': {'position': cfg['vertices']}, 'faces': cfg['faces']})(), truth)
    mesh = _recipe_fit(cfg, max_iters=39, remesh_frequency=5, curvature_weight=20.0, neck_first_iter=-1, remesher=remesher)
    q = E.fit_quality(mesh, truth)
    print(name, remesher, 'start', q0, 'fitted', q, 'vertices', mesh.vertices.shape[0])
    assert q0['mse_rms'] >= 20.0
    assert q['mse_rms'] <= 0.6 * q0['mse_rms']
    # the absolute limits of round 3 stay beside the relative statement (without them a drift of the remesher would go unnoticed --
    # C2 moved from 6.8 to 8.1 nm when the partitioned remesher and its rim rules came in; 8.1 / 11.2 nm observed since)
    assert q['mse_rms'] <= limit, (name, q['mse_rms'], limit)
    assert len(mesh.block_log) == 7 and np.isfinite(mesh.vertices).all()
    # closed 2-manifold after seven remeshing passes: every edge shared by exactly two faces
    f = mesh.faces
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), 1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert (cnt == 2).all()
