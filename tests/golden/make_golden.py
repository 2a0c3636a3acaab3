"""
Generates the committed golden fixtures (tests/golden/*.npz) by running the REFERENCE implementation in the
build container (needs /root/reference and oracle/_ref, see oracle/ref_harness.py).  The fixtures are data only:
inputs (mesh arrays, points, sigma) and the reference's outputs / intermediates.  Re-run with

    python tests/golden/make_golden.py

The reference exposes no hooks, so intermediates are captured by wrapping bound methods of the reference
optimiser INSTANCE (`_compute_weight_matrix4`, `_ncc`, `subsearch`) with recorders; the reference code
itself runs unmodified.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_harness                                    # noqa: E402
from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere         # noqa: E402


def sphere_cloud(n, radius, sigma, seed, background=0.0, dtype='f4'):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    pts = d * radius + rng.normal(scale=sigma, size=(n, 3))
    nb = int(round(background * n))
    if nb:
        pts[:nb] = rng.uniform(-1.6 * radius, 1.6 * radius, size=(nb, 3))
    return pts.astype(dtype)


def mesh_inputs(mesh):
    return dict(vertices=mesh.vertices.copy(), faces=mesh.faces.copy(), normals=mesh.vertex_normals.copy(),
                nbr=mesh.neighbor_vertex_table(), valid=(mesh._vertices['halfedge'] != -1))


def run_reference(mesh, pts, lams, num_iters, sigma_inv, weights=None, record=False, cg=None):
    if cg is None:
        cg = ref_harness.new_reference_optimiser(mesh, pts, search_k=200, search_rad=100,
                                                 shield_sigma=float(mesh._mean_edge_length) / 2.0)
    rec = []
    if record:
        cur = {}
        o_w, o_ncc, o_sub = cg._compute_weight_matrix4, cg._ncc, cg.subsearch

        def w4(f, *a, **k):
            v_idx, w = o_w(f, *a, **k)
            cur['v_idx'], cur['w'], cur['dmean'] = v_idx.copy(), w.copy(), cg.d[:, 0].copy()
            return v_idx, w

        def ncc():
            vc = o_ncc()
            cur['fdef'] = vc.copy()
            cur['pi'] = np.asarray(mesh.point_influence).copy()
            return vc

        def sub(f0, res, fdefs, Afunc, Lfuncs, lams_, S):
            cur['res_masked'] = res.copy()
            cur['S'] = S.copy()
            cur['f0'] = f0.copy()
            fnew, cpred, wpreds = o_sub(f0, res, fdefs, Afunc, Lfuncs, lams_, S)
            cur['fnew'] = np.asarray(fnew).copy()
            cur['cpred'] = float(cpred)
            cur['wpred'] = float(wpreds[0])
            rec.append(dict(cur))
            cur.clear()
            return fnew, cpred, wpreds

        cg._compute_weight_matrix4, cg._ncc, cg.subsearch = w4, ncc, sub
    out = cg.search(pts, lams=lams, num_iters=num_iters, sigma_inv=sigma_inv, weights=weights)
    if record:
        cg._compute_weight_matrix4, cg._ncc, cg.subsearch = o_w, o_ncc, o_sub
    return cg, np.array(out), rec


def logs(cg):
    return dict(tests=np.array(cg.tests, 'f8'), ress=np.array(cg.ress, 'f8'),
                prefs=np.array([p[0] for p in cg.prefs], 'f8'), cpred=float(cg.cpred), wpred=float(cg.wpreds[0]),
                loopcount=int(cg.loopcount))


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote %s (%.1f KiB)' % (path, os.path.getsize(path) / 1024.0))


def golden_stages():
    v, f = icosphere(3, 120.0)
    mesh = TriMesh(v, f)
    inp = mesh_inputs(mesh)
    pts = sphere_cloud(2000, 100.0, 10.0, seed=1)
    sigma = np.full((2000, 3), 10.0, 'f4')
    s = 1.0 / sigma.ravel()
    cg, out, rec = run_reference(mesh, pts, [10.0], 3, s, record=True)
    arrays = dict(points=pts, sigma=sigma, lams=np.array([10.0]), positions=out, S_final=cg.S.copy(), res_final=cg.res.copy(), **inp)
    for k, v_ in logs(cg).items():
        arrays['log_' + k] = v_
    for it, r in enumerate(rec):
        for k, v_ in r.items():
            arrays['it%d_%s' % (it, k)] = v_
    save('stages_642', **arrays)


def golden_c1():
    v, f = icosphere(4, 120.0)
    mesh = TriMesh(v, f)
    inp = mesh_inputs(mesh)
    pts = sphere_cloud(10000, 100.0, 10.0, seed=0)
    sigma = np.full((10000, 3), 10.0, 'f4')
    s = 1.0 / sigma.ravel()
    cg, out, rec = run_reference(mesh, pts, [10.0], 20, s, record=True)
    arrays = dict(points=pts, sigma=sigma, lams=np.array([10.0]), positions_20=out,
                  positions_5=rec[4]['fnew'].reshape(-1, 3).astype('f4'),
                  positions_1=rec[0]['fnew'].reshape(-1, 3).astype('f4'), **inp)
    for k, v_ in logs(cg).items():
        arrays['log_' + k] = v_
    save('c1_sphere_10k', **arrays)


def golden_variants():
    v, f = icosphere(3, 120.0)
    N = 3000
    arrays = {}
    # (i) scalar sigma: the driver passes s = float(sigma) un-inverted (_membrane_mesh.pyx:1460-1461)
    mesh = TriMesh(v, f)
    arrays.update({'mesh_' + k: a for k, a in mesh_inputs(mesh).items()})
    pts = sphere_cloud(N, 100.0, 10.0, seed=2)
    arrays['points'] = pts
    cg, out, _ = run_reference(mesh, pts, [10.0], 5, 10.0)
    arrays['scalar_positions'] = out
    arrays.update({'scalar_log_' + k: a for k, a in logs(cg).items()})
    # (ii) explicit weights with zeros (mask) and a non-uniform sigma
    rng = np.random.default_rng(3)
    sigma = rng.uniform(5.0, 20.0, size=(N, 3)).astype('f4')
    wts = (1.0 / sigma.ravel()).astype('f4')
    wts[rng.random(3 * N) < 0.1] = 0
    mesh = TriMesh(v, f)
    cg, out, _ = run_reference(mesh, pts, [10.0], 5, 1.0 / sigma.ravel(), weights=wts)
    arrays['weights_sigma'] = sigma
    arrays['weights_weights'] = wts
    arrays['weights_positions'] = out
    arrays.update({'weights_log_' + k: a for k, a in logs(cg).items()})
    # (iii) two extra vertex slots that belong to no face (halfedge == -1), 10 % uniform background
    mesh = TriMesh(v, f, max_vertices=v.shape[0] + 2)
    mesh._vertices['position'][-2:] = [[500, 0, 0], [0, 500, 0]]
    pts_bg = sphere_cloud(N, 100.0, 10.0, seed=4, background=0.1)
    sig3 = np.full((N, 3), 10.0, 'f4')
    arrays.update({'holes_mesh_' + k: a for k, a in mesh_inputs(mesh).items()})
    arrays['holes_points'] = pts_bg
    cg, out, _ = run_reference(mesh, pts_bg, [10.0, 0.5], 5, 1.0 / sig3.ravel())
    arrays['holes_positions'] = out
    arrays['holes_mesh_positions'] = mesh.vertices.copy()
    arrays.update({'holes_log_' + k: a for k, a in logs(cg).items()})
    # (iv) two consecutive search() calls on one optimiser object (history of `tests` carries over,
    #      the second call restarts from the mesh's current positions)
    mesh = TriMesh(v, f)
    cg, out1, _ = run_reference(mesh, pts, [10.0], 3, 1.0 / sig3.ravel())
    cg, out2, _ = run_reference(mesh, pts, [10.0], 3, 1.0 / sig3.ravel(), cg=cg)
    arrays['twice_positions_a'] = out1
    arrays['twice_positions_b'] = out2
    arrays.update({'twice_log_' + k: a for k, a in logs(cg).items()})
    save('variants_642', **arrays)


def golden_f64_and_regulariser():
    """(a) float64 localizations: the reference's residual, A f and Gc follow the dtype of `points` (mesh_conj_grad.py:179-181,
    537-545); the MI355X path computes in float32.  Same C1-like cloud given as float64 -- once with exactly the float32 values
    (isolates the arithmetic) and once un-rounded (adds the input rounding) -- so that the deviation can be asserted.
    (b) a non-identity regulariser selected by name (mesh_conj_grad.py:36-39, 257-258): Lfuncs = Lhfuncs = ["wfunc"]."""
    v, f = icosphere(3, 120.0)
    N = 4000
    arrays = {}
    mesh = TriMesh(v, f)
    arrays.update({'mesh_' + k: a for k, a in mesh_inputs(mesh).items()})
    raw = sphere_cloud(N, 100.0, 10.0, seed=6, dtype='f8')
    p32 = raw.astype('f4')
    sig = np.full((N, 3), 10.0, 'f4')
    s = 1.0 / sig.ravel()
    arrays['points_f64_raw'] = raw
    arrays['points_f32'] = p32
    for name, pts in (('f32', p32), ('f64_same', p32.astype('f8')), ('f64_raw', raw)):
        mesh = TriMesh(v, f)
        cg, out, _ = run_reference(mesh, pts, [10.0], 5, s)
        arrays[name + '_positions'] = out
        arrays[name + '_res_dtype'] = np.array(str(cg.res.dtype))
        arrays.update({name + '_log_' + k: a for k, a in logs(cg).items()})
    # (b) Lfuncs = Lhfuncs = ["wfunc"]: the one alternate regulariser that RUNS in the reference's loop.  ("Lfunc", "Lfunc2", "Lfunc3",
    # "Lfunc4" hand `f - _ncc()` -- float64 because of _ncc's integer division -- to conj_grad_utils.c, which reads the buffer as
    # float32 (no dtype check, :286-302): NaN / AssertionError in the first iteration; tests/test_oracle_golden.py shows it live.)
    mesh = TriMesh(v, f)
    cg = ref_harness.new_reference_optimiser(mesh, p32, search_k=200, search_rad=100, shield_sigma=float(mesh._mean_edge_length) / 2.0)
    cg.Lfuncs, cg.Lhfuncs = ["wfunc"], ["wfunc"]
    cg, out, rec = run_reference(mesh, p32, [100.0], 4, s, record=True, cg=cg)
    arrays['wfunc_lams'] = np.array([100.0])
    arrays['wfunc_positions'] = out
    arrays['wfunc_S_final'] = cg.S.copy()
    arrays.update({'wfunc_log_' + k: a for k, a in logs(cg).items()})
    for it, r in enumerate(rec):
        arrays['wfunc_it%d_fnew' % it] = r['fnew']
        arrays['wfunc_it%d_S' % it] = r['S']
    save('f64_and_wfunc', **arrays)



def golden_data_target():
    """search(data, ...) with `data` different from the localizations the optimiser was built with: the weight matrix comes from
    `self.points` (mesh_conj_grad.py:222 -> :433), the residual targets `data` (:180-181, :222).  No upstream caller does this
    (_membrane_mesh.pyx:1516 passes the same array), but the reference's signature allows it."""
    v, f = icosphere(3, 120.0)
    N = 4000
    pts = sphere_cloud(N, 100.0, 10.0, seed=8)
    rng = np.random.default_rng(9)
    data = (pts + rng.normal(scale=3.0, size=pts.shape)).astype('f4')           # e.g. a drift-corrected copy of the table
    s = 1.0 / np.full(3 * N, 10.0, 'f4')
    mesh = TriMesh(v, f)
    arrays = {'mesh_' + k: a for k, a in mesh_inputs(mesh).items()}
    cg = ref_harness.new_reference_optimiser(mesh, pts, search_k=200, search_rad=100, shield_sigma=float(mesh._mean_edge_length) / 2.0)
    out = np.array(cg.search(data, lams=[10.0], num_iters=5, sigma_inv=s))
    arrays.update(points=pts, data=data, positions=out, res=np.array(cg.res).copy())
    arrays.update({'log_' + k: a for k, a in logs(cg).items()})
    save('data_target', **arrays)

def golden_lfuncs():
    """Outputs of the reference's compiled C helpers (conj_grad_utils.c) on a small sphere."""
    _, _, cgu = ref_harness.load()
    v, f = icosphere(2, 50.0)
    mesh = TriMesh(v, f, max_vertices=v.shape[0] + 1)
    nbr = mesh.neighbor_vertex_table()
    M, NB = nbr.shape
    rng = np.random.default_rng(5)
    x = rng.normal(size=3 * M).astype('f4')
    f0 = (mesh.vertices + rng.normal(scale=0.5, size=(M, 3))).astype('f4').ravel()
    arrays = dict(nbr=nbr, x=x, f0=f0)
    for name, fn, second in (('l', cgu.c_shrinkwrap_l_func, None), ('lh', cgu.c_shrinkwrap_lh_func, None),
                             ('lw', cgu.c_shrinkwrap_lw_func, f0), ('lhw', cgu.c_shrinkwrap_lhw_func, f0)):
        d = np.zeros_like(x)
        w = d if second is None else second
        fn(np.ascontiguousarray(x), nbr, w, d, 3, 0, M, NB)
        arrays['out_' + name] = d
    o = np.zeros(3 * M, 'f4')
    cgu.vertex_area_weights(f0, nbr, o, M, NB)
    arrays['out_vaw'] = o
    # scatter helper
    N = 500
    v_idx = rng.integers(0, M - 1, size=(N, 3)).astype('i4')
    w = rng.random((N, 3)).astype('f4')
    r = rng.normal(size=(N, 3)).astype('f4')
    out = np.zeros((M, 3), 'f4')
    cgu.c_shrinkwrap_ah_helper(v_idx, w, r, out)
    arrays.update(ah_v_idx=v_idx, ah_w=w, ah_r=r, ah_out=out)
    save('native_helpers', **arrays)


def golden_curvature():
    """c_curvature_grad (membrane_mesh_utils.c:915-1250) through oracle/_ref/libref_curvature.so on a noisy geodesic sphere
    with one unused vertex slot; the rand() stream of the run (srand(seed)) is recorded so it can be replayed."""
    import ctypes
    from ch_shrinkwrap_amd.trimesh import geodesic_sphere
    ref = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_ref', 'libref_curvature.so'))
    libc = ctypes.CDLL('libc.so.6')
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    v, f = geodesic_sphere(9, 40.0)
    rng = np.random.default_rng(8)
    v = (v * np.array([1.0, 0.7, 1.3]) + rng.normal(scale=0.25, size=v.shape)).astype('f4')     # ellipsoid + noise
    m = TriMesh(v, f, max_vertices=v.shape[0] + 1)
    M = m._vertices.shape[0]
    names = ['k0', 'k1', 'e0', 'e1', 'H', 'K', 'dH', 'dK', 'E', 'pE', 'dEn', 'dEdN']
    shp = {'e0': (M, 3), 'e1': (M, 3), 'dEdN': (M, 3)}
    r = {n: np.zeros(shp.get(n, (M,)), 'f4') for n in names}
    f32 = ctypes.c_float
    seed, dN, kc, kg, c0 = 77, 0.1, 0.514, -0.514, 0.01
    ref.ref_c_curvature_grad.argtypes = [ctypes.c_void_p] * 3 + [f32, f32, ctypes.c_int] + [ctypes.c_void_p] * 11 + [f32, f32, f32, ctypes.c_void_p, ctypes.c_uint]
    ref.ref_c_curvature_grad(P(m._vertices), P(m._faces), P(m._halfedges), dN, 0.0, M, P(r['k0']), P(r['k1']), P(r['e0']), P(r['e1']), P(r['H']), P(r['K']),
                             P(r['dH']), P(r['dK']), P(r['E']), P(r['pE']), P(r['dEn']), kc, kg, c0, P(r['dEdN']), seed)
    libc.srand(seed)
    libc.rand.restype = ctypes.c_int
    valid = m._vertices['halfedge'] != -1
    jit = np.zeros((M, 3))
    for i in range(M):
        if valid[i]:
            for k in range(3):
                jit[i, k] = libc.rand() / 2147483648.0
    # the same call with skip_prob > 0 (membrane_mesh_utils.c:962): `(halfedge == -1) || (r2() < skip_prob)` draws one float32 uniform per
    # used vertex, and only the vertices that are kept go on to draw their three jitter values
    skip_prob = 0.3
    rs = {n: np.zeros(shp.get(n, (M,)), 'f4') for n in names}
    ref.ref_c_curvature_grad(P(m._vertices), P(m._faces), P(m._halfedges), dN, skip_prob, M, P(rs['k0']), P(rs['k1']), P(rs['e0']), P(rs['e1']), P(rs['H']), P(rs['K']),
                             P(rs['dH']), P(rs['dK']), P(rs['E']), P(rs['pE']), P(rs['dEn']), kc, kg, c0, P(rs['dEdN']), seed)
    libc.srand(seed)
    skip_u = np.ones(M, 'f4')
    jit_s = np.zeros((M, 3))
    for i in range(M):
        if valid[i]:
            skip_u[i] = np.float32(libc.rand()) / np.float32(2147483648.0)
            if not (skip_u[i] < np.float32(skip_prob)):
                for k in range(3):
                    jit_s[i, k] = libc.rand() / 2147483648.0
    save('curvature_geo9', vertices=m.vertices.copy(), faces=m.faces.copy(), normals=m.vertex_normals.copy(), face_area=m._faces['area'].copy(),
         params=np.array([dN, kc, kg, c0]), jitter=jit, skip_prob=np.float32(skip_prob), skip_u=skip_u, skip_jitter=jit_s,
         **{'out_' + n: a for n, a in r.items()}, **{'skip_out_' + n: a for n, a in rs.items()})


def golden_sdf_shapes():
    """Signed distances of the reference's own CSG shapes (shape.py / sdf.py) at seeded sample positions: pins the
    synthetic-cloud generator (ch_shrinkwrap_amd/synth.py) behind the BASELINE.json configs."""
    sh = ref_harness.load_shapes()
    rng = np.random.default_rng(2024)
    P = rng.uniform(-700, 700, size=(3000, 3))
    P[:, 2] *= 0.3
    out = dict(points=P)
    out['sphere_r100'] = sh.Sphere(radius=100.0).sdf(P.T)
    out['capsule_c2'] = sh.Capsule([0, -500, 0], [0, 500, 0], 50.0).sdf(P.T)
    out['two_lobe_c3'] = sh.UnionShape(sh.Sphere(radius=300, centroid=np.array([-250., 0, 0])),
                                       sh.Sphere(radius=300, centroid=np.array([250., 0, 0])), k=50).sdf(P.T)
    out['round_box'] = sh.Box(np.array([66, 83, 25.]), 25.0).sdf(P.T)
    out['sheet'] = sh.Sheet(np.array([226, 200, 100 / 3]), 100 / 3).sdf(P.T)
    out['three_way_junction'] = sh.ThreeWayJunction(300, 50, k=20).sdf(P.T)
    out['er_sim2'] = sh.ERSim2().sdf(P.T)
    out['difference'] = sh.DifferenceShape(sh.Capsule([-40, 0, -100], [-40, 0, 100], 50.0), sh.Sphere(radius=200.0), k=25).sdf(P.T)
    save('sdf_shapes', **out)


def golden_fit_quality():
    """The reference's fit-quality metric (evaluation_utils.points_from_mesh :35-150 with the recipe's defaults dx_min = 5, p = 1;
    average_squared_distance :153-180 -> mse01, mse10, mse_rms as recipe_modules/surface_feature_extraction.py:133-138 forms them) on a
    stretched, shifted icosphere against seeded points near it, and on a coarser grid.  Pins ch_shrinkwrap_amd/evaluation.py."""
    ev = ref_harness.load_evaluation_utils()
    v, f = icosphere(2, 1.0)
    v = (v * np.array([60.0, 45.0, 80.0], 'f4') + np.array([300.0, -120.0, 40.0], 'f4')).astype('f4')
    mesh = TriMesh(v, f)
    rng = np.random.default_rng(77)
    d = rng.normal(size=(4000, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    truth = (d * np.array([58.0, 47.0, 77.0]) + np.array([300.0, -120.0, 40.0]) + rng.normal(scale=0.5, size=d.shape)).astype('f4')
    out = dict(vertices=v, faces=f, truth=truth)
    for tag, dx in (('dx5', 5.0), ('dx11', 11.0)):
        np.random.seed(5)                                             # (p = 1: the reference's subsample is a permutation of all points)
        pts = np.asarray(ev.points_from_mesh(mesh, dx_min=dx, p=1.0))
        pts = pts[np.lexsort((pts[:, 2], pts[:, 1], pts[:, 0]))]
        m0, m1 = ev.average_squared_distance(pts, truth)
        out['points_' + tag] = pts
        out['mse_' + tag] = np.array([m0, m1, np.sqrt((m0 + m1) / 2)])
    save('fit_quality', **out)


if __name__ == '__main__':
    if not ref_harness.available():
        raise SystemExit('reference not available here')
    if len(sys.argv) > 1:                                             # python make_golden.py fit_quality ... : only the named fixtures
        for name in sys.argv[1:]:
            globals()['golden_' + name]()
        raise SystemExit(0)
    golden_stages()
    golden_c1()
    golden_variants()
    golden_f64_and_regulariser()
    golden_data_target()
    golden_lfuncs()
    golden_curvature()
    golden_sdf_shapes()
    golden_fit_quality()
