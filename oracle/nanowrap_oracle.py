"""
TEST INFRASTRUCTURE -- CPU oracle for the NanoWrap inner loop.  NOT part of the product path.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module; the shipped
optimiser (ch_shrinkwrap_amd.mesh_conj_grad) never does and fails loudly when the HIP library is missing.

This is a NumPy restatement -- written for this build, organised as plain functions over plain arrays --
of the reference's live per-iteration path:

    ShrinkwrapMeshConjGrad.search            /root/reference/ch_shrinkwrap/mesh_conj_grad.py:150-292
    ShrinkwrapMeshConjGrad._compute_weight_matrix4                                        :433-516
    ShrinkwrapMeshConjGrad.Afunc / Ahfunc                                                 :518-588
    ShrinkwrapMeshConjGrad._ncc / _defaults                                               :770-820, 875-890
    MembraneMesh.point_influence             /root/reference/ch_shrinkwrap/_membrane_mesh.pyx:1625-1634
    TikhonovConjugateGradient.subsearch      /root/reference/ch_shrinkwrap/conj_grad.py:183-229
    c_shrinkwrap_ah_helper                   /root/reference/ch_shrinkwrap/conj_grad_utils.c:123-167 (-> nw_oracle.c)

Parity pinning: the reference holds NO golden vectors or tests for this path, so the
oracle is pinned against outputs of the reference itself run in the build container (oracle/ref_harness.py):
tests/test_oracle_golden.py::test_live_reference_* (live, when /root/reference is present) and the committed fixtures under
tests/golden/ produced by tests/golden/make_golden.py (checked everywhere, including the GPU box).

Float-width notes (the restatement keeps the reference's mixed precision because it is observable):
  * points / vertices / weights are float32; the kd-tree works in float64 and `d` stays float64;
  * the prior `fdef` comes out float64 because of the int division at mesh_conj_grad.py:782,800;
  * S, prefs, AS, LS are float32; Hc/Gc are float32 dot products; H is accumulated in place in float32
    (conj_grad.py:208-215) and solved with float32 LAPACK gesv.
"""
import os
import ctypes
import numpy as np
import scipy.spatial

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    """ctypes handle on oracle/libnw_oracle.so (built by oracle/Makefile / __graft_entry__.build())."""
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, 'libnw_oracle.so')
        if not os.path.exists(path):
            raise RuntimeError('oracle/libnw_oracle.so missing: run `make -C oracle` or __graft_entry__.build()')
        L = ctypes.CDLL(path)
        i64, i32 = ctypes.c_int64, ctypes.c_int
        p = ctypes.c_void_p
        L.nwo_scatter_At.argtypes = [p, p, p, i64, p]
        L.nwo_lfunc.argtypes = [p, p, i32, i32, p]
        L.nwo_lhfunc.argtypes = [p, p, i32, i32, p]
        L.nwo_lwfunc.argtypes = [p, p, p, i32, i32, p]
        L.nwo_lhwfunc.argtypes = [p, p, p, i32, i32, p]
        L.nwo_vertex_area_weights.argtypes = [p, p, i32, i32, p]
        L.nwo_nearest_centroid.argtypes = [p, i64, p, i64, p, p]
        for fn in ('nwo_scatter_At', 'nwo_lfunc', 'nwo_lhfunc', 'nwo_lwfunc', 'nwo_lhwfunc',
                   'nwo_vertex_area_weights', 'nwo_nearest_centroid'):
            getattr(L, fn).restype = None
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


# ------------------------------------------------------------------------------------------------
# stage functions
# ------------------------------------------------------------------------------------------------
def face_centroids(fv, faces):
    """mesh_conj_grad.py:443 -- float32 mean of the three corner positions."""
    return fv[faces].mean(1)


def nearest_faces(centroids, points, workers=-1, brute=False):
    """mesh_conj_grad.py:451-454 -- exact Euclidean 1-NN of every point over the face centroids.
    Returns (dmean float64 (N,), face index (N,))."""
    if brute:
        idx = np.empty(points.shape[0], 'i4')
        d = np.empty(points.shape[0], 'f8')
        c = np.ascontiguousarray(centroids, 'f4')
        p = np.ascontiguousarray(points, 'f4')
        lib().nwo_nearest_centroid(_ptr(c), c.shape[0], _ptr(p), p.shape[0], _ptr(idx), _ptr(d))
        return d, idx
    tree = scipy.spatial.cKDTree(centroids)
    d, idx = tree.query(points, k=1, workers=workers)
    return d, idx


def weight_matrix(fv, faces, points, workers=-1, brute=False):
    """mesh_conj_grad.py:433-516 (_compute_weight_matrix4): nearest face per point, inverse-distance weights
    to its three vertices, rows normalised.  Returns v_idx (N,3) i4, w (N,3) f4, dmean (N,) f8."""
    cent = face_centroids(fv, faces)
    dmean, fidx = nearest_faces(cent, points, workers=workers, brute=brute)
    v_idx = faces[fidx, :]
    d = np.zeros(v_idx.shape, 'f4')
    for j in range(3):
        dv = fv[v_idx[:, j]] - points
        d[:, j] = np.sqrt(np.sum(dv * dv, 1))
    w = 1.0 / np.maximum(d, 1e-6)
    w = w / w.sum(1)[:, None]
    if np.any(np.isnan(w)):
        raise AssertionError('NaN in weight matrix')          # mesh_conj_grad.py:514
    return np.ascontiguousarray(v_idx, 'i4'), w, dmean, fidx


def apply_A(x, v_idx, w, like):
    """mesh_conj_grad.py:537-551 -- y_i = sum_j w_ij x[v_idx_ij]; accumulates in the dtype of `like` (the
    points array), corner by corner."""
    xv = x.reshape(-1, 3)
    y = np.zeros_like(like)
    for j in range(3):
        y += xv[v_idx[:, j]] * w[:, j][:, None]
    if np.any(np.isnan(y)):
        raise AssertionError('NaN in A f')                    # mesh_conj_grad.py:548
    return y.ravel()


def apply_At(r, v_idx, w, M):
    """mesh_conj_grad.py:562-588 + conj_grad_utils.c:153-162 -- z[v_idx_ij] += w_ij r_i, float32, serial."""
    out = np.zeros((M, 3), 'f4')
    rv = np.ascontiguousarray(r.reshape(-1, 3).astype('f4'))
    vi = np.ascontiguousarray(v_idx, 'i4')
    ww = np.ascontiguousarray(w, 'f4')
    lib().nwo_scatter_At(_ptr(vi), _ptr(ww), _ptr(rv), rv.shape[0], _ptr(out))
    if np.any(np.isnan(out)):
        raise AssertionError('NaN in A^T r')                  # mesh_conj_grad.py:580
    return out.ravel()


def point_influence(v_idx, w, n_points, M):
    """_membrane_mesh.pyx:1625-1634 -- || A^T 1 ||_2 per vertex (float32)."""
    s = apply_At(np.ones(3 * n_points, 'f4'), v_idx, w, M).reshape(M, 3)
    return np.sqrt((s * s).sum(1))


def ncc_prior(pos, nrm, nbr, pi):
    """mesh_conj_grad.py:770-820 (_ncc).  `nbr` is the (M, NB) table of 1-ring vertex ids (-1 padded), i.e.
    `mesh._halfedges['vertex'][mesh.vertex_neighbors]` with the -1 slots kept as -1.  The reference indexes
    with the raw -1 (wrapping to the last row) and masks afterwards; masked slots contribute exactly 0 to
    every sum there, so using row 0 for them here is equivalent."""
    mask = nbr > -1
    ms = mask.sum(1)
    vnn = np.where(mask, nbr, 0)
    with np.errstate(invalid='ignore', divide='ignore'):
        vc = (pos[vnn, :] * mask[:, :, None]).sum(1) / ms[:, None]          # f32 slot-ordered sum -> f64
        c_n = pos[vnn, :] - vc[:, None, :]                                    # f64
        n_n = nrm[vnn, :]                                                     # f32
        n_dot_n = (n_n * nrm[:, None, :]).sum(2)                              # f32
        alpha = ((c_n * n_n).sum(2)) / np.sqrt(2 * (np.maximum(n_dot_n, 0) + 1))
        alpha = (alpha * mask).sum(1) / ms
        alpha = alpha * np.minimum(pi ** 2, 1)
        vc = vc + alpha[:, None] * nrm
    vc[ms == 0, :] = pos[ms == 0, :]
    return vc


def vertex_area_weights(f, nbr):
    """mesh_conj_grad.py:724-735 (wfunc's weights) -> conj_grad_utils.c:500-548: 1/sqrt(sum_n |f_n - f_i|^2 + 1) per vertex,
    repeated over the three coordinates (float32)."""
    M, NB = nbr.shape
    w = np.zeros(3 * M, 'f4')
    lib().nwo_vertex_area_weights(_ptr(np.ascontiguousarray(f, dtype=np.float32)), _ptr(np.ascontiguousarray(nbr, dtype=np.int32)), M, NB, _ptr(w))
    return w


def subspace_solve(f0, res_m, fdef, apply_A_masked, lams, S, L=None):
    """conj_grad.py:183-229 (subsearch) with Lfuncs = ['I'] (mesh_conj_grad.py:38), or, with `L` = the diagonal weights of
    'wfunc' (mesh_conj_grad.py:724-735), Lfuncs = ['wfunc'].
    Returns fnew, cpred, wpreds, and the small matrices for tracing."""
    n_search = S.shape[1]
    c0 = (res_m * res_m).sum()
    prefs = [f0 - fdef] if L is None else [(f0 - fdef) * L]      # float64 (not the f32 copy kept by search())
    wpreds = [(p * p).sum() for p in prefs]
    AS = np.zeros((res_m.size, n_search), 'f')
    LS = np.zeros((prefs[0].size, n_search, 1), 'f')
    for k in range(n_search):
        AS[:, k] = apply_A_masked(S[:, k])
        LS[:, k, 0] = S[:, k] if L is None else S[:, k] * L
    Hc = np.dot(AS.T, AS)
    Gc = np.dot(AS.T, res_m)
    Hc0, Gc0 = Hc.copy(), Gc.copy()
    Hw = np.zeros((n_search, n_search, 1))
    Gw = np.zeros((n_search, 1))
    H, G = Hc, Gc                                         # aliases: accumulated IN PLACE in float32
    ls = LS[:, :, 0]
    Hw[:, :, 0] = np.dot(ls.T, ls)
    Gw[:, 0] = np.dot((-ls).T, prefs[0])
    l2 = lams[0] * lams[0]
    H += l2 * Hw[:, :, 0]
    G += l2 * Gw[:, 0]
    c = np.linalg.solve(H, G)
    # NB: Hc/Gc alias H/G, so the reference's "cpred" uses the regularised matrices (conj_grad.py:223)
    cpred = c0 + np.dot(np.dot(c.T, Hc), c) - np.dot(c.T, Gc)
    wpreds[0] += np.dot(np.dot(c.T, Hw[:, :, 0]), c) - np.dot(c.T, Gw[:, 0])
    fnew = f0 + np.dot(S, c)
    small = dict(Hc=Hc0, Gc=Gc0, Hw=Hw[:, :, 0].copy(), Gw=Gw[:, 0].copy(), H=H.copy(), G=G.copy(), c=c.copy(), c0=c0)
    return fnew, cpred, wpreds, small


def stop_cond(tests):
    """mesh_conj_grad.py:1009-1016."""
    if len(tests) < 3:
        return False
    a, b, c = tests[-3:]
    return (c < b) and (b < a) and (a < 1e-6)


# ------------------------------------------------------------------------------------------------
# the iteration driver
# ------------------------------------------------------------------------------------------------
class OracleResult(object):
    pass


def search(pos, nrm, nbr, faces, points, lams, num_iters=10, sigma_inv=1.0, weights=None, valid=None,
           pos_constraint=False, last_step=True, tests=None, trace=None, workers=-1, brute_nn=False, regulariser='I', data=None):
    """mesh_conj_grad.py:150-292 for one fixed-topology block.

    pos (M,3) f4 vertex positions at block start, nrm (M,3) f4 block-stale vertex normals, nbr (M,NB) i4 1-ring
    vertex ids, faces (F,3) i4, points (N,3) f4.  `tests` = history list carried by the optimiser object
    (the stop condition looks at it before the first iteration).  If `trace` is a list, one dict of
    intermediates is appended per iteration.  regulariser: 'I' (mesh_conj_grad.py:38, the live setting) or 'wfunc' (:724-735; the
    other names hand float64 data to float32 C code upstream and fail in the first iteration).  `data`: the first argument of the
    reference's search() -- the target of the residual (:164, :180-181, :222), by default the localizations the weight matrix is built
    from (`self.points`, :222 -> :433), which is what every upstream caller passes.  Returns an OracleResult with the final positions and logs."""
    M = pos.shape[0]
    N = points.shape[0]
    if valid is None:
        valid = np.ones(M, bool)
    if weights is None:
        weights = sigma_inv
    data = (points if data is None else np.asarray(data)).ravel()
    assert data.size == points.size
    if not np.isscalar(weights):
        mask = weights > 0
        weights = weights / weights.mean()
    else:
        mask = np.isfinite(data)
    if type(lams) is float:
        lams = [lams]
    fs = pos.copy()
    f = fs.ravel()
    res = 0 * data
    n_smooth = min(1, len(lams))
    n_search = n_smooth + 1
    s_size = n_search + 1
    prefs = np.zeros((f.size, n_smooth), 'f')
    S = np.zeros((f.size, s_size), 'f')
    out = OracleResult()
    out.tests = [] if tests is None else tests
    out.ress, out.prefs, out.cpred, out.wpreds = [], [], None, None
    loopcount = 0
    cur = pos.copy()              # what mesh.vertices reads: updated only at `valid` rows (:289)
    wm = None
    while (loopcount < num_iters) and (not stop_cond(out.tests)):
        loopcount += 1
        # 1. weight matrix from the current estimate, A f, residual                      (:222, :518-551)
        v_idx, w, dmean, fidx = weight_matrix(f.reshape(-1, 3), faces, points, workers=workers, brute=brute_nn)
        wm = (v_idx, w)
        Af = apply_A(f, v_idx, w, points)
        res[:] = weights * (data - Af)
        res_pre = res.copy() if trace is not None else None
        # 3. curvature prior (uses mesh.vertices == `cur`, block-stale normals)         (:224, :770-820)
        pi = point_influence(v_idx, w, N, M)
        fdef = ncc_prior(cur, nrm, nbr, pi).ravel()
        # 4. distance de-weighting with the UN-normalised sigma_inv                     (:231, :248)
        d3 = np.vstack([dmean, dmean, dmean]).T
        wd = 1.0 / (d3.ravel() * sigma_inv / 2.0 + 1)
        res *= wd
        # 5./6. search directions                                                       (:253-258)
        S[:, 0] = apply_At(res, v_idx, w, M)
        Lw = None
        if regulariser == 'wfunc':
            Lw = vertex_area_weights(f, nbr)                                             # from the CURRENT estimate (:733)
            prefs[:, 0] = (f - fdef) * Lw                                                # float64 product, float32 store (:257)
            S[:, 1] = -1.0 * (prefs[:, 0] * Lw)                                          # float32 (:258)
        elif regulariser == 'I':
            prefs[:, 0] = f - fdef
            S[:, 1] = -1.0 * prefs[:, 0]
        else:
            raise ValueError(regulariser)
        # 7. test statistic and logs                                                    (:262-271)
        test = 1.0
        test -= abs((S[:, 0] * S[:, 1]).sum() / (np.linalg.norm(S[:, 0]) * np.linalg.norm(S[:, 1])))
        out.tests.append(test)
        out.ress.append(np.linalg.norm(res))
        out.prefs.append(np.linalg.norm(prefs, axis=0))
        # 8. subspace minimisation                                                      (:274)
        fnew, out.cpred, out.wpreds, small = subspace_solve(
            f, res[mask], fdef, lambda x: apply_A(x, v_idx, w, points)[mask], lams, S[:, 0:n_search], L=Lw)
        if pos_constraint:
            fnew = fnew * (fnew > 0)
        if trace is not None:
            trace.append(dict(face=fidx.copy(), dmean=dmean.copy(), v_idx=v_idx.copy(), w=w.copy(), Af=Af.copy(),
                              res_pre=res_pre, res=res.copy(), pi=pi.copy(), fdef=fdef.copy(),
                              S=S.copy(), n_search=n_search, fnew=np.asarray(fnew).copy(), test=test, **small))
        # 9. last step becomes a search direction                                       (:281-283)
        if last_step:
            S[:, s_size - 1] = fnew - f
            n_search = s_size
        # 10. write-back                                                                 (:288-290)
        f[:] = fnew
        cur[valid] = fnew.reshape(M, 3)[valid]
    out.positions = fs
    out.mesh_positions = cur
    out.S = S
    out.res = res
    out.w = wm
    out.loopcount = loopcount
    return out
