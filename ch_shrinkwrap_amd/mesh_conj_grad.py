"""
Host-side mirror of the reference optimiser class for the NanoWrap inner loop, backed by the HIP library.

Drop-in seam: the reference's outer loop imports the optimiser by name,

    from ch_shrinkwrap.mesh_conj_grad import ShrinkwrapMeshConjGrad          (_membrane_mesh.pyx:1428)
    cg = ShrinkwrapMeshConjGrad(mesh, points, search_k=, search_rad=, shield_sigma=)      (:1510-1512)
    vp = cg.search(points, lams=lams, num_iters=n_it, sigma_inv=s, weights=weights)       (:1516-1517)

and later reads cg.S, cg.res, cg.points, cg.Ahfunc(...), cg.w (:1563-1634).  This class keeps that constructor,
`search` signature, attribute surface, log lists (tests / ress / prefs / cpred / wpreds / loopcount) and error
behaviour (AssertionError on NaN, numpy.linalg.LinAlgError on a singular subspace system), while every array
of the iteration lives in HBM inside one `nw_ctx` (include/nanowrap.h) for the whole block.

Differences that are deliberate and documented:
  * float32 only: points/sigma/weights are cast to float32 on upload (the reference follows the dtype of `points`);
  * `data` passed to search() must be the localizations the optimiser was constructed with (the only way the
    reference ever calls it, _membrane_mesh.pyx:1516);
  * `mesh._vertices['position']` is written back (and `mesh._initialize_curvature_vectors()` called) once at the
    end of search() instead of after every iteration (mesh_conj_grad.py:289-290) -- same final state;
  * `defaults` is accepted and ignored, exactly as the reference overwrites it at mesh_conj_grad.py:224;
  * the unused point kd-tree of the reference constructor (mesh_conj_grad.py:127-130) is not built.
There is no CPU fallback: without libnanowrap_hip.so and a GPU the constructor raises.
"""
import ctypes
import os
import numpy as np

from . import _lib as nw


class NativeContext(object):
    """Owns one nw_ctx.  Shared between consecutive optimiser objects of one fit so that the localizations stay
    resident in HBM across remesh blocks (the reference builds a new optimiser per block, _membrane_mesh.pyx:1510)."""

    def __init__(self, device=0, stream=None):
        self.L = nw.load()
        self.h = ctypes.c_void_p()
        code = self.L.nw_create(int(device), ctypes.byref(self.h))
        if code != nw.NW_OK:
            raise nw.NanoWrapError(code, 'nw_create failed (is a MI355X visible? HIP extension present?)')
        if stream is not None:
            self.check(self.L.nw_set_stream(self.h, ctypes.c_void_p(int(stream))))
        self.device = int(device)
        self.points_key = None
        self._keep = []

    def check(self, code):
        nw.check(self.L, self.h, code)

    def close(self):
        if self.h:
            self.L.nw_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# NW_ROWS_ASYNC=0 (developer knob): the vertex records are complete when search() returns, as until round 4
_ROWS_ASYNC = os.environ.get('NW_ROWS_ASYNC', '1') != '0'


def _as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class ShrinkwrapMeshConjGrad(object):
    """MI355X-native counterpart of ch_shrinkwrap.mesh_conj_grad.ShrinkwrapMeshConjGrad (mesh_conj_grad.py:20)."""

    def __init__(self, mesh, points, sigma=None, search_k=200, search_rad=100, shield_sigma=None, use_octree=False,
                 device=0, native=None, stream=None, reuse_device_mesh=False, device_tables=False):
        # TikhonovConjugateGradient.__init__ (conj_grad.py:35-43)
        self.tests, self.ress, self.prefs = [], [], []
        self.Lfuncs, self.Lhfuncs = ["I"], ["I"]            # mesh_conj_grad.py:38
        self.cpred, self.wpreds = None, None
        self.loopcount = 0
        self.mesh = mesh
        self.sigma = sigma
        self.search_k = min(search_k, points.shape[0])      # kept for API compatibility; unused on the live path
        self.search_rad = max(search_rad, 1.0)
        self._use_octreee = use_octree
        self.nn_max_ring = 0
        self.mean_dist = 0.0
        self._raw_logs = []
        self._iter_logs = []
        self._fs_pool = []

        # (a mesh that builds its topology records on demand -- trimesh.TriMesh -- is asked in a way that does not trigger the build)
        records = mesh._position_records() if hasattr(mesh, '_position_records') else mesh._vertices
        self._mesh_vertex_mask = mesh.valid_vertex_mask() if hasattr(mesh, 'valid_vertex_mask') else records['halfedge'] != -1    # :44
        self._all_valid = bool(self._mesh_vertex_mask.all())
        self._vertices_view = records['position']                                      # :46 (view)
        self.M = self._vertices_view.shape[0]
        self.dims = self._vertices_view.shape[1]
        self.shape = self._vertices_view.shape
        self.faces = mesh.faces                                                        # :47
        # device_tables (the driver's block loop, membrane_mesh.MembraneMesh): the library builds the 1-ring table and the vertex normals
        # itself from faces and positions (nw_set_mesh with nbr = nrm = NULL; the same ring order as the host substrate's) -- no table is
        # built or uploaded here; `vertex_neighbors` is then made on first use
        self._device_tables = bool(device_tables) and getattr(mesh, '_accepts_deferred_rows', False) and os.environ.get('NW_HOST_TABLES', '0') != '1'
        self._vertex_neighbors = None
        if not self._device_tables:
            self._vertex_neighbors = self._host_ring_table()                           # :50-54
        self.N = int(records.dtype['neighbors'].shape[0])

        self._records = (lambda: mesh._position_records()) if hasattr(mesh, '_position_records') else (lambda: mesh._vertices)
        self._native = native if native is not None else NativeContext(device, stream)
        self._L = self._native.L
        self._h = self._native.h
        self.points = points
        topo = (id(mesh), self.M, int(np.asarray(self.faces).shape[0]), self.N)
        if reuse_device_mesh and getattr(self._native, 'mesh_key', None) == topo:
            # same mesh object, same topology, positions/normals already current on the device (refresh_normals()):
            # a new optimiser only restarts the logs and the stop-condition history
            self._native.check(self._L.nw_reset_history(self._h))
        else:
            self._upload_mesh()
            self._native.mesh_key = topo
        self._weights_key = None
        self._cache = {}
        self.fs = None
        self.f = None
        self.mask = None

    # -- uploads ------------------------------------------------------------------------------
    def _host_ring_table(self):
        mesh = self.mesh
        if hasattr(mesh, 'neighbor_vertex_table'):                                     # (native table builder of the substrate)
            return mesh.neighbor_vertex_table()
        n = mesh._halfedges['vertex'][mesh._vertices['neighbors']]
        n[mesh._vertices['neighbors'] == -1] = -1
        return np.ascontiguousarray(n, dtype=np.int32)

    @property
    def vertex_neighbors(self):
        if self._vertex_neighbors is None:
            self._vertex_neighbors = self._host_ring_table()
        return self._vertex_neighbors

    @vertex_neighbors.setter
    def vertex_neighbors(self, table):
        self._vertex_neighbors = table

    def _upload_mesh(self):
        pos = _as_f32(self._vertices_view)
        faces = np.ascontiguousarray(self.faces, dtype=np.int32)
        valid = np.ascontiguousarray(self._mesh_vertex_mask, dtype=np.uint8)
        if self._device_tables:
            self._native.check(self._L.nw_set_mesh(self._h, nw.ptr(pos), None, None, nw.ptr(valid), nw.ptr(faces), pos.shape[0], faces.shape[0], self.N))
            return
        nrm = _as_f32(self.mesh.vertex_normals)
        self._native.check(self._L.nw_set_mesh(self._h, nw.ptr(pos), nw.ptr(nrm), nw.ptr(self.vertex_neighbors), nw.ptr(valid),
                                               nw.ptr(faces), pos.shape[0], faces.shape[0], self.vertex_neighbors.shape[1]))

    @property
    def points(self):
        return self._points

    @points.setter
    def points(self, pts):
        self._points = pts
        self._points_f32 = _as_f32(pts)
        if self._points_f32.ndim != 2 or self._points_f32.shape[1] != 3:
            raise ValueError('points must be (N, 3)')

    @property
    def vertices(self):
        return self._vertices

    def _upload_points(self, sigma_inv, weights, prenormalized=None):
        """search() weight handling, mesh_conj_grad.py:156-164, done on the device at upload.
        `prenormalized`: (3N,) weights already divided by the mean over ALL ranks (parallel.run_search)."""
        N3 = self._points_f32.size
        key = (id(self._points), id(sigma_inv) if not np.isscalar(sigma_inv) else float(sigma_inv),
               None if weights is None else (id(weights) if not np.isscalar(weights) else float(weights)),
               None if prenormalized is None else 'pre')
        if self._native.points_key == key:
            return
        s_arr, s_sc = None, 1.0
        if np.isscalar(sigma_inv):
            s_sc = float(sigma_inv)
        else:
            s_arr = _as_f32(np.asarray(sigma_inv).ravel())
            if s_arr.size != N3:
                raise ValueError('sigma_inv must be a scalar or have 3N entries')
        w_arr, w_sc, mode = None, 1.0, nw.NW_WEIGHTS_FROM_SIGMA_INV
        if weights is not None:
            if np.isscalar(weights):
                mode, w_sc = nw.NW_WEIGHTS_SCALAR, float(weights)
            else:
                mode = nw.NW_WEIGHTS_ARRAY
                w_arr = _as_f32(np.asarray(weights).ravel())
                if w_arr.size != N3:
                    raise ValueError('weights must be a scalar or have 3N entries')
        if prenormalized is not None:
            mode = nw.NW_WEIGHTS_PRENORMALIZED
            w_arr = _as_f32(np.asarray(prenormalized).ravel())
        self._native.check(self._L.nw_set_points(self._h, nw.ptr(self._points_f32), self._points_f32.shape[0], nw.ptr(s_arr), s_sc,
                                                 mode, nw.ptr(w_arr), w_sc))
        self._native.points_key = key
        self._native.data_key = None            # (nw_set_points drops a residual target set for the previous upload)
        self._native._keep = [self._points, sigma_inv, weights]      # keep ids alive while they key the cache
        # host view of the mask for API parity (`cg.mask`)
        if mode in (nw.NW_WEIGHTS_ARRAY, nw.NW_WEIGHTS_PRENORMALIZED):
            self.mask = w_arr > 0
        elif mode == nw.NW_WEIGHTS_FROM_SIGMA_INV and s_arr is not None:
            self.mask = s_arr > 0
        else:
            self.mask = np.isfinite(self._points_f32.ravel())

    def _regulariser_flag(self):
        """Lfuncs / Lhfuncs are selected by name (mesh_conj_grad.py:36-39, 257-258).  ["I"] is the live setting; ["wfunc"] is the
        one alternative that runs upstream.  "Lfunc", "Lfunc2", "Lfunc3", "Lfunc4" fail in the reference's first iteration (they
        hand the float64 `f - _ncc()` to conj_grad_utils.c, which reads it as float32 and asserts on the NaNs), so there is no
        behaviour to reproduce; nw_lfunc() offers those operators on their own."""
        L, Lh = list(self.Lfuncs), list(self.Lhfuncs)
        if L == ["I"] and Lh == ["I"]:
            return 0
        if L == ["wfunc"] and Lh == ["wfunc"]:
            return nw.NW_FLAG_WFUNC
        raise NotImplementedError('Lfuncs=%r / Lhfuncs=%r: only ["I"] and ["wfunc"] run in the reference\'s loop' % (L, Lh))

    # -- the hot path ---------------------------------------------------------------------------
    def search(self, data, lams, defaults=None, num_iters=10, weights=None, sigma_inv=1.0, pos=False, last_step=True,
               comm_flags=0, prenormalized=None, to_host=True):
        """mesh_conj_grad.py:150-292.  Returns the (M,3) float32 vertex estimate.

        comm_flags / prenormalized / to_host are this rank's part of a multi-GPU block (parallel.run_search over a NativeComm): one of
        NW_FLAG_COMM_*, the weights divided by the mean over ALL ranks, and whether this rank's result goes to the host mesh (a sharded
        mesh gathers the whole mesh afterwards instead)."""
        # `data` is the target of the residual (mesh_conj_grad.py:164, 180-181, 222); the weight matrix always comes from the
        # localizations the optimiser was built with (:222 -> :433).  Upstream passes the same array for both (_membrane_mesh.pyx:1516).
        target = None
        if data is not self._points:
            d = np.asarray(data)
            if d.size != self._points_f32.size:
                raise ValueError('data must have as many entries as the localizations (%d)' % self._points_f32.size)
            if d.shape != np.asarray(self._points).shape or not np.array_equal(d, self._points):
                target = _as_f32(d.reshape(-1, 3))
        if type(lams) is float or np.isscalar(lams):
            lams = [float(lams)]
        lams_a = np.ascontiguousarray(lams, dtype=np.float32)
        self._upload_points(sigma_inv, weights, prenormalized)
        dkey = None if target is None else (id(data), self._native.points_key)
        if getattr(self._native, 'data_key', None) != dkey or (target is not None and not getattr(self, '_data_is_static', False)):
            self._native.check(self._L.nw_set_data(self._h, nw.ptr(target)))
            self._native.data_key = dkey
        num_iters = int(num_iters)
        flags = (nw.NW_FLAG_POSITIVITY if pos else 0) | (0 if last_step else nw.NW_FLAG_NO_LAST_STEP) | self._regulariser_flag() | int(comm_flags)
        logs = (nw.IterLog * max(num_iters, 1))()
        lc = ctypes.c_int(0)
        self._cache = {}
        if not to_host:
            code = self._L.nw_search(self._h, nw.ptr(lams_a), lams_a.size, num_iters, flags, None, logs, ctypes.byref(lc))
            self._native.check(code)
            self._consume_logs(logs, lc.value)
            self._accumulate_stage_ms()
            return None
        # write-back (mesh_conj_grad.py:288-290) inside the call: the library copies the positions out in slices (the (M,3) result and
        # the strided mesh._vertices['position'] rows of the valid vertices) while the rest of the transfer is still in flight
        out = self._result_buffer()
        # a mesh that knows about deferred rows (trimesh.TriMesh): its records are reached WITHOUT waiting for the previous block's rows --
        # the library orders the two copies itself -- and this block's rows may be written while the caller goes on (NW_FLAG_ROWS_ASYNC)
        records = self.mesh.__dict__.get('_vertex_records') if getattr(self.mesh, '_accepts_deferred_rows', False) else None
        posv = (records if records is not None else self._records())['position']
        direct = posv.dtype == np.float32 and posv.strides[1] == 4 and posv.strides[0] >= 12
        deferred = records is not None and direct and _ROWS_ASYNC and num_iters > 0
        self._native.check(self._L.nw_set_write_back(self._h, ctypes.c_void_p(posv.ctypes.data) if direct else None, posv.strides[0] if direct else 0))
        code = self._L.nw_search(self._h, nw.ptr(lams_a), lams_a.size, num_iters, flags | (nw.NW_FLAG_ROWS_ASYNC if deferred else 0), nw.ptr(out), logs, ctypes.byref(lc))
        if deferred:
            self.mesh.__dict__['_rows_pending'] = self._flush_rows      # (before anything can raise: the rows may be in flight)
        self._native.check(self._L.nw_set_write_back(self._h, None, 0))
        self._native.check(code)
        self._consume_logs(logs, lc.value)
        self._accumulate_stage_ms()
        if not direct:                          # exotic vertex layout: let NumPy do the strided copy
            np.copyto(posv, out, where=self._mesh_vertex_mask[:, None])
        self.fs = out
        self.f = self.fs.ravel()
        self.mesh._initialize_curvature_vectors()
        return np.real(self.fs)

    @property
    def _vertices(self):
        """the view of the mesh's position rows the reference keeps (mesh_conj_grad.py:46) -- complete (see _flush_rows)"""
        if self.mesh.__dict__.get('_rows_pending') is not None:
            self._flush_rows()
        return self._vertices_view

    def _flush_rows(self):
        """Wait for the host threads that fill the mesh's vertex records behind a block (NW_FLAG_ROWS_ASYNC); trimesh.TriMesh calls
        this before anybody reads `_vertices`."""
        self.mesh.__dict__['_rows_pending'] = None
        self._native.check(self._L.nw_synchronize(self._h))

    def synchronize(self):
        """Device stream drained and host-side copies finished (nw_synchronize)."""
        self._flush_rows()

    def _consume_logs(self, logs, executed):
        self.loopcount = executed
        for i in range(executed):
            L = logs[i]
            self.tests.append(np.float32(L.test))
            self.ress.append(np.float32(L.res_norm))
            self.prefs.append(np.array([L.prefs_norm], np.float32))
            self.cpred = np.float32(L.cpred)
            self.wpreds = [np.float64(L.wpred)]
            self.nn_max_ring = max(self.nn_max_ring, int(L.nn_max_ring))
            self.mean_dist = float(L.mean_dist)
            self.max_dist = max(getattr(self, 'max_dist', 0.0), float(L.max_dist))
        if executed:
            self._raw_logs.append((logs, executed))             # expanded on demand (iter_logs): keeps the per-block host time short

    @property
    def iter_logs(self):
        """Per-iteration records of the normal equations (H, G, c), norms and query diagnostics, one dict per executed iteration."""
        for logs, executed in self._raw_logs:
            for i in range(executed):
                L = logs[i]
                self._iter_logs.append(dict(test=L.test, res_norm=L.res_norm, prefs_norm=L.prefs_norm, cpred=L.cpred, wpred=L.wpred,
                                            c=np.array(L.c[:]), H=np.array(L.H[:]).reshape(3, 3), G=np.array(L.G[:]),
                                            mean_dist=L.mean_dist, n_search=int(L.n_search), nn_max_ring=int(L.nn_max_ring)))
        self._raw_logs = []
        return self._iter_logs

    def _result_buffer(self):
        """(M,3) float32 array for the next result.  A fresh 2.4 MB array costs ~600 page faults per block; an array handed out
        earlier is recycled ONLY when nothing but this pool still refers to it (the caller dropped the result and `self.fs` has
        moved on), so a result the caller kept is never overwritten."""
        import sys
        pool = self._fs_pool
        for i in range(len(pool)):
            # an idle array is referenced by the pool list and by getrefcount's argument, nothing else
            if sys.getrefcount(pool[i]) <= 2 and pool[i].shape == (self.M, 3):
                return pool[i]
        b = np.empty((self.M, 3), np.float32)
        pool.append(b)
        if len(pool) > 3:
            del pool[0]
        return b

    def _finish(self):
        """write-back (mesh_conj_grad.py:288-290): one D2H into pinned memory, then the (M,3) result array and the strided
        mesh._vertices['position'] rows (valid vertices only) are filled by the library."""
        out = self._result_buffer()
        posv = self._records()['position']
        stride = posv.strides[0]
        if posv.dtype == np.float32 and posv.strides[1] == 4 and stride >= 12:
            self._native.check(self._L.nw_write_back(self._h, nw.ptr(out), ctypes.c_void_p(posv.ctypes.data), stride))
        else:                                   # exotic vertex layout: let NumPy do the strided copy
            self._native.check(self._L.nw_write_back(self._h, nw.ptr(out), None, 0))
            np.copyto(posv, out, where=self._mesh_vertex_mask[:, None])
        self.fs = out
        self.f = self.fs.ravel()
        self.mesh._initialize_curvature_vectors()

    # -- state the mesh reads back (_membrane_mesh.pyx:1563-1634) ----------------------------------
    def _get(self, what, shape, dtype):
        if what in self._cache:
            return self._cache[what]
        a = np.empty(shape, dtype)
        self._native.check(self._L.nw_get(self._h, what, nw.ptr(a), a.nbytes))
        self._cache[what] = a
        return a

    @property
    def S(self):
        return self._get(nw.NW_ARR_S, (3 * self.M, 3), np.float32)

    @property
    def res(self):
        return self._get(nw.NW_ARR_RES, (self._points_f32.size,), np.float32)

    @property
    def w(self):
        n = self._points_f32.shape[0]
        return (self._get(nw.NW_ARR_VIDX, (n, 3), np.int32), self._get(nw.NW_ARR_W, (n, 3), np.float32))

    @property
    def d(self):
        dm = self._get(nw.NW_ARR_DIST, (self._points_f32.shape[0],), np.float32).astype(np.float64)
        return np.vstack([dm, dm, dm]).T                                           # mesh_conj_grad.py:483

    @property
    def nearest_face(self):
        return self._get(nw.NW_ARR_FACE, (self._points_f32.shape[0],), np.int32)

    @property
    def fdef(self):
        return self._get(nw.NW_ARR_FDEF, (self.M, 3), np.float32)

    @property
    def point_influence(self):
        return self._get(nw.NW_ARR_PI, (self.M,), np.float32)

    def Afunc(self, f):
        """mesh_conj_grad.py:518-551 with the cached weight matrix."""
        x = _as_f32(np.asarray(f).ravel())
        if x.size != 3 * self.M:
            raise ValueError('Afunc expects 3M values')
        y = np.empty(self._points_f32.size, np.float32)
        self._native.check(self._L.nw_apply_A(self._h, nw.ptr(x), nw.ptr(y)))
        return y

    def Ahfunc(self, f):
        """mesh_conj_grad.py:553-588."""
        r = _as_f32(np.asarray(f).ravel())
        if r.size != self._points_f32.size:
            raise ValueError('Ahfunc expects 3N values')
        z = np.empty(3 * self.M, np.float32)
        self._native.check(self._L.nw_apply_At(self._h, nw.ptr(r), nw.ptr(z)))
        if np.any(np.isnan(z)):
            raise AssertionError('NaN in A^T r')                                    # :580
        return z

    def I(self, f):
        return f                                                                    # :912-914

    # alternate regularisers (default off; mesh_conj_grad.py:590-736 -> conj_grad_utils.c)
    def _lfunc(self, kind, f, f0=None):
        x = _as_f32(np.asarray(f).ravel())
        d = np.zeros_like(x)
        f0a = None if f0 is None else _as_f32(np.asarray(f0).ravel())
        self._native.check(self._L.nw_lfunc(self._h, kind, nw.ptr(x), nw.ptr(f0a), nw.ptr(d)))
        if np.any(np.isnan(d)):
            raise AssertionError('NaN in regulariser output')
        return d

    def Lfunc(self, f):
        return self._lfunc(0, f)

    def Lhfunc(self, f):
        return self._lfunc(1, f)

    def Lfunc3(self, f):
        return self._lfunc(2, f, self.f if self.f is not None else self._vertices)

    def Lhfunc3(self, f):
        return self._lfunc(3, f, self.f if self.f is not None else self._vertices)

    def vertex_area_weights(self, f=None):
        return self._lfunc(4, self._vertices if f is None else f)

    def start_guess(self, data):
        return self._vertices.copy()                                                # :1002-1007

    def _stop_cond(self):
        if len(self.tests) < 3:                                                     # :1009-1016
            return False
        a, b, c = self.tests[-3:]
        return (c < b) and (b < a) and (a < 1e-6)

    def refresh_normals(self, fetch=True):
        """Block-boundary refresh for an unchanged topology (_membrane_mesh.pyx:1524-1527) on the device: vertex normals are
        recomputed from the device-resident positions and kept in HBM for the next block.  fetch=True writes them to mesh.vertex_normals at
        once; fetch=False (the driver's block loop, with a mesh whose `vertex_normals` property can fetch on first use) leaves them on the
        device until somebody asks."""
        lazy = not fetch and hasattr(self.mesh, '_normals_stale')
        nrm = None if lazy else np.empty((self.M, 3), np.float32)
        self._native.check(self._L.nw_refresh_normals(self._h, nw.ptr(nrm), 0.0))
        if lazy:
            self.mesh._normals_stale = self._fetch_normals
            return None
        self._records()['normal'][:] = nrm
        if hasattr(self.mesh, '_normals_stale'):
            self.mesh._normals_stale = False
        return nrm

    def refresh_normals_lazy(self):
        return self.refresh_normals(fetch=False)

    def _fetch_normals(self):
        nrm = np.empty((self.M, 3), np.float32)
        self._native.check(self._L.nw_get(self._h, nw.NW_ARR_NRM, nw.ptr(nrm), nrm.nbytes))
        self._records()['normal'][:] = nrm

    # -- timing hooks for bench.py ------------------------------------------------------------------
    def set_profiling(self, level=2):
        """0/False off; 1 = HIP events around every NN query launch; 2/True = around every stage (each event pair serialises the
        stream for a few microseconds; 1 and 2 launch every kernel from the host); 4 = the LAST iteration of each block is launched
        from the host with its NN query bracketed (while the replayed hipGraph of everything before it is still running: what bench.py keeps
        on in its timed region: one event pair per block instead of one per iteration).  (3, the two-half-graphs form of ABI 2, was removed.)"""
        level = 2 if level is True else int(level)
        self._native.check(self._L.nw_set_profiling(self._h, level))
        self._profiling = level > 0
        self._profiling_level = level
        self._profiled_stages = ('nn',) if level in (1, 4) else None          # (levels 1 and 4 time nothing else)
        self.stage_ms_total = {k: (0.0, 0) for k in ('total', 'grid', 'nn', 'attract', 'prior', 'as', 'update', 'fixup')}

    def _accumulate_stage_ms(self):
        # HIP-event timings of the last search() (per stage: summed ms, number of timed spans), accumulated over calls
        if not getattr(self, '_profiling', False):
            return
        for k, (ms, n) in self.stage_ms(getattr(self, '_profiled_stages', None)).items():
            a, b = self.stage_ms_total[k]
            self.stage_ms_total[k] = (a + ms, b + n)

    def separate_attraction(self, on=True):
        """Per-stage timings want the attraction step as a launch of its own (k_attract behind the query) instead of in workgroups appended to
        the query launch (the default; bit-identical results): nw_debug(what = 3)."""
        self._native.check(self._L.nw_debug(self._h, 3, None, None, 1 if on else 0, None))

    def optimize_layout(self):
        """do the library's pending one-off set-up now (the projection re-sort after the first block) instead of at the next search()"""
        self._native.check(self._L.nw_optimize_layout(self._h))

    def nn_stats(self):
        """developer counters of the nearest-face query since the previous call (first call: switches them on)"""
        out = (ctypes.c_int64 * 17)()
        self._native.check(self._L.nw_debug(self._h, 0, out, None, 0, None))
        names = ['candidates', 'rows_nonempty', 'rows_visited', 'cells_tested', 'cells_visited', 'box_rows', 'rounds', 'max_wave_cycles_16', 'stream_cycles_16', 'wave_cycles_16',
                 'prologue_cycles_16', 'tail_cycles_16', 'lane_cells', 'lane_cells_max', 'lane_candidates', 'lane_candidates_max', 'items']
        return dict(zip(names, [int(v) for v in out]))

    def stage_ms(self, only=None):
        names = ['total', 'grid', 'nn', 'attract', 'prior', 'as', 'update', 'fixup']
        out = {}
        for i, nme in enumerate(names):
            if only is not None and nme not in only:
                continue
            ms, n = ctypes.c_double(0), ctypes.c_int64(0)
            self._native.check(self._L.nw_stage_ms(self._h, i, ctypes.byref(ms), ctypes.byref(n)))
            out[nme] = (ms.value, n.value)
        return out
