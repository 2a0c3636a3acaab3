"""
Outer-loop driver for the NanoWrap hot path: the host-side mirror of

    MembraneMesh.__init__                 /root/reference/ch_shrinkwrap/_membrane_mesh.pyx:79-120
    MembraneMesh.opt_conjugate_gradient   :1427-1560   (block scheduler around ShrinkwrapMeshConjGrad.search)
    MembraneMesh.shrink_wrap              :1641-1669
    diagnostics S0..S3, _S0, point_dis, rms_point_sc, point_influence            :1563-1634

built on this package's own half-edge substrate (trimesh.TriMesh) because the reference's base class is PYME's
TriangleMesh (third party, absent).  What is in scope is the DRIVER: sigma handling, lambda, block sizes, one
optimiser per block, the post-block normal refresh, the remesh target-length schedule and `truncate_at`.  What is
PYME's part is the topology surgery at block boundaries -- `remesh`, the deletion half of `remove_necks`,
`punch_holes`, `remove_extra_short_edges`: they are hooks (`remesher`, `neck_remover`, `hole_puncher`, `edge_cleaner`).
For `remesh` this package ships its own implementation of the published algorithm (`remesher='builtin'`, remesh.py /
csrc/remesh.cpp); when no hook is installed the topology is held fixed and that is logged once.  The numerical path of every block is the HIP library; there is no CPU fallback.
"""
import math
import os
import numpy as np

from .trimesh import TriMesh
from .mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext

KBT = 0.0257                                             # _membrane_mesh.pyx:22
DESCENT_METHODS = ['conjugate_gradient', 'skeleton']     # :19
DEFAULT_DESCENT_METHOD = 'conjugate_gradient'            # :20


class MembraneMesh(TriMesh):
    def __init__(self, vertices=None, faces=None, mesh=None, device=0, **kwargs):
        if mesh is not None:
            vertices, faces = np.asarray(mesh.vertices), np.asarray(mesh.faces)
        TriMesh.__init__(self, vertices, faces)
        # defaults: _membrane_mesh.pyx:82-117
        self.kc = 20.0 * KBT
        self.kg = -20.0 * KBT
        self.a = 1.0
        self.c = 1.0
        self.c0 = 0.0
        self.step_size = 1
        self.beta_1 = 0.8
        self.beta_2 = 0.7
        self.eps = 1e-8
        self.max_iter = 250
        self.remesh_frequency = 100
        self.delaunay_remesh_frequency = 150
        self.delaunay_eps = 0.0
        self.search_k = 200
        self.search_rad = 100
        self.skip_prob = 0.0
        self.shrink_weight = 0.0
        self.smooth_curvature = False
        self.vertex_properties.extend(['E', 'curvature_principal0', 'curvature_principal1', 'point_dis', 'rms_point_sc', 'point_influence'])
        self.vertex_vector_properties.extend(['S0', 'S1', 'S2', 'S3'])
        self._points = None
        self._sigma = None
        self.cg = None
        # block-boundary topology hooks (PYME's job in the reference; see module docstring)
        self.remesher = None          # None | 'builtin' (host C++) | 'device' (GPU) | callable(mesh, n, target_edge_length, l, n_relax)
        self.neck_remover = None      # callable(mesh, vertex_ids): delete + repair + remesh (PYME's part of remove_necks)
        self.hole_puncher = None      # callable(mesh, points, eps)
        self.edge_cleaner = None      # callable(mesh)  (remove_extra_short_edges)
        self._device = device
        self._native = None
        self._warned_fixed_topology = False
        self.block_log = []
        self.neck_log = []
        self._initialize_curvature_vectors()
        for key, value in kwargs.items():                # :119-120
            setattr(self, key, value)

    # -- topology hooks ---------------------------------------------------------------------------------------
    def _topology_changed(self, vertices, faces, all_referenced=False, mean_edge=None):
        """Rebuild the half-edge tables for a new (vertices, faces) pair; the optimiser of the old topology is dropped."""
        props, vprops = self.vertex_properties, self.vertex_vector_properties
        # (inside a fit the vertex normals of a new topology are the device's to compute: nw_set_mesh with nrm = NULL)
        # ... and the host's half-edge records and 1-rings are built when somebody asks for them (lazy_topology): the device builds its own
        # tables and the remesher works from the face array
        in_fit = getattr(self, '_in_fit', False)
        # ... as are face normals, areas and edge lengths, when the remesher has said what the block loop wants of them: the mean edge length
        TriMesh.__init__(self, vertices, faces, vertex_normals=not in_fit, lazy_topology=in_fit, all_referenced=all_referenced, mean_edge=mean_edge if in_fit else None)
        self.vertex_properties, self.vertex_vector_properties = props, vprops
        self._initialize_curvature_vectors()

    def remesh(self, n=5, target_edge_length=-1, l=0.5, n_relax=10):
        """TriangleMesh.remesh(n, target_edge_length, l, n_relax) as the reference calls it (_membrane_mesh.pyx:1546, :1219).
        `self.remesher`: None = topology held fixed; 'builtin' = this package's isotropic remesher (remesh.py, host C++); 'device' = the same
        algorithm as kernels (include/nanowrap.h: nw_remesh_device);
        or any callable(mesh, n, target_edge_length, l, n_relax), e.g. one that drives PYME.  Returns True if it ran."""
        if self.remesher is None:
            if not self._warned_fixed_topology:
                print("MembraneMesh: no remesher installed (mesh.remesher = 'device', 'builtin' or a callable) -- topology held fixed")
                self._warned_fixed_topology = True
            return False
        if isinstance(self.remesher, str):
            if self.remesher not in ('builtin', 'device'):
                raise ValueError("remesher must be None, 'builtin', 'device' or a callable")
            from .remesh import builtin_remesher, device_remesher
            (device_remesher if self.remesher == 'device' else builtin_remesher)(self, n, target_edge_length, l, n_relax)
        else:
            self.remesher(self, n, target_edge_length, l, n_relax)
        return True

    def neck_vertices(self, neck_curvature_threshold_low=-1e-4, neck_curvature_threshold_high=1e-2):
        """The selection half of `remove_necks` (_membrane_mesh.pyx:1201-1215): curvature refreshed by the GPU kernel,
        then every vertex whose Gaussian curvature lies outside [low, high] is a neck candidate."""
        # (between two blocks only the Gaussian curvature is looked at: the kernel computes all twelve outputs, one is brought back -- 0.8 MB
        # instead of 15 at 2 10^5 vertices; the others come with the next full refresh, which a read of any of them triggers)
        self._curvature_outputs = ('_K',) if (getattr(self, '_in_fit', False) and not self.smooth_curvature) else None
        try:
            self._populate_curvature_grad()
        finally:
            self._curvature_outputs = None
        K = self.curvature_gaussian
        return np.flatnonzero((K < neck_curvature_threshold_low) | (K > neck_curvature_threshold_high))

    def remove_necks(self, neck_curvature_threshold_low=-1e-4, neck_curvature_threshold_high=1e-2):
        """_membrane_mesh.pyx:1201-1219.  Deleting the selected vertices, repairing, remeshing and dropping inner
        surfaces are PYME TriangleMesh operations; they run through the `neck_remover` hook.  Returns the selection."""
        verts = self.neck_vertices(neck_curvature_threshold_low, neck_curvature_threshold_high)
        if len(verts) > 0 and self.neck_remover is not None:
            self.neck_remover(self, verts)
            self.cg = None
            self._host_mesh_changed()
        return verts

    # -- curvature (block-boundary kernel) --------------------------------------------------------------------
    def _neighbor_tables(self):
        """Flat per-slot tables the curvature kernel needs besides the 1-ring vertex ids: the vertex the NEXT half-edge
        points to and the area of the half-edge's face (membrane_mesh_utils.c:1099-1104)."""
        if not getattr(TriMesh, '_numpy_topology', False):
            from .remesh import ring_tables
            _, nxt, area = ring_tables(self._halfedges, self._vertices, self._faces, ring_vertex=False, ring_next=True, ring_area=True)
            return nxt, area
        he = self._halfedges
        nb = self._vertices['neighbors']
        ok = nb != -1
        safe = np.where(ok, nb, 0)
        nxt = he['vertex'][he['next'][safe]]
        nxt[~ok] = -1
        area = self._faces['area'][he['face'][safe]]
        area[~ok] = 0
        return np.ascontiguousarray(nxt, 'i4'), np.ascontiguousarray(area, 'f4')

    def curvature_grad_c(self, dN=0.1, skip_prob=0.0, jitter=None, skip_u=None, outputs=None):
        """_membrane_mesh.pyx:323-347 -> c_curvature_grad (membrane_mesh_utils.c:915-1250) on the GPU.  Fills the
        per-vertex curvature arrays and returns dEdN (M,3).  `jitter`: optional (M,3) float64 array in [0,1) replacing the
        reference's rand() stream; None = deterministic hash.  `skip_prob` > 0 (:962, never used on the live path): a vertex whose
        uniform draw -- `skip_u` (M,), default a seeded generator -- is below it is treated like an unused slot (outputs zeroed)."""
        import ctypes
        from . import _lib as nw
        if self._native is None:
            self._native = NativeContext(self._device)
        nat = self._native
        M = self._position_records().shape[0]
        host_tables = os.environ.get('NW_HOST_TABLES', '0') == '1'
        key = getattr(nat, 'mesh_key', None)
        if skip_prob == 0 and not host_tables and key is not None and key[0] == id(self) and key[1] == M and key[2] == int(self.faces.shape[0]):
            # the block that has just ended left THIS mesh on the device -- positions of its last iteration, normals refreshed there
            # (_block_boundary): nothing is uploaded, and the kernel's two tables are built on the device as well (nw_curvature with NULL
            # tables, round 5) -- the block boundary's neck selection runs without a host table builder
            nxt, area = None, None
        else:
            pos = np.ascontiguousarray(self._vertices['position'], 'f4')
            nrm = np.ascontiguousarray(self.vertex_normals, 'f4')
            nbr = self.neighbor_vertex_table()
            valid = np.ascontiguousarray(self._vertices['halfedge'] != -1, 'u1')
            if skip_prob > 0:                               # `(halfedge == -1) || (r2() < skip_prob)`: float32 draw against the float32 argument
                u = np.random.default_rng(0).random(M) if skip_u is None else np.asarray(skip_u)
                valid = np.ascontiguousarray(valid & ~(u.astype(np.float32) < np.float32(skip_prob)), 'u1')
            faces = np.ascontiguousarray(self.faces, 'i4')
            nat.check(nat.L.nw_set_mesh(nat.h, nw.ptr(pos), nw.ptr(nrm), nw.ptr(nbr), nw.ptr(valid), nw.ptr(faces), M, faces.shape[0], nbr.shape[1]))
            nat.mesh_key = None                             # uploaded outside an optimiser: do not assume it is reusable
            nxt, area = self._neighbor_tables() if host_tables else (None, None)      # (the substrate's ring order is the library's: the tables can be built there)
        jit = None if jitter is None else np.ascontiguousarray(jitter, 'f8')
        self._initialize_curvature_vectors()
        # outputs: names of the per-vertex arrays to bring back (None = all twelve, as the reference fills them); the C-ABI skips a NULL
        # destination, and an array that is not fetched stays cleared (zeros on first read, and `_curv` refreshes everything then)
        out = lambda name: getattr(self, name) if (outputs is None or name in outputs) else None
        dEdN = np.zeros((M, 3), 'f4') if (outputs is None or 'dEdN' in outputs) else None
        nat.check(nat.L.nw_curvature(nat.h, nw.ptr(nxt), nw.ptr(area), nw.ptr(jit), float(self.kc), float(self.kg), float(self.c0), float(dN),
                                     nw.ptr(out('_k_0')), nw.ptr(out('_k_1')), nw.ptr(out('_e_0')), nw.ptr(out('_e_1')), nw.ptr(out('_H')), nw.ptr(out('_K')),
                                     nw.ptr(out('_dH')), nw.ptr(out('_dK')), nw.ptr(out('_E')), nw.ptr(out('_pE')), nw.ptr(out('_dE_neighbors')), nw.ptr(dEdN)))
        return dEdN

    _CURVATURE_SCALARS = ('_H', '_K', '_E', '_k_0', '_k_1', '_pE', '_dH', '_dK', '_dE_neighbors')
    _CURVATURE_VECTORS = ('_e_0', '_e_1')

    def _initialize_curvature_vectors(self):
        """_membrane_mesh.pyx:188-200 zeroes eleven per-vertex arrays; the optimiser calls this after every block (mesh_conj_grad.py:290).
        Here the arrays are dropped and come back as zeros when somebody reads one (`__getattr__`): zeroing 7 MB per block for nobody was
        a third of a millisecond."""
        for name in self._CURVATURE_SCALARS + self._CURVATURE_VECTORS:
            self.__dict__.pop(name, None)

    def __getattr__(self, name):
        # (only reached when normal lookup fails: a curvature array that has not been asked for since it was last cleared)
        if name in MembraneMesh._CURVATURE_SCALARS or name in MembraneMesh._CURVATURE_VECTORS:
            records = self.__dict__.get('_vertex_records')
            if records is None:
                raise AttributeError(name)
            sz = records.shape[0]
            a = np.zeros(sz if name in MembraneMesh._CURVATURE_SCALARS else (sz, 3), np.float32)
            self.__dict__[name] = a
            return a
        raise AttributeError(name)

    def smooth_per_vertex_data(self, data):
        """PYME's TriangleMesh.smooth_per_vertex_data is not in the reference tree; this build's definition: mean over
        the vertex and its 1-ring."""
        nbr = self.neighbor_vertex_table()
        ok = nbr >= 0
        s = np.asarray(data, 'f8') + (np.asarray(data, 'f8')[np.where(ok, nbr, 0)] * ok).sum(1)
        return (s / (1 + ok.sum(1))).astype(np.float32)

    def _populate_curvature_grad(self):
        self.curvature_grad_c(outputs=getattr(self, '_curvature_outputs', None))       # :176-186 (neck_vertices asks for one array only)
        if self.smooth_curvature:
            self._H = self.smooth_per_vertex_data(self._H)
            self._K = self.smooth_per_vertex_data(self._K)
            self._k_0 = self.smooth_per_vertex_data(self._k_0)
            self._k_1 = self.smooth_per_vertex_data(self._k_1)

    def _curv(self, name):
        if not np.any(getattr(self, name)):
            self._populate_curvature_grad()
        return getattr(self, name)

    E = property(lambda self: np.nan_to_num(self._curv('_E'), nan=0.0))                       # :122-127
    pE = property(lambda self: np.nan_to_num(self._curv('_pE'), nan=0.0))                     # :129-134
    curvature_principal0 = property(lambda self: self._curv('_k_0'))                          # :140-144
    curvature_principal1 = property(lambda self: self._curv('_k_1'))
    eigenvector_principal0 = property(lambda self: self._curv('_e_0'))
    eigenvector_principal1 = property(lambda self: self._curv('_e_1'))
    curvature_mean = property(lambda self: self._curv('_H'))                                  # :164-168
    curvature_gaussian = property(lambda self: self._curv('_K'))                              # :170-174

    # -- the driver -------------------------------------------------------------------------------------------
    # Outer loop of the fit (_membrane_mesh.pyx:1427-1560), decomposed into: the block plan (how many iterations run on one
    # topology, which target edge length each remesh gets), the per-block optimiser run, and the block boundary.
    class _BlockPlan(object):
        """Block length = greatest common divisor of the active surgery periods (:1430-1441); the remesh target length moves
        linearly from the current mean edge length to the final one over the planned iterations (:1443-1455, :1544)."""

        def __init__(self, mesh, max_iter, sigma, minimum_edge_length):
            periods = [p for p in (mesh.remesh_frequency, mesh.delaunay_remesh_frequency) if p != 0 and p <= max_iter]
            self.remesh = mesh.remesh_frequency != 0 and mesh.remesh_frequency <= max_iter
            self.punch = mesh.delaunay_remesh_frequency != 0 and mesh.delaunay_remesh_frequency <= max_iter
            self.block = max_iter
            if periods:
                self.block = periods[0] if len(periods) == 1 else math.gcd(*periods)
            self.n_iter = min(max_iter, getattr(mesh, 'truncate_at', max_iter))                      # :1490
            self.length0 = self.slope = None
            if self.remesh:
                self.length0 = mesh._mean_edge_length
                final = np.clip(np.min(sigma) / 2.5, 1.0, 50.0) if minimum_edge_length < 0 else minimum_edge_length
                self.slope = (final - self.length0) / (self.block * np.ceil(max_iter / self.block))

        def target_length(self, iterations_done):
            return self.length0 + self.slope * (iterations_done + 1)

    @staticmethod
    def _inverse_sigma(points, sigma):
        """sigma -> the `sigma_inv` argument of search() (:1460-1473).  NB a scalar sigma is passed through UN-inverted, as upstream."""
        if np.isscalar(sigma):
            return float(sigma)
        n, d = points.shape
        if sigma.ndim == 1 and sigma.shape[0] == n:
            return 1.0 / np.repeat(sigma, d)
        if sigma.ndim == 2 and sigma.shape == (n, d):
            return 1.0 / sigma.ravel()
        raise ValueError('Sigma must be of shape (%d,) or (%d,%d).' % (n, n, d))

    def _host_mesh_changed(self):
        """The HBM copy of the mesh (positions, normals, 1-ring table, valid flags) no longer mirrors this object: the next optimiser
        uploads it again.  Called wherever host code may have touched the mesh -- at the start of every fit (the caller may have
        edited or smoothed positions since the last one) and after every surgery hook."""
        if self._native is not None:
            self._native.mesh_key = None
        # the cached 1-ring vertex table (TriMesh.neighbor_vertex_table) is derived from the half-edge / vertex records a hook may have
        # edited in place: the next optimiser rebuilds it from the records as they are now
        self._ring_vertex_table = None

    def _block_boundary(self, points, done, plan):
        # :1524-1527 -- geometry refreshed from the new positions: vertex normals on the device (they feed the next block's
        # curvature prior; positions and normals stay resident), face areas / edge lengths on the host
        # (with this package's optimiser they stay on the device until somebody reads mesh.vertex_normals; any other optimiser object
        # offering refresh_normals() is called the plain way)
        (getattr(self.cg, 'refresh_normals_lazy', None) or self.cg.refresh_normals)()
        # (face normals / areas / edge lengths of a mesh that this package's own remesher is about to replace, with no hook installed that
        # could look at them first, are computed for nobody: 3.5 ms at 4 10^5 faces)
        own_remesher = isinstance(self.remesher, str) and plan.remesh and done % self.remesh_frequency == 0
        unobserved = own_remesher and self.hole_puncher is None and self.edge_cleaner is None and self.neck_remover is None
        if not unobserved:
            self.update_geometry(vertex_normals=False)
        if plan.punch and done % self.delaunay_remesh_frequency == 0 and self.hole_puncher is not None:   # :1530-1532
            self.hole_puncher(self, points, self.delaunay_eps)
            self._host_mesh_changed()
        if plan.remesh and done % self.remesh_frequency == 0:                                             # :1537-1549
            first = getattr(self, 'neck_first_iter', -1)
            if first > 0 and done > first:                                                                # :1538-1540
                verts = self.remove_necks(getattr(self, 'neck_threshold_low', -1e-4), getattr(self, 'neck_threshold_high', 1e-2))
                self.neck_log.append(dict(iteration=done, candidates=int(len(verts))))
                self._host_mesh_changed()
            if self.edge_cleaner is not None:
                self.edge_cleaner(self)
                self._host_mesh_changed()
            target = plan.target_length(done)
            if self.remesh(5, target, 0.5, n_relax=0):
                self.cg = None
                self._host_mesh_changed()
            self.block_log.append(dict(iteration=done, target_length=float(target), mean_length=float(self._mean_edge_length)))

    def opt_conjugate_gradient(self, points, sigma, max_iter=10, step_size=1.0, weights=None, **kwargs):
        """_membrane_mesh.pyx:1427-1560: blocks of iterations on a fixed topology, mesh surgery between them."""
        plan = self._BlockPlan(self, max_iter, sigma, kwargs.get('minimum_edge_length', -1))
        s = self._inverse_sigma(points, sigma)
        lams = [step_size * self.kc / 2.0] + ([self.shrink_weight] if self.shrink_weight > 0 else [])   # :1483-1486
        if self._native is None:
            self._native = NativeContext(self._device)      # localizations stay in HBM across blocks
        self._host_mesh_changed()
        self.cg = None
        done = 0
        self._in_fit = True
        try:
            return self._run_blocks(points, lams, s, weights, plan)
        finally:
            self._in_fit = False

    def _run_blocks(self, points, lams, s, weights, plan):
        done = 0
        while done < plan.n_iter:
            # a new optimiser per block (:1510-1512).  The localizations stay resident in HBM; while nothing touched the host mesh
            # the device copy is current (positions written by the last block, normals refreshed on the device) and only the
            # optimiser's history restarts
            self.cg = ShrinkwrapMeshConjGrad(self, points, search_k=self.search_k, search_rad=self.search_rad,
                                             shield_sigma=self._mean_edge_length / 2.0, native=self._native, reuse_device_mesh=True, device_tables=True)
            n = min(plan.n_iter - done, plan.block)
            self.cg.search(points, lams=lams, num_iters=n, sigma_inv=s, weights=weights)                  # :1516-1517
            done += n
            self._block_boundary(points, done, plan)
        return done

    def shrink_wrap(self, points=None, sigma=None, method='conjugate_gradient', max_iter=None, **kwargs):
        """_membrane_mesh.pyx:1641-1669."""
        if method not in DESCENT_METHODS:
            print('Unknown gradient descent method. Using {}.'.format(DEFAULT_DESCENT_METHOD))
            method = DEFAULT_DESCENT_METHOD
        if method != 'conjugate_gradient':
            raise NotImplementedError("only method='conjugate_gradient' is on the NanoWrap hot path")
        if max_iter is None:
            max_iter = self.max_iter
        if points is None:
            points = self._points
        if sigma is None:
            sigma = self._sigma
        self._points = points
        self._sigma = sigma
        opts = dict(points=points, sigma=sigma, max_iter=max_iter, step_size=self.step_size, beta_1=self.beta_1, beta_2=self.beta_2,
                    eps=self.eps, **kwargs)
        return self.opt_conjugate_gradient(**opts)

    # -- diagnostics that re-enter the optimiser (:1563-1634) -------------------------------------------------
    @property
    def _S0(self):
        return self.cg.Ahfunc(self.cg.res).reshape(self.vertices.shape)

    @property
    def S0(self):
        return self.cg.S[:, 0].reshape(self.vertices.shape)

    @property
    def S1(self):
        return self.cg.S[:, 1].reshape(self.vertices.shape)

    @property
    def S2(self):
        return self.cg.S[:, 2].reshape(self.vertices.shape)

    @property
    def S3(self):
        return self.cg.S[:, 3].reshape(self.vertices.shape)      # IndexError with 3 columns, exactly as upstream

    @property
    def point_dis(self):
        s0 = self._S0
        return np.sqrt((s0 * s0).sum(1))

    @property
    def rms_point_sc(self):
        rn = (np.sqrt((self.cg.res * self.cg.res).reshape(self.cg.points.shape).sum(1))[:, None] * np.ones(3)[None, :]).ravel()
        rme = self.cg.Ahfunc(rn).reshape(self.vertices.shape)
        return np.sqrt((rme * rme).sum(1))

    @property
    def point_influence(self):
        s = self.cg.Ahfunc(np.ones_like(self.cg.res)).reshape(self.vertices.shape)
        return np.sqrt((s * s).sum(1))


class ShrinkwrapMembrane(object):
    """Parameter surface of the PYME recipe module `ShrinkwrapMembrane`
    (/root/reference/ch_shrinkwrap/recipe_modules/surface_fitting.py:11-115) without the PYME/traits machinery: same trait
    names and defaults, `execute(namespace)` with the same namespace protocol (input mesh under `input`, a table-like
    with 'x','y','z' and the sigma columns under `points`, result stored under `output`).  INTEGRATION.md shows the
    two-line change that makes the real PYME module use this package."""

    def __init__(self, **kw):
        self.input, self.output, self.points = 'surf', 'membrane', 'filtered_localizations'
        self.max_iters = 39
        self.curvature_weight = 20.0
        self.finishing_iters = 0
        self.finishing_curvature_weight = 20.0
        self.shrink_weight = 0.0
        self.kc = 1.0
        self.remesh_frequency = 5
        self.punch_frequency = 0
        self.min_hole_radius = 100.0
        self.sigma_x, self.sigma_y, self.sigma_z = 'error_x', 'error_y', 'error_z'
        self.neck_threshold_low = -1e-3
        self.neck_threshold_high = 1e-2
        self.neck_first_iter = 9
        self.truncate_at = 1000
        self.minimum_edge_length = 5.0
        self.smooth_curvature = True
        self.device = 0
        self.remesher = 'device'                   # not a trait upstream (PYME always remeshes): 'device' = this package's remesher on the GPU (nw_remesh_device), 'builtin' = the same algorithm on the host (nwr_remesh), None holds the topology fixed
        for k, v in kw.items():
            if not hasattr(self, k):
                raise AttributeError('unknown parameter %s' % k)
            setattr(self, k, v)

    def execute(self, namespace):
        import time
        inp = namespace[self.input]
        n_faces = len(inp.faces)
        if not n_faces > 4:                                                                # surface_fitting.py:51-53
            raise RuntimeError('Input mesh only has %d faces, a valid surface needs at least 4 faces' % n_faces)
        mesh = MembraneMesh(mesh=inp, device=self.device, kc=self.kc, max_iter=self.max_iters, step_size=self.curvature_weight,
                            remesh_frequency=self.remesh_frequency, delaunay_remesh_frequency=self.punch_frequency,
                            delaunay_eps=self.min_hole_radius, neck_threshold_low=self.neck_threshold_low,
                            neck_threshold_high=self.neck_threshold_high, neck_first_iter=self.neck_first_iter,
                            shrink_weight=self.shrink_weight, truncate_at=self.truncate_at, remesher=self.remesher)
        namespace[self.output] = mesh
        src = namespace[self.points]
        pts = np.ascontiguousarray(np.vstack([src['x'], src['y'], src['z']]).T)
        try:
            sigma = np.vstack([src[self.sigma_x], src[self.sigma_y], src[self.sigma_z]]).T
        except Exception:
            try:
                sigma = src[self.sigma_x]
            except KeyError:
                print('%s not found in data source, defaulting to 10 nm precision.' % self.sigma_x)
                sigma = 10 * np.ones_like(src['x'])
        start = time.time()
        mesh.shrink_wrap(pts, sigma, method='conjugate_gradient', minimum_edge_length=self.minimum_edge_length)
        if self.finishing_iters > 0:
            mesh.step_size = self.finishing_curvature_weight
            mesh.shrink_wrap(pts, sigma, method='conjugate_gradient', minimum_edge_length=self.minimum_edge_length, max_iter=self.finishing_iters)
        mesh.runtime = time.time() - start          # the reference stores this under md['Processing.ShrinkwrapMembrane.Runtime']
        return mesh
