"""
Minimal half-edge triangle-mesh substrate for the NanoWrap hot path.

The reference inherits all of this from ``PYME.experimental._triangle_mesh.TriangleMesh``
(cimported at /root/reference/ch_shrinkwrap/_membrane_mesh.pyx:11-13,78-80), which is a third-party
dependency that is NOT part of the reference tree.  The optimiser only consumes a handful of
attributes of that object (/root/reference/ch_shrinkwrap/mesh_conj_grad.py:44-54, 777-788, 807,
289-290):

    mesh._vertices   structured array, fields position(3f4) normal(3f4) halfedge(i4) valence(i4)
                     neighbors(20 i4, half-edge ids, -1 padded) component(i4) locally_manifold(i4)
                     -- layout of vertex_t, /root/reference/ch_shrinkwrap/membrane_mesh_utils.h:57-65
    mesh._halfedges  structured array, fields vertex face twin next prev length component
                     -- halfedge_t, membrane_mesh_utils.h:31-39
    mesh._faces      structured array, fields halfedge normal(3f4) area component
                     -- face_t, membrane_mesh_utils.h:41-46
    mesh.faces (F,3) i4, mesh.vertices, mesh.vertex_normals, mesh.vertex_neighbors,
    mesh.face_normals, mesh._initialize_curvature_vectors()

This module provides exactly that surface, with this build's OWN definitions of the things PYME
leaves unpinned:

  * half-edge ``3f+k`` runs faces[f,k] -> faces[f,(k+1)%3]; its ``vertex`` field is the vertex it
    points TO (so ``_halfedges['vertex'][_vertices['neighbors']]`` are the 1-ring vertex ids, as the
    reference expects at mesh_conj_grad.py:50);
  * the 1-ring is ordered by rotating ``h -> twin[prev[h]]`` starting from the lowest-numbered
    outgoing half-edge (for boundary vertices: from the outgoing half-edge whose ``prev`` has no twin);
  * vertex normals are the normalised sum of the (area-weighted) cross products of incident faces.

The same arrays are fed to the reference (golden generation), the oracle and the HIP path, so hot-loop
parity does not depend on PYME's conventions.
"""
import numpy as np

NEIGHBORSIZE = 20   # membrane_mesh_utils.h:29
VECTORSIZE = 3      # membrane_mesh_utils.h:26

VERTEX_DTYPE = np.dtype([('position', '3f4'), ('normal', '3f4'), ('halfedge', 'i4'), ('valence', 'i4'),
                         ('neighbors', '%di4' % NEIGHBORSIZE), ('component', 'i4'),
                         ('locally_manifold', 'i4')])
HALFEDGE_DTYPE = np.dtype([('vertex', 'i4'), ('face', 'i4'), ('twin', 'i4'), ('next', 'i4'), ('prev', 'i4'),
                           ('length', 'f4'), ('component', 'i4')])
FACE_DTYPE = np.dtype([('halfedge', 'i4'), ('normal', '3f4'), ('area', 'f4'), ('component', 'i4')])

assert VERTEX_DTYPE.itemsize == 120 and HALFEDGE_DTYPE.itemsize == 28 and FACE_DTYPE.itemsize == 24


def icosahedron():
    t = (1.0 + np.sqrt(5.0)) / 2.0
    v = np.array([[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0],
                  [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
                  [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype='f8')
    v /= np.linalg.norm(v, axis=1)[:, None]
    f = np.array([[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11],
                  [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6], [7, 1, 8],
                  [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9],
                  [4, 9, 5], [2, 4, 11], [6, 2, 10], [8, 6, 7], [9, 8, 1]], dtype='i4')
    return v, f


def subdivide(v, f):
    """One 1->4 subdivision; new vertices at edge midpoints (not re-projected)."""
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    es = np.sort(e, 1)
    key = es[:, 0].astype('i8') * (v.shape[0] + 1) + es[:, 1]
    uk, first, inv = np.unique(key, return_index=True, return_inverse=True)
    mid = 0.5 * (v[es[first, 0]] + v[es[first, 1]])
    nv = np.concatenate([v, mid], 0)
    F = f.shape[0]
    m01 = v.shape[0] + inv[0:F]
    m12 = v.shape[0] + inv[F:2 * F]
    m20 = v.shape[0] + inv[2 * F:3 * F]
    nf = np.concatenate([np.stack([f[:, 0], m01, m20], 1),
                         np.stack([f[:, 1], m12, m01], 1),
                         np.stack([f[:, 2], m20, m12], 1),
                         np.stack([m01, m12, m20], 1)], 0).astype('i4')
    return nv, nf


def icosphere(nsub=4, radius=1.0, dtype='f4'):
    """Unit icosahedron subdivided ``nsub`` times, vertices pushed to ``radius``.
    nsub=4 -> 2562 vertices / 5120 faces (config C1)."""
    v, f = icosahedron()
    for _ in range(nsub):
        v, f = subdivide(v, f)
        v /= np.linalg.norm(v, axis=1)[:, None]
    return (v * radius).astype(dtype), f


def geodesic_sphere(n, radius=1.0, dtype='f4'):
    """Class-I geodesic sphere of frequency n: every icosahedron face is split into n^2 triangles, giving
    10 n^2 + 2 vertices and 20 n^2 faces (n=141 -> 198 812 vertices, the '200k' of BASELINE.json's headline config;
    n=63 -> 39 692; n=283 -> 800 892).  Vertices are pushed to `radius`."""
    v0, f0 = icosahedron()
    n = int(n)
    # integer barycentric lattice of one face
    ij = [(i, j) for i in range(n + 1) for j in range(n + 1 - i)]
    ij = np.array(ij, 'i8')
    i, j = ij[:, 0], ij[:, 1]
    k = n - i - j
    lut = -np.ones((n + 1, n + 1), 'i8')
    lut[i, j] = np.arange(ij.shape[0])
    # small triangles of the lattice: "up" (i,j),(i+1,j),(i,j+1) and "down" (i+1,j),(i+1,j+1),(i,j+1)
    up = ij[(i + j) < n]
    dn = ij[(i + j) < n - 1]
    tri = np.concatenate([np.stack([lut[up[:, 0], up[:, 1]], lut[up[:, 0] + 1, up[:, 1]], lut[up[:, 0], up[:, 1] + 1]], 1),
                          np.stack([lut[dn[:, 0] + 1, dn[:, 1]], lut[dn[:, 0] + 1, dn[:, 1] + 1], lut[dn[:, 0], dn[:, 1] + 1]], 1)], 0)
    P = []
    T = []
    nl = ij.shape[0]
    for fi, (a, b, c) in enumerate(f0):
        # weights (k, i, j) on corners (a, b, c): orientation-preserving
        p = (k[:, None] * v0[a][None, :] + i[:, None] * v0[b][None, :] + j[:, None] * v0[c][None, :]) / float(n)
        P.append(p)
        T.append(tri + fi * nl)
    P = np.concatenate(P, 0)
    T = np.concatenate(T, 0)
    # merge the lattice points shared along icosahedron edges / corners
    key = np.round(P * (4.0 * n)).astype('i8')
    _, first, inv = np.unique(key, axis=0, return_index=True, return_inverse=True)
    inv = inv.reshape(-1)
    V = P[first]
    V /= np.linalg.norm(V, axis=1)[:, None]
    F = inv[T].astype('i4')
    return (V * radius).astype(dtype), F


def _build_halfedges(faces, n_vertices):
    F = faces.shape[0]
    he = np.zeros(3 * F, HALFEDGE_DTYPE)
    idx = np.arange(3 * F, dtype='i4')
    k = idx % 3
    fi = idx // 3
    origin = faces[fi, k]
    dest = faces[fi, (k + 1) % 3]
    he['vertex'] = dest
    he['face'] = fi
    he['next'] = 3 * fi + (k + 1) % 3
    he['prev'] = 3 * fi + (k + 2) % 3
    he['component'] = 0
    # twins: match (origin,dest) with (dest,origin) -- linear-time native pairing (include/nw_remesh.h) when the mesh is an
    # oriented 2-manifold, the sort-based NumPy pairing below otherwise (it tolerates non-manifold edges)
    try:
        from .remesh import halfedge_twins
        he['twin'] = halfedge_twins(faces, n_vertices)
        return he, origin.astype('i4')
    except (RuntimeError, ValueError):                # an input the library rejects (non-manifold edge): NumPy definition below.
        pass                                          # (a missing library raises ImportError and is NOT hidden)
    nv = np.int64(n_vertices)
    key = origin.astype('i8') * nv + dest
    rkey = dest.astype('i8') * nv + origin
    order = np.argsort(key, kind='stable')
    pos = np.searchsorted(key[order], rkey)
    pos = np.minimum(pos, 3 * F - 1)
    cand = order[pos]
    twin = np.where(key[cand] == rkey, cand, -1).astype('i4')
    he['twin'] = twin
    return he, origin.astype('i4')


class TriMesh(object):
    """Duck-typed stand-in for the PYME mesh object the optimiser reads (see module docstring)."""

    # The vertex records may be behind the optimiser by one block: with NW_FLAG_ROWS_ASYNC (include/nanowrap.h) `search()` returns when
    # its (M,3) result is complete and the library's host threads fill the 'position' rows of the 120-byte records while the caller goes
    # on -- typically while the next block runs on the GPU.  Every access to `_vertices` first waits for them (`_rows_pending`, set by
    # the optimiser), so nobody sees a half-written array; a mesh class without this property (PYME's) gets the synchronous write-back.
    _accepts_deferred_rows = True

    # The half-edge records and the vertex records' topology fields ('halfedge', 'valence', 'neighbors') may not have been built yet
    # (`lazy_topology`: the driver's block loop, where the device builds its own tables from the faces and the remesher works from the face
    # array -- nobody on the host reads them between two blocks, and building them was a fifth of a block boundary).  `_vertices`,
    # `_halfedges` and `_origin` build them on first access; code that only wants positions / normals asks `_position_records()`.
    @property
    def _vertices(self):
        pending = self.__dict__.get('_rows_pending')
        if pending is not None:
            pending()
        if self.__dict__.get('_topology_pending'):
            self._build_topology()
        return self.__dict__['_vertex_records']

    def _position_records(self):
        """The vertex records for somebody who reads or writes 'position' / 'normal' only: waits for rows a block is still writing, not for
        the topology fields."""
        pending = self.__dict__.get('_rows_pending')
        if pending is not None:
            pending()
        return self.__dict__['_vertex_records']

    @property
    def _halfedges(self):
        if self.__dict__.get('_topology_pending'):
            self._build_topology()
        return self.__dict__['_halfedge_records']

    # Face normals, face areas and edge lengths may be pending too (`mean_edge` given to the constructor: the driver's block loop, which needs
    # nothing of the geometry between two blocks but the mean edge length, and gets that from the remesher's statistics): `_faces`,
    # `face_normals`, `area()`, the half-edge lengths compute them on first access.
    @property
    def _faces(self):
        if self.__dict__.get('_geometry_pending'):
            self._flush_geometry()
        return self.__dict__['_face_records']

    @_faces.setter
    def _faces(self, records):
        self.__dict__['_face_records'] = records

    def _flush_geometry(self):
        self.__dict__['_geometry_pending'] = False
        if self.__dict__.get('_face_records') is None:
            fr = np.zeros(self._faces_arr.shape[0], FACE_DTYPE)
            fr['halfedge'] = 3 * np.arange(self._faces_arr.shape[0], dtype='i4')
            self.__dict__['_face_records'] = fr
        if self.__dict__.get('_halfedge_records') is None and self.__dict__.get('_lengths_packed') is None:
            self.__dict__['_lengths_packed'] = np.zeros(3 * self._faces_arr.shape[0], 'f4')
        mean = self.__dict__.get('_mean_edge_cache')
        self.update_geometry(vertex_normals=False)
        self.__dict__['_mean_edge_cache'] = mean          # (the value the caller has been given stays the value)

    def _edge_lengths(self):
        """Where update_geometry writes the half-edge lengths (half-edge 3f + k is edge k of face f: the faces suffice): the records' 'length'
        field, or -- while the records of a lazy topology have not been asked for -- a packed array that is copied into them when they are."""
        if self.__dict__.get('_geometry_pending'):
            self._flush_geometry()
        rec = self.__dict__.get('_halfedge_records')
        return rec['length'] if rec is not None else self.__dict__['_lengths_packed']

    @_halfedges.setter
    def _halfedges(self, records):
        self.__dict__['_halfedge_records'] = records

    @property
    def _origin(self):
        if self.__dict__.get('_topology_pending'):
            self._build_topology()
        return self.__dict__['_halfedge_origin']

    @_origin.setter
    def _origin(self, origin):
        self.__dict__['_halfedge_origin'] = origin

    def valid_vertex_mask(self):
        """`_vertices['halfedge'] != -1` (the reference's test for a vertex slot in use, mesh_conj_grad.py:44) without building the topology:
        a slot is in use exactly when a face refers to it."""
        if not self.__dict__.get('_topology_pending'):
            return self._vertices['halfedge'] != -1
        if self.__dict__.get('_all_referenced'):          # (a remesher's output: it drops what no face refers to)
            return np.ones(self.__dict__['_vertex_records'].shape[0], bool)
        mask = np.zeros(self.__dict__['_vertex_records'].shape[0], bool)
        mask[self._faces_arr.ravel()] = True
        return mask

    @_vertices.setter
    def _vertices(self, records):
        pending = self.__dict__.get('_rows_pending')
        if pending is not None:
            pending()
        self.__dict__['_vertex_records'] = records

    def __init__(self, vertices, faces, max_vertices=None, vertex_normals=True, lazy_topology=False, all_referenced=False, mean_edge=None):
        vertices = np.ascontiguousarray(vertices, dtype='f4')
        faces = np.ascontiguousarray(faces, dtype='i4')
        M = vertices.shape[0] if max_vertices is None else int(max_vertices)
        self.__dict__['_topology_pending'] = False
        self._vertices = np.zeros(M, VERTEX_DTYPE)
        rec = self.__dict__['_vertex_records']
        rec['position'][:vertices.shape[0]] = vertices
        self._nv = vertices.shape[0]
        self._faces_arr = faces
        self._ring_vertex_table = None
        self.__dict__['_geometry_pending'] = False
        self.__dict__['_face_records'] = None
        self._origin = None
        self.__dict__['_all_referenced'] = bool(all_referenced) and M == vertices.shape[0]
        if lazy_topology and faces.shape[0] > 0 and int(faces.min()) >= 0 and int(faces.max()) < M:
            # (no half-edge records yet -- 28 bytes a half-edge, zeroed and paged in for nothing if nobody asks: until then the lengths
            # update_geometry computes live in a packed array)
            self.__dict__['_topology_pending'] = True
            self.__dict__['_halfedge_records'] = None
            self.__dict__['_lengths_packed'] = None
        else:
            self._halfedges = np.zeros(3 * faces.shape[0], HALFEDGE_DTYPE)
            self._build_topology()
        # vertex_normals=False (the driver's block loop): the device computes them with the next upload and hands them back after the block;
        # whoever asks before that gets them computed here, on first use (`vertex_normals`)
        self._normals_stale = not vertex_normals
        if self.__dict__['_topology_pending'] and not vertex_normals and mean_edge is not None and float(mean_edge) > 0:
            # (geometry on first use; the one number the block loop asks for is the remesher's)
            self.__dict__['_geometry_pending'] = True
            self.__dict__['_mean_edge_cache'] = np.float32(mean_edge)
        else:
            self._flush_geometry()
            if vertex_normals:
                self.update_geometry(vertex_normals=True)
        self.cg = None
        self.vertex_properties = []
        self.vertex_vector_properties = []

    # -- topology ---------------------------------------------------------------------------
    def _build_topology(self):
        """Half-edge records and 1-rings of the face array (at construction, or -- `lazy_topology` -- when somebody first asks)."""
        self.__dict__['_topology_pending'] = False
        rec = self.__dict__['_vertex_records']
        rec['halfedge'] = -1
        rec['neighbors'] = -1
        if self.__dict__.get('_halfedge_records') is None:
            if self.__dict__.get('_geometry_pending'):
                self._flush_geometry()                    # (the records are born with their 'length' field filled)
            he = np.zeros(3 * self._faces_arr.shape[0], HALFEDGE_DTYPE)
            packed = self.__dict__.pop('_lengths_packed', None)
            if packed is not None:
                he['length'] = packed
            self.__dict__['_halfedge_records'] = he
        if not self._build_topology_native(self._faces_arr):
            lengths = self.__dict__['_halfedge_records']['length'].copy()
            self._halfedges, self._origin = _build_halfedges(self._faces_arr, rec.shape[0])
            self.__dict__['_halfedge_records']['length'] = lengths
            self._build_rings()

    def _build_topology_native(self, faces):
        """Half-edge records and 1-rings in one native pass (include/nw_remesh.h: nwr_build_topology, the same conventions as the
        NumPy definitions below, which stay the fallback for inputs the library rejects -- non-manifold edges)."""
        if getattr(TriMesh, '_numpy_topology', False) or faces.shape[0] < 1:
            return False
        from .remesh import build_topology
        he = self.__dict__['_halfedge_records']           # (the native pass fills vertex / face / twin / next / prev; 'length' is update_geometry's)
        rec = self.__dict__['_vertex_records']
        try:
            origin = build_topology(faces, he, rec)
        except (RuntimeError, ValueError):                # (a missing library raises ImportError and is NOT hidden)
            rec['halfedge'] = -1
            rec['neighbors'] = -1
            rec['valence'] = 0
            return False
        self._origin = origin
        rec['locally_manifold'] = 1
        rec['component'] = 0
        return True

    def _build_rings(self):
        he = self._halfedges
        M = self.__dict__['_vertex_records'].shape[0]
        nhe = he.shape[0]
        origin = self._origin
        twin, prev = he['twin'], he['prev']
        # choose the start half-edge per vertex: lowest-numbered outgoing; boundary vertices start at the outgoing
        # half-edge that has no twin (nothing precedes it), so that the counter-clockwise walk covers the whole fan
        start = np.full(M, -1, 'i4')
        order = np.arange(nhe - 1, -1, -1, dtype='i4')
        start[origin[order]] = order                      # lowest index wins (written last)
        bnd = np.nonzero(twin == -1)[0]
        if bnd.size:
            start[origin[bnd]] = bnd.astype('i4')
        self._vertices['halfedge'] = start
        nb = np.full((M, NEIGHBORSIZE), -1, 'i4')
        cur = start.copy()
        alive = cur != -1
        valence = np.zeros(M, 'i4')
        for s in range(NEIGHBORSIZE):
            if not alive.any():
                break
            nb[alive, s] = cur[alive]
            valence[alive] += 1
            p = prev[np.where(alive, cur, 0)]
            nxt = np.where(alive, twin[p], -1)
            alive = alive & (nxt != -1) & (nxt != start)
            cur = np.where(alive, nxt, -1).astype('i4')
        # boundary fans end with the incoming boundary edge's origin: add it as a last neighbour is
        # NOT done -- open fans simply list their outgoing half-edges (documented convention).
        self._vertices['neighbors'] = nb
        self._vertices['valence'] = valence
        self._vertices['locally_manifold'] = 1
        self._vertices['component'] = 0

    # -- geometry ---------------------------------------------------------------------------
    def update_geometry(self, vertex_normals=True):
        """Face normals/areas, half-edge lengths and (unless they were refreshed on the device) vertex normals from the
        current positions.  This is the block-boundary refresh the reference triggers at _membrane_mesh.pyx:1524-1527."""
        rec = self._position_records()
        pos = rec['position']
        f = self._faces_arr
        if self.__dict__.get('_geometry_pending') or self.__dict__.get('_face_records') is None:
            # (an explicit refresh of a mesh whose geometry was still pending: the records first -- _flush_geometry calls back here)
            self.__dict__['_geometry_pending'] = False
            self._flush_geometry()
            if not vertex_normals:
                self.__dict__['_mean_edge_cache'] = None
                return
        self.__dict__['_mean_edge_cache'] = None
        if vertex_normals:
            self.__dict__['_normals_stale'] = False       # (computed here, below)
        if not getattr(self, '_numpy_geometry', False):
            try:                                              # native, bit-identical to the NumPy definition below
                from .remesh import mesh_geometry
                # (written straight into the records' fields: no staging arrays)
                mesh_geometry(pos, f, vertex_normals, out=(self._faces['normal'], self._faces['area'], self._edge_lengths(),
                                                           rec['normal'] if vertex_normals else None))
                return
            except (RuntimeError, ValueError):                # an input the library rejects: NumPy definition below
                pass
        v0, v1, v2 = pos[f[:, 0]], pos[f[:, 1]], pos[f[:, 2]]
        cr = np.cross(v1 - v0, v2 - v0)                      # f32, |cr| = 2*area
        nrm = np.sqrt((cr * cr).sum(1))
        with np.errstate(invalid='ignore', divide='ignore'):
            fn = cr / nrm[:, None]
        fn[~np.isfinite(fn)] = 0
        self._faces['normal'] = fn
        self._faces['area'] = 0.5 * nrm
        he = self._halfedges
        d = pos[he['vertex']] - pos[self._origin]
        he['length'] = np.sqrt((d * d).sum(1))
        if not vertex_normals:
            return
        vn = np.zeros((pos.shape[0], 3), 'f8')
        fi = f.T.ravel()                                     # corner-major: all corner-0 ids, then corner-1, corner-2
        for c in range(3):
            vn[:, c] = np.bincount(fi, weights=np.tile(cr[:, c].astype('f8'), 3), minlength=pos.shape[0])
        l = np.sqrt((vn * vn).sum(1))
        with np.errstate(invalid='ignore', divide='ignore'):
            vn = vn / l[:, None]
        vn[~np.isfinite(vn)] = 0
        rec['normal'] = vn.astype('f4')

    # -- PYME-like attribute surface --------------------------------------------------------
    @property
    def faces(self):
        return self._faces_arr

    @property
    def vertices(self):
        return self._position_records()['position']

    @property
    def vertex_normals(self):
        stale = self.__dict__.get('_normals_stale', False)
        if stale:
            self._normals_stale = False
            if callable(stale):
                stale()                                   # the device holds them (the optimiser refreshed them there): fetched on first use
            else:
                self.update_geometry(vertex_normals=True)
        return self._position_records()['normal']

    @property
    def face_normals(self):
        return self._faces['normal']

    @property
    def vertex_neighbors(self):
        return self._vertices['neighbors']

    @property
    def _mean_edge_length(self):
        # (cached per geometry refresh: the driver asks twice per block, 2.3 ms a time at 1.2 million half-edges)
        cached = self.__dict__.get('_mean_edge_cache')
        if cached is None:
            l = self._edge_lengths()                         # (of every face's three edges: no topology needed)
            live = l != -1
            cached = np.mean(l if live.all() else l[live])   # (the same number either way: both are one contiguous float32 array to numpy's pairwise sum)
            self.__dict__['_mean_edge_cache'] = cached
        return cached

    def area(self):
        return float(self._faces['area'].sum())

    def neighbor_vertex_table(self):
        """(M, NEIGHBORSIZE) i4 table of 1-ring VERTEX ids, -1 padded -- what the reference caches
        at mesh_conj_grad.py:50-54."""
        if not getattr(TriMesh, '_numpy_topology', False):
            # built once per topology (TriMesh.__init__ drops it); callers treat the table as read-only
            if getattr(self, '_ring_vertex_table', None) is None:
                from .remesh import ring_tables
                self._ring_vertex_table = ring_tables(self._halfedges, self._vertices)[0]
            return self._ring_vertex_table
        n = self._halfedges['vertex'][self._vertices['neighbors']]
        n[self._vertices['neighbors'] == -1] = -1
        return np.ascontiguousarray(n, dtype='i4')

    @property
    def point_influence(self):
        # _membrane_mesh.pyx:1625-1634
        s = self.cg.Ahfunc(np.ones_like(self.cg.res)).reshape(self.vertices.shape)
        return np.sqrt((s * s).sum(1))

    def _initialize_curvature_vectors(self):
        # _membrane_mesh.pyx:188-214 zeroes the cached curvature arrays; nothing cached here.
        pass
