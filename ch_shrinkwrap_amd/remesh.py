"""
ctypes binding of include/nw_remesh.h (libnw_remesh.so): the block-boundary isotropic remesher.

`remesh(vertices, faces, n, target_edge_length, l, n_relax)` mirrors the signature of the PYME method the reference
calls between optimiser blocks, `self.remesh(5, target_length, 0.5, n_relax=0)`
(/root/reference/ch_shrinkwrap/_membrane_mesh.pyx:1546); `builtin_remesher` is the hook form used by
`MembraneMesh.remesher` (ch_shrinkwrap_amd/membrane_mesh.py).  PYME's own implementation is not in the reference tree;
this is the published algorithm it follows (Botsch & Kobbelt 2004) -- parity with PYME is unpinned, see DESIGN.md.
"""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, 'libnw_remesh.so')
SYMBOLS = ['nwr_abi_version', 'nwr_configure', 'nwr_remesh', 'nwr_free', 'nwr_halfedge_twins', 'nwr_mesh_geometry', 'nwr_build_topology', 'nwr_ring_tables']
ERRORS = {-1: 'bad argument (sizes, indices or a non-finite vertex)', -2: 'the mesh is not an oriented 2-manifold', -3: 'out of memory',
          -4: 'runaway: far more splits than the target length can explain (degenerate input)'}

_lib = None


class Stats(ctypes.Structure):
    _fields_ = [('n_split', ctypes.c_int64), ('n_collapse', ctypes.c_int64), ('n_flip', ctypes.c_int64),
                ('mean_edge_length', ctypes.c_double), ('max_valence', ctypes.c_int32), ('reserved', ctypes.c_int32)]


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            # never compile lazily: the first caller may run under rocprofv3 (a compiler child started from a process whose GPU is
            # initialised is the exec hop the GPU pool forbids), and a silent g++ run hides a missing build step
            raise ImportError('%s not found: build it with `python -m ch_shrinkwrap_amd.build` (or __graft_entry__.build())' % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        L.nwr_abi_version.restype = ctypes.c_int
        L.nwr_remesh.restype = ctypes.c_int
        L.nwr_remesh.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                 ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int64),
                                 ctypes.POINTER(Stats)]
        L.nwr_halfedge_twins.restype = ctypes.c_int
        L.nwr_halfedge_twins.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        L.nwr_mesh_geometry.restype = ctypes.c_int
        L.nwr_mesh_geometry.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64] + [ctypes.c_void_p, ctypes.c_int64] * 4
        L.nwr_free.restype = None
        L.nwr_free.argtypes = [ctypes.c_void_p]
        L.nwr_build_topology.restype = ctypes.c_int
        L.nwr_build_topology.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p] + [ctypes.c_int64] * 6 + \
                                         [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int64] * 4 + [ctypes.c_int32]
        L.nwr_ring_tables.restype = ctypes.c_int
        L.nwr_ring_tables.argtypes = [ctypes.c_void_p] + [ctypes.c_int64] * 5 + [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int64,
                                      ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.nwr_configure.restype = ctypes.c_int
        L.nwr_configure.argtypes = [ctypes.c_int, ctypes.c_int]
        if L.nwr_abi_version() != 4:
            raise RuntimeError('libnw_remesh.so ABI version mismatch')
        _lib = L
    return _lib


def remesh(vertices, faces, n=5, target_edge_length=-1, l=0.5, n_relax=10, max_valence=16, return_stats=False, serial=False):
    """Returns (vertices float32 (V,3), faces int32 (F,3)) of the remeshed surface.

    Meshes of 40 000 faces and more are remeshed in pieces on all cores when n_relax == 0 (a valid result of the same algorithm,
    independent of the number of threads, but not the serial algorithm's arrays); serial=True keeps to the serial algorithm (whose
    output only changes when the algorithm does: the benchmark's mesh generator uses it)."""
    L = load()
    if serial:
        old = L.nwr_configure(0, 0)
        try:
            return remesh(vertices, faces, n, target_edge_length, l, n_relax, max_valence, return_stats, serial=False)
        finally:
            L.nwr_configure(0, old)
    v = np.ascontiguousarray(vertices, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3:
        raise ValueError('vertices must be (V,3) and faces (F,3)')
    ov, of = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    st = Stats()
    rc = L.nwr_remesh(v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], int(n), float(target_edge_length), float(l), int(n_relax),
                      int(max_valence), ctypes.byref(ov), ctypes.byref(nv), ctypes.byref(of), ctypes.byref(nf), ctypes.byref(st))
    if rc != 0:
        raise RuntimeError('nwr_remesh: %s' % ERRORS.get(rc, 'error %d' % rc))
    try:
        out_v = np.ctypeslib.as_array(ctypes.cast(ov, ctypes.POINTER(ctypes.c_float)), shape=(nv.value, 3)).copy()
        out_f = np.ctypeslib.as_array(ctypes.cast(of, ctypes.POINTER(ctypes.c_int32)), shape=(nf.value, 3)).copy()
    finally:
        L.nwr_free(ov)
        L.nwr_free(of)
    if return_stats:
        return out_v, out_f, dict(n_split=st.n_split, n_collapse=st.n_collapse, n_flip=st.n_flip,
                                  mean_edge_length=st.mean_edge_length, max_valence=st.max_valence)
    return out_v, out_f


def halfedge_twins(faces, n_vertices):
    """twin[3f+k] for an oriented (F,3) face array (-1 on boundaries); raises on non-manifold edges."""
    L = load()
    f = np.ascontiguousarray(faces, np.int32)
    twin = np.empty(3 * f.shape[0], np.int32)
    rc = L.nwr_halfedge_twins(f.ctypes.data, f.shape[0], int(n_vertices), twin.ctypes.data)
    if rc != 0:
        raise RuntimeError('nwr_halfedge_twins: %s' % ERRORS.get(rc, 'error %d' % rc))
    return twin


def build_topology(faces, halfedges, vertices):
    """Fill the half-edge records (vertex, face, twin, next, prev) and the vertex records (halfedge, valence, neighbors) of a
    TriMesh from its face array; returns the origin vertex of every half-edge.  Raises RuntimeError on a non-manifold edge."""
    L = load()
    f = np.ascontiguousarray(faces, np.int32)
    origin = np.empty(3 * f.shape[0], np.int32)
    ho = lambda name: halfedges.dtype.fields[name][1]
    vo = lambda name: vertices.dtype.fields[name][1]
    nbsize = vertices.dtype.fields['neighbors'][0].shape[0]
    rc = L.nwr_build_topology(f.ctypes.data, f.shape[0], vertices.shape[0], halfedges.ctypes.data, halfedges.strides[0], ho('vertex'), ho('face'),
                              ho('twin'), ho('next'), ho('prev'), origin.ctypes.data, vertices.ctypes.data, vertices.strides[0], vo('halfedge'),
                              vo('valence'), vo('neighbors'), nbsize)
    if rc != 0:
        raise RuntimeError('nwr_build_topology: %s' % ERRORS.get(rc, 'error %d' % rc))
    return origin


def ring_tables(halfedges, vertices, faces_rec=None, ring_vertex=True, ring_next=False, ring_area=False):
    """(ring vertex ids, vertex after each ring half-edge, area of each ring half-edge's face) as (M, NEIGHBORSIZE) tables; entries
    not asked for are None."""
    L = load()
    ho = lambda name: halfedges.dtype.fields[name][1]
    nbsize = vertices.dtype.fields['neighbors'][0].shape[0]
    M = vertices.shape[0]
    rv = np.empty((M, nbsize), np.int32) if ring_vertex else None
    rn = np.empty((M, nbsize), np.int32) if ring_next else None
    ra = np.empty((M, nbsize), np.float32) if ring_area else None
    fa_ptr, fa_stride = (None, 0)
    if ring_area:
        fa_ptr, fa_stride = faces_rec.ctypes.data + faces_rec.dtype.fields['area'][1], faces_rec.strides[0]
    ptr = lambda a: a.ctypes.data if a is not None else None
    rc = L.nwr_ring_tables(halfedges.ctypes.data, halfedges.strides[0], ho('vertex'), ho('face'), ho('next'), halfedges.shape[0],
                           vertices.ctypes.data, vertices.strides[0], vertices.dtype.fields['neighbors'][1], nbsize, M,
                           fa_ptr, fa_stride, ptr(rv), ptr(rn), ptr(ra))
    if rc != 0:
        raise RuntimeError('nwr_ring_tables: %s' % ERRORS.get(rc, 'error %d' % rc))
    return rv, rn, ra


def mesh_geometry(positions, faces, vertex_normals=True, out=None):
    """(face normals (F,3), face areas (F,), half-edge lengths (3F,), vertex normals (V,3) or None) of a mesh whose
    positions are a float32 (V,3) array, possibly a strided view into vertex records.  out = (fn, fa, hl, vn): float32 arrays -- packed or
    fields of record arrays -- the library writes into directly (vn may be None)."""
    L = load()
    pos = positions
    if pos.dtype != np.float32 or pos.ndim != 2 or pos.shape[1] != 3 or pos.strides[1] != 4 or pos.strides[0] < 12:
        pos = np.ascontiguousarray(positions, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    V, F = pos.shape[0], f.shape[0]
    if out is None:
        out = (np.empty((F, 3), np.float32), np.empty(F, np.float32), np.empty(3 * F, np.float32), np.empty((V, 3), np.float32) if vertex_normals else None)
    fn, fa, hl, vn = out
    for a, shape in ((fn, (F, 3)), (fa, (F,)), (hl, (3 * F,)), (vn, (V, 3))):
        if a is not None and (a.dtype != np.float32 or a.shape != shape or (a.ndim == 2 and a.strides[1] != 4)):
            raise ValueError('mesh_geometry: an output array of the wrong type or shape')
    rc = L.nwr_mesh_geometry(pos.ctypes.data, pos.strides[0], V, f.ctypes.data, F, fn.ctypes.data, fn.strides[0], fa.ctypes.data, fa.strides[0],
                             hl.ctypes.data, hl.strides[0], vn.ctypes.data if vn is not None else None, vn.strides[0] if vn is not None else 0)
    if rc != 0:
        raise RuntimeError('nwr_mesh_geometry: %s' % ERRORS.get(rc, 'error %d' % rc))
    return fn, fa, hl, vn


class DeviceStats(ctypes.Structure):
    _fields_ = [('n_split', ctypes.c_int64), ('n_collapse', ctypes.c_int64), ('n_flip', ctypes.c_int64), ('mean_edge_length', ctypes.c_double),
                ('max_valence', ctypes.c_int32), ('rounds_split', ctypes.c_int32), ('rounds_collapse', ctypes.c_int32), ('rounds_flip', ctypes.c_int32)]


def remesh_device(vertices, faces, n=5, target_edge_length=-1, l=0.5, n_relax=0, max_valence=16, return_stats=False, device=0):
    """The same remeshing step on the GPU (include/nanowrap.h: nw_remesh_device; csrc/nw_remesh_dev.hip): split / collapse / flip as rounds of
    independent operations, then n_relax steps of tangential relaxation (default 0: what the block boundary asks for).  A valid result of the algorithm and the same arrays on every run, but not the host remesher's
    arrays.  Needs libnanowrap_hip.so and a GPU: there is no fallback."""
    from . import _lib as nw
    L = nw.load()
    v = np.ascontiguousarray(vertices, np.float32)
    f = np.ascontiguousarray(faces, np.int32)
    if v.ndim != 2 or v.shape[1] != 3 or f.ndim != 2 or f.shape[1] != 3:
        raise ValueError('vertices must be (V,3) and faces (F,3)')
    target = float(target_edge_length)
    if not target > 0:                                   # PYME's default: the mean edge length of the input
        e = v[f] - v[np.roll(f, -1, 1)]
        target = float(np.sqrt((e.astype('f8') ** 2).sum(2)).mean())
    ov, of = ctypes.c_void_p(), ctypes.c_void_p()
    nv, nf = ctypes.c_int64(), ctypes.c_int64()
    st = DeviceStats()
    L.nw_remesh_device.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    L.nw_host_free.argtypes = [ctypes.c_void_p]
    L.nw_host_free.restype = None
    rc = L.nw_remesh_device(int(device), v.ctypes.data, v.shape[0], f.ctypes.data, f.shape[0], int(n), target, float(l), int(n_relax), int(max_valence),
                            ctypes.byref(ov), ctypes.byref(nv), ctypes.byref(of), ctypes.byref(nf), ctypes.byref(st))
    if rc != 0:
        names = {nw.NW_ERR_BADARG: ERRORS[-1], nw.NW_ERR_NONMANIFOLD: ERRORS[-2], nw.NW_ERR_NOMEM: ERRORS[-3], nw.NW_ERR_HIP: 'a HIP call failed (no GPU?)',
                 nw.NW_ERR_INTERNAL: 'the half-edge structure broke (a fan that does not close)'}
        raise RuntimeError('nw_remesh_device: %s' % names.get(rc, 'error %d' % rc))
    try:
        out_v = np.ctypeslib.as_array(ctypes.cast(ov, ctypes.POINTER(ctypes.c_float)), shape=(nv.value, 3)).copy()
        out_f = np.ctypeslib.as_array(ctypes.cast(of, ctypes.POINTER(ctypes.c_int32)), shape=(nf.value, 3)).copy()
    finally:
        L.nw_host_free(ov)
        L.nw_host_free(of)
    if return_stats:
        return out_v, out_f, dict(n_split=st.n_split, n_collapse=st.n_collapse, n_flip=st.n_flip, mean_edge_length=st.mean_edge_length, max_valence=st.max_valence,
                                  rounds=(st.rounds_split, st.rounds_collapse, st.rounds_flip))
    return out_v, out_f


def device_remesher(mesh, n=5, target_edge_length=-1, l=0.5, n_relax=10):
    """`MembraneMesh.remesher = 'device'`: the block boundary's remesh on the GPU (the reference's call has n_relax = 0: _membrane_mesh.pyx:1546)."""
    builtin_remesher(mesh, n, target_edge_length, l, n_relax, _remesh=lambda v, f, n, t, l, r: remesh_device(v, f, n, t, l, r, return_stats=True, device=getattr(mesh, '_device', 0) or 0))


def builtin_remesher(mesh, n=5, target_edge_length=-1, l=0.5, n_relax=10, _remesh=None):
    """`MembraneMesh.remesher` hook: remesh the valid part of `mesh` and rebuild its half-edge tables in place."""
    # (asked in a way that does not make a TriMesh with lazy topology build its half-edge records: the remesher works from the faces)
    valid = mesh.valid_vertex_mask() if hasattr(mesh, 'valid_vertex_mask') else mesh._vertices['halfedge'] != -1
    pos = mesh.vertices if hasattr(mesh, 'valid_vertex_mask') else mesh._vertices['position']
    # (the block that has just ended returned the same positions as one contiguous array; while nothing has touched the host mesh since --
    # the driver's mesh_key says so -- that array saves gathering 12-byte rows out of 120-byte records)
    cg, nat = getattr(mesh, 'cg', None), getattr(mesh, '_native', None)
    key = getattr(nat, 'mesh_key', None)
    fs = getattr(cg, 'fs', None)
    if key is not None and key[0] == id(mesh) and isinstance(fs, np.ndarray) and fs.dtype == np.float32 and fs.shape == pos.shape and fs.flags.c_contiguous:
        pos = fs
    if valid.all():                                   # (the usual case: no spare or deleted vertex slots -- nothing to renumber)
        v, f = pos, mesh.faces
    else:
        remap = np.cumsum(valid) - 1
        v = pos[valid]
        f = remap[mesh.faces]
    if _remesh is None:
        nv, nf, st = remesh(v, f, n, target_edge_length, l, n_relax, return_stats=True)
    else:
        nv, nf, st = _remesh(v, f, n, target_edge_length, l, n_relax)
    try:
        # (both remeshers drop the vertices no face refers to, and both report the mean edge length of what they return)
        mesh._topology_changed(nv, nf, all_referenced=True, mean_edge=st.get('mean_edge_length'))
    except TypeError:                                                 # (a mesh class of the caller's with the two-argument hook)
        mesh._topology_changed(nv, nf)
