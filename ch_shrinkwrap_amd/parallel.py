"""
Multi-GPU form of the NanoWrap iteration: one process per GPU, `torch.distributed` over RCCL (xGMI).

The reference is single-process (no collective anywhere); what shards naturally is the
localization cloud, because A is row-separable -- each localization touches the three vertices of ONE face
(SURVEY.md section 8e).  Two exact decompositions are provided, both driving the SAME split-phase C-ABI
(nw_iter_attract / nw_iter_directions / nw_iter_update, include/nanowrap.h):

  mode 'tiles'       every rank owns a spatial tile of the scene = its localizations AND the mesh component(s) inside
                     it; tiles share no vertices (BASELINE.json configs[4]: independent vesicles).  The boundary set is
                     empty, so no vertex data is exchanged; but the reference solves ONE <=3x3 subspace system for the
                     whole mesh (conj_grad.py:202-219), so the 24 normal-equation partial sums are all-reduced once per
                     iteration and every rank solves the same system.
  mode 'replicated'  the mesh is replicated, only the localizations are sharded.  Per iteration: all-reduce(sum) of the
                     per-vertex accumulator {A^T res, A^T 1} (M x 4 float32) and of the 13 point-side scalars.

  mode 'halo'        ONE mesh, sharded (BASELINE.json north_star "all-reduce on the boundary-vertex forces
                     only"): space is cut into tiles by recursive bisection of the localization cloud; a rank holds the
                     localizations of its tile, the faces whose centroid lies within one search radius H of the tile, their
                     vertices (on which it runs the complete iteration) and one more ring of ghost vertices (positions and
                     normals only, for the curvature prior).  A vertex is OWNED by the tile that contains it.  Per iteration:
                       (1) all-reduce(sum) of the rows of the fixed-point accumulator {A^T res, sum w} that belong to BOUNDARY
                           vertices -- vertices present on more than one rank -- packed into one dense buffer;
                       (2) all-reduce(sum) of the normal-equation scalars (vertex-side sums run over owned vertices only);
                       (3) the owners' new positions of the boundary vertices, same buffer shape, owner-only non-zero rows.
                     Every rank solves the same <=3x3 system; integer accumulators make the shared rows bit-identical everywhere.
                     The nearest-face query stays exact as long as no localization is further than H from its nearest centroid
                     (checked every iteration from the device-side maximum).

Collectives are issued on the stream the kernels run on (the optimiser is constructed with torch's current stream), so
an iteration is kernels -> all-reduce -> kernels with no host synchronisation.  Messages are KB..MB sized: the per-link
xGMI bandwidth is irrelevant for the scalar exchange (latency-bound), and the vertex accumulator is one bucket.

The orchestration below is backend-agnostic: it talks to an "executor" (HipExecutor for the product; tests drive the
same code over gloo with a CPU executor built from the oracle) so that the N > 1 protocol is covered on CPU.
"""
import ctypes
import numpy as np

from . import _lib as nw


class _DevArray(object):
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned device memory without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


class HipExecutor(object):
    """Phases of one iteration on this rank's nw_ctx + torch views of the buffers that get all-reduced."""

    def __init__(self, cg):
        self.cg = cg
        self.L, self.h, self.native = cg._L, cg._h, cg._native
        stride = self.L.nw_scalar_stride()          # doubles per slot (its ordered partial sums, k_reduce_scalars)
        self.n_point_scalars = self.L.nw_n_point_scalars() * stride
        self.n_scalars = self.L.nw_n_scalars() * stride
        self._views = {}

    def new_tensor(self, values):
        import torch
        return torch.tensor(values, dtype=torch.float64, device='cuda')

    def begin(self, data, lams, num_iters, sigma_inv, weights, prenormalized, pos, last_step):
        cg = self.cg
        cg._upload_points(sigma_inv, weights, prenormalized)
        lams_a = np.ascontiguousarray(lams, dtype=np.float32)
        flags = (nw.NW_FLAG_POSITIVITY if pos else 0) | (0 if last_step else nw.NW_FLAG_NO_LAST_STEP) | cg._regulariser_flag()
        cg._cache = {}
        self.native.check(self.L.nw_search_begin(self.h, nw.ptr(lams_a), lams_a.size, int(num_iters), flags))
        self._num_iters = int(num_iters)

    def attract(self):
        self.native.check(self.L.nw_iter_attract(self.h))

    def directions(self):
        self.native.check(self.L.nw_iter_directions(self.h))

    def update(self):
        self.native.check(self.L.nw_iter_update(self.h))

    def _view(self, what, count, typestr):
        import torch
        p, nb = ctypes.c_void_p(), ctypes.c_int64()
        self.native.check(self.L.nw_device_ptr(self.h, what, ctypes.byref(p), ctypes.byref(nb)))
        key = (p.value, count, typestr)
        if key not in self._views:
            self._views[key] = torch.as_tensor(_DevArray(p.value, (count,), typestr), device='cuda')
        return self._views[key]

    def scalars(self, count):
        # the 24 sums of the current iteration (first nw_n_point_scalars(): point side)
        return self._view(nw.NW_ARR_SCALARS, count, '<f8')

    def vertex_accumulator(self):
        # (M, 4) int64 fixed-point sums {A^T res, sum w}: integer all-reduce = exact, order independent
        return self._view(nw.NW_ARR_VACC, 4 * self.cg.M, '<i8')

    # -- 'halo' mode: boundary rows <-> one dense buffer over the global boundary list ----------------------------------
    def set_boundary(self, b_local, b_slot, n_boundary, owned_local):
        """b_local: local ids of this rank's vertices that are boundary vertices; b_slot: their rows in the global boundary
        list (length n_boundary); owned_local: uint8 (M_local) ownership flags."""
        import torch
        dev = torch.device('cuda', torch.cuda.current_device())
        self.b_local = torch.as_tensor(np.ascontiguousarray(b_local, dtype=np.int64), device=dev)
        self.b_slot = torch.as_tensor(np.ascontiguousarray(b_slot, dtype=np.int64), device=dev)
        own_b = np.asarray(owned_local, bool)[np.asarray(b_local, np.int64)]
        self.b_owned = torch.as_tensor(np.ascontiguousarray(own_b), device=dev)
        self.n_boundary = int(n_boundary)
        self.native.check(self.L.nw_set_owned(self.h, nw.ptr(np.ascontiguousarray(owned_local, dtype=np.uint8))))

    def pack_boundary_accumulator(self):
        import torch
        acc = self.vertex_accumulator().view(-1, 4)
        buf = torch.zeros((self.n_boundary, 4), dtype=acc.dtype, device=acc.device)
        buf[self.b_slot] = acc[self.b_local]
        return buf

    def unpack_boundary_accumulator(self, buf):
        self.vertex_accumulator().view(-1, 4)[self.b_local] = buf[self.b_slot]

    def _positions(self):
        M = self.cg.M
        return self._view(nw.NW_ARR_POS, 3 * M, '<f4').view(-1, 3), self._view(nw.NW_ARR_MESHPOS, 3 * M, '<f4').view(-1, 3)

    def pack_owned_boundary_positions(self):
        import torch
        pos, _ = self._positions()
        buf = torch.zeros((self.n_boundary, 3), dtype=pos.dtype, device=pos.device)
        rows = pos[self.b_local]
        buf[self.b_slot] = torch.where(self.b_owned[:, None], rows, torch.zeros_like(rows))
        return buf

    def unpack_boundary_positions(self, buf):
        # every holder takes the owner's value: ghosts get their update, shared computed copies cannot drift
        pos, meshpos = self._positions()
        rows = buf[self.b_slot]
        pos[self.b_local] = rows
        meshpos[self.b_local] = rows

    def quantum(self, value=0.0):
        """quantum of the fixed-point accumulator; value > 0 fixes it (ranks that all-reduce the accumulator must agree)"""
        q = ctypes.c_double(float(value))
        self.native.check(self.L.nw_accumulator_quantum(self.h, ctypes.byref(q)))
        return q.value

    def end(self):
        cg = self.cg
        logs = (nw.IterLog * max(self._num_iters, 1))()
        lc = ctypes.c_int(0)
        code = self.L.nw_search_end(self.h, None, logs, ctypes.byref(lc))
        self.native.check(code)
        cg.max_dist = 0.0
        cg._consume_logs(logs, lc.value)
        cg._accumulate_stage_ms()
        cg._finish()
        self.max_dist = cg.max_dist
        return cg.fs


def run_search(ex, dist, mode, data, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True):
    """One search() call of `num_iters` iterations over all ranks of `dist` (a torch.distributed-like module with an
    initialised default group).  Every rank calls this collectively with its own executor."""
    if mode not in ('tiles', 'replicated', 'halo'):
        raise ValueError(mode)
    # weights = weights / weights.mean() (mesh_conj_grad.py:160-162): the mean runs over the WHOLE scene, i.e. all ranks
    w_eff = weights if weights is not None else sigma_inv
    prenorm = None
    if not np.isscalar(w_eff):
        # once per weights OBJECT (every rank passes the same object again for the next block of a fit, so all ranks hit or
        # miss together): the global mean costs a blocking all-reduce and a pass over 3N floats on the host
        cached = getattr(ex, '_prenorm_cache', None)
        if cached is not None and cached[0] is w_eff:
            prenorm = cached[1]
        else:
            w_arr = np.asarray(w_eff, dtype=np.float32).ravel()
            t = ex.new_tensor([float(w_arr.astype(np.float64).sum()), float(w_arr.size)])
            dist.all_reduce(t)
            prenorm = (w_arr / np.float32(float(t[0]) / float(t[1]))).astype(np.float32)
            ex._prenorm_cache = (w_eff, prenorm)
    ex.begin(data, lams, num_iters, sigma_inv, weights, prenorm, pos, last_step)
    if mode != 'tiles' and hasattr(ex, 'quantum'):
        # the ranks add their integer accumulators: one common quantum (the coarsest any rank chose for its own localizations)
        t = ex.new_tensor([ex.quantum()])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ex.quantum(float(t[0]))
    n_red = ex.n_point_scalars if mode == 'replicated' else ex.n_scalars
    timer = getattr(ex, 'collective_timer', None)        # optional (bench.py): device time spent inside the collectives

    def all_reduce(t):
        if timer is None:
            dist.all_reduce(t)
        else:
            with timer:
                dist.all_reduce(t)

    for _ in range(int(num_iters)):
        ex.attract()
        if mode == 'replicated':
            all_reduce(ex.vertex_accumulator())
        elif mode == 'halo':
            buf = ex.pack_boundary_accumulator()             # (|B|, 4): this rank's partial sums of the boundary vertices it holds
            all_reduce(buf)
            ex.unpack_boundary_accumulator(buf)
        ex.directions()
        all_reduce(ex.scalars(n_red))
        ex.update()
        if mode == 'halo':
            buf = ex.pack_owned_boundary_positions()         # (|B|, 3): rows of the boundary vertices this rank OWNS, zero elsewhere
            all_reduce(buf)
            ex.unpack_boundary_positions(buf)
    return ex.end()


class CollectiveTimer(object):
    """Context manager that brackets collectives with events on the current torch stream; total_ms() synchronises and sums."""

    def __init__(self):
        self.pairs = []
        self.ms = 0.0
        self.count = 0

    def __enter__(self):
        import torch
        self._a = torch.cuda.Event(enable_timing=True)
        self._a.record()

    def __exit__(self, *exc):
        import torch
        b = torch.cuda.Event(enable_timing=True)
        b.record()
        self.pairs.append((self._a, b))

    def total_ms(self):
        import torch
        torch.cuda.synchronize()
        for a, b in self.pairs:
            self.ms += a.elapsed_time(b)
            self.count += 1
        self.pairs = []
        return self.ms, self.count


class TiledScene(object):
    """Convenience front end used by bench.py: single-GPU -> plain cg.search; multi-GPU -> run_search.

    Stream discipline for N > 1: kernels and collectives must share ONE stream.  torch's default stream has the handle 0,
    which the C-ABI reads as "use your own stream", so the optimiser must be constructed on a dedicated `torch.cuda.Stream`
    (`stream=s.cuda_stream`) and that same torch stream is made current around every collective here."""

    def __init__(self, cg, dist=None, mode='tiles', torch_stream=None):
        self.cg = cg
        self.dist = dist
        self.mode = mode
        self.torch_stream = torch_stream
        if dist is not None and torch_stream is None:
            raise ValueError('multi-GPU runs need the torch stream the optimiser was constructed on (see class docstring)')
        self.ex = HipExecutor(cg) if dist is not None else None

    def search(self, data, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True):
        if self.dist is None:
            return self.cg.search(data, lams=lams, num_iters=num_iters, sigma_inv=sigma_inv, weights=weights, pos=pos, last_step=last_step)
        import torch
        if type(lams) is float or np.isscalar(lams):
            lams = [float(lams)]
        with torch.cuda.stream(self.torch_stream):
            return run_search(self.ex, self.dist, self.mode, data, lams, num_iters, sigma_inv, weights, pos, last_step)


def partition_by_tiles(points, n_ranks):
    """Spatial tiling of a localization cloud for mode 'replicated': recursive splits along the longest axis, each split
    dividing the counts in proportion to the ranks on either side (balanced counts for any n_ranks, compact tiles ->
    each rank's scatter touches a compact set of vertices).  Returns a list of index arrays, one per rank."""
    def split(part, k):
        if k == 1:
            return [part]
        p = points[part]
        ax = int(np.argmax(p.max(0) - p.min(0))) if part.size else 0
        order = np.argsort(p[:, ax], kind='stable')
        kl = k // 2
        cut = (part.size * kl) // k
        return split(part[order[:cut]], kl) + split(part[order[cut:]], k - kl)
    return split(np.arange(points.shape[0]), int(n_ranks))


# =====================================================================================================================
# 'halo' mode: one mesh sharded over the ranks by spatial tiles of the localization cloud
# =====================================================================================================================
def bisect_tiles(points, n_ranks):
    """Recursive bisection of the cloud (the cuts of partition_by_tiles kept as a tree).  Returns (parts, classify): `parts` = one
    index array per rank, `classify(xyz)` = the rank whose tile contains each position -- a partition of SPACE, so every mesh vertex
    has exactly one owner, found with the same cuts on every rank."""
    cuts = []

    def split(part, k, base):
        if k == 1:
            return [part], ('leaf', base)
        p = points[part]
        ax = int(np.argmax(p.max(0) - p.min(0))) if part.size else 0
        order = np.argsort(p[:, ax], kind='stable')
        kl = k // 2
        cut = (part.size * kl) // k
        lo, hi = part[order[:cut]], part[order[cut:]]
        if lo.size and hi.size:
            plane = 0.5 * (float(points[lo[-1], ax]) + float(points[hi[0], ax]))
        else:
            plane = float(p[:, ax].mean()) if part.size else 0.0
        pl, tl = split(lo, kl, base)
        pr, tr = split(hi, k - kl, base + kl)
        return pl + pr, ('node', ax, plane, tl, tr)

    parts, tree = split(np.arange(points.shape[0]), int(n_ranks), 0)

    def classify(xyz):
        xyz = np.asarray(xyz)
        out = np.empty(xyz.shape[0], np.int32)

        def walk(node, idx):
            if node[0] == 'leaf':
                out[idx] = node[1]
                return
            _, ax, plane, tl, tr = node
            left = xyz[idx, ax] < plane
            walk(tl, idx[left])
            walk(tr, idx[~left])
        walk(tree, np.arange(xyz.shape[0]))
        return out
    return parts, classify


class HaloPartition(object):
    """Host-side decomposition of ONE mesh for the 'halo' mode, computed identically on every rank (no communication).

    For rank r: its localizations (tile r of the bisection); the faces whose centroid lies within `halo` of the tile's bounding
    box (every centroid closer than `halo` to one of its localizations is among them, so the local nearest-face query is the
    global one as long as no nearest distance exceeds `halo`); V_r = their vertices, on which the rank runs the complete
    iteration; W_r = further 1-ring neighbours of V_r, ghosts that only carry positions / normals for the curvature prior.
    A vertex is owned by the tile that contains it.  Boundary vertices = present (in V or W) on more than one rank."""

    def __init__(self, pos, nrm, nbr, faces, points, n_ranks, halo):
        pos = np.asarray(pos, np.float32)
        faces = np.asarray(faces, np.int32)
        nbr = np.asarray(nbr, np.int32)
        M = pos.shape[0]
        self.M, self.n_ranks, self.halo = M, int(n_ranks), float(halo)
        self.parts, classify = bisect_tiles(points, n_ranks)
        self.owner = classify(pos)
        cent = ((pos[faces[:, 0]] + pos[faces[:, 1]]) + pos[faces[:, 2]]) / np.float32(3.0)
        count = np.zeros(M, np.int32)
        self.ranks = []
        for r in range(self.n_ranks):
            p = points[self.parts[r]]
            if p.shape[0]:
                lo, hi = p.min(0) - self.halo, p.max(0) + self.halo
                fsel = np.nonzero(((cent >= lo) & (cent <= hi)).all(1))[0]
            else:
                fsel = np.zeros(0, np.int64)
            # the faces of every owned vertex belong to the owner as well (its prior and its update must be complete there)
            own_v = np.nonzero(self.owner == r)[0]
            if own_v.size:
                isown = np.zeros(M, bool)
                isown[own_v] = True
                fsel = np.union1d(fsel, np.nonzero(isown[faces].any(1))[0])
            gV = np.unique(faces[fsel].ravel()) if fsel.size else np.zeros(0, np.int64)
            gV = np.union1d(gV, own_v)
            ring = nbr[gV]
            ring = np.unique(ring[ring >= 0]) if gV.size else np.zeros(0, np.int64)
            gW = np.setdiff1d(ring, gV)
            gv = np.concatenate([gV, gW]).astype(np.int64)
            g2l = -np.ones(M, np.int64)
            g2l[gv] = np.arange(gv.size)
            nbr_l = -np.ones((gv.size, nbr.shape[1]), np.int32)
            rows = nbr[gV]
            nbr_l[:gV.size] = np.where(rows >= 0, g2l[np.where(rows >= 0, rows, 0)], -1)
            valid = np.zeros(gv.size, np.uint8)
            valid[:gV.size] = 1
            owned = (self.owner[gv] == r).astype(np.uint8)
            owned[gV.size:] = 0
            count[gv] += 1
            self.ranks.append(dict(pidx=self.parts[r], gv=gv, nV=int(gV.size), faces=g2l[faces[fsel]].astype(np.int32),
                                   nbr=nbr_l, valid=valid, owned=owned))
        self.boundary = np.nonzero(count > 1)[0]
        slot = -np.ones(M, np.int64)
        slot[self.boundary] = np.arange(self.boundary.size)
        for d in self.ranks:
            s = slot[d['gv']]
            d['b_local'] = np.nonzero(s >= 0)[0]
            d['b_slot'] = s[s >= 0]


class ArrayMesh(object):
    """The few attributes ShrinkwrapMeshConjGrad reads from a mesh object, over plain arrays (a rank's sub-mesh in 'halo' mode):
    `neighbors` holds 1-ring VERTEX ids and `_halfedges['vertex']` is the identity, so the optimiser's
    `_halfedges['vertex'][_vertices['neighbors']]` (mesh_conj_grad.py:50) gives the table back."""

    def __init__(self, pos, nrm, nbr, faces, valid):
        M, NB = nbr.shape
        self._vertices = np.zeros(M, dtype=[('position', '3f4'), ('normal', '3f4'), ('halfedge', 'i4'), ('neighbors', '%di4' % NB)])
        self._vertices['position'] = pos
        self._vertices['normal'] = nrm
        self._vertices['halfedge'] = np.where(np.asarray(valid) != 0, 0, -1)
        self._vertices['neighbors'] = nbr
        self._halfedges = np.zeros(max(M, 1), dtype=[('vertex', 'i4')])
        self._halfedges['vertex'] = np.arange(max(M, 1))
        self.faces = np.ascontiguousarray(faces, np.int32)

    @property
    def vertices(self):
        return self._vertices['position']

    @property
    def vertex_normals(self):
        return self._vertices['normal']

    def _initialize_curvature_vectors(self):
        pass


class HaloScene(object):
    """Front end of the 'halo' mode: every rank holds the whole mesh on the HOST (it is small: 24 bytes per vertex) and its own
    share of everything on the DEVICE.  search() = one block: partition for the current mesh, one optimiser over the rank's
    sub-mesh, run_search(..., 'halo'), then one all-reduce of owner-only rows returns the new positions to every rank's host mesh.

    make_executor(local_mesh, local_points) -> executor (HipExecutor over a ShrinkwrapMeshConjGrad in production; the CPU tests pass
    an oracle-backed one).  `halo` = search radius in length units; the run raises if a nearest-face distance exceeds it."""

    def __init__(self, mesh, points, dist, halo, make_executor=None, native=None, torch_stream=None):
        self.mesh, self.points, self.dist, self.halo = mesh, np.ascontiguousarray(points, np.float32), dist, float(halo)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.make_executor = make_executor
        self.native = native
        self.torch_stream = torch_stream
        self.last_partition = None

    def _hip_executor(self, local_mesh, local_points):
        from .mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext
        if self.native is None:
            import torch
            self.native = NativeContext(torch.cuda.current_device(), self.torch_stream.cuda_stream if self.torch_stream is not None else None)
        self.native.mesh_key = None
        cg = ShrinkwrapMeshConjGrad(local_mesh, local_points, native=self.native)
        return HipExecutor(cg)

    def search(self, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True):
        import contextlib
        mesh = self.mesh
        nbr = mesh._halfedges['vertex'][mesh._vertices['neighbors']]
        nbr[mesh._vertices['neighbors'] == -1] = -1
        part = HaloPartition(mesh._vertices['position'], mesh.vertex_normals, nbr, mesh.faces, self.points, self.world, self.halo)
        self.last_partition = part
        d = part.ranks[self.rank]
        gv = d['gv']
        local = ArrayMesh(mesh._vertices['position'][gv], np.asarray(mesh.vertex_normals)[gv], d['nbr'], d['faces'], d['valid'])
        pidx = d['pidx']
        lp = np.ascontiguousarray(self.points[pidx])

        def take3(a):
            if a is None or np.isscalar(a):
                return a
            return np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, 3)[pidx].ravel())
        ex = (self.make_executor or self._hip_executor)(local, lp)
        ex.set_boundary(d['b_local'], d['b_slot'], part.boundary.size, d['owned'])
        if type(lams) is float or np.isscalar(lams):
            lams = [float(lams)]
        ctx = contextlib.nullcontext()
        if self.torch_stream is not None:
            import torch
            ctx = torch.cuda.stream(self.torch_stream)
        with ctx:
            out = run_search(ex, self.dist, 'halo', lp, lams, num_iters, take3(sigma_inv), take3(weights), pos, last_step)
            # exactness of the sharded nearest-face query: no localization further than the halo from its nearest centroid
            worst = ex.new_tensor([float(getattr(ex, 'max_dist', 0.0))])
            self.dist.all_reduce(worst, op=self.dist.ReduceOp.MAX)
            if float(worst[0]) > self.halo:
                raise RuntimeError("halo mode: a localization is %.3g from its nearest face centroid, beyond the halo radius %.3g: "
                                   "the sharded query is not guaranteed exact (increase `halo`)" % (float(worst[0]), self.halo))
            # owners return their rows; one all-reduce per BLOCK gives every rank the whole new mesh
            full = np.zeros((part.M, 3), np.float64)
            own = d['owned'].astype(bool)
            full[gv[own]] = np.asarray(out, np.float64)[own]
            t = ex.new_tensor(full.ravel())
            self.dist.all_reduce(t)
        newpos = np.asarray(t.cpu()).reshape(-1, 3).astype(np.float32)
        valid = mesh._vertices['halfedge'] != -1
        mesh._vertices['position'][valid] = newpos[valid]
        self.ex = ex
        return mesh._vertices['position'].copy()
