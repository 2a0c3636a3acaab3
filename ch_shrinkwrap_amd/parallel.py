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
                     localizations of its tile, the faces whose centroid lies within (nearest distance + margin) of each of its
                     localizations (or within one radius H of the tile), their vertices (on which it runs the complete iteration)
                     and one more ring of ghost vertices (positions and normals only, for the curvature prior).  A vertex is OWNED
                     by the tile that contains it.  Per iteration, for the vertices more than one rank holds:
                       (1) the rows of the fixed-point accumulator {A^T res, sum w}: the copies' partial sums go to the vertex's
                           owner, which adds them and sends the sum back to the copies (owner-wise exchange between the ranks that
                           share vertices; exchange='dense': one all-reduce over the global list of such vertices);
                       (2) all-reduce(sum) of the normal-equation scalars (vertex-side sums run over owned vertices only);
                       (3) the owners' new positions to the copies (dense: owner-only non-zero rows, all-reduced).
                     Every rank solves the same <=3x3 system; integer accumulators make the shared rows bit-identical everywhere.
                     The nearest-face query stays exact while (growth of any nearest distance + drift of the mesh) since the
                     shares were cut stays within the margin (one radius: largest nearest distance + drift within H), checked on
                     the device after every block.

Who issues the collectives.  In production the LIBRARY does: every rank joins an RCCL communicator of its own nw_ctx (NativeComm ->
nw_comm_init) and a block is ONE nw_search call with an NW_FLAG_COMM_* flag -- phases and ncclAllReduce on the ctx's own stream,
recorded into the block's hipGraph like every kernel launch (include/nanowrap.h "multi-GPU: RCCL inside the library").  Python keeps
the partition and the set-up; torch.distributed is only the host channel that carries the communicator's id (and whatever barriers
the application wants).  Messages are KB..MB sized: the per-link xGMI bandwidth is irrelevant for the scalar exchange
(latency-bound), and the vertex accumulator is one bucket.

The same protocol over an EXTERNAL process group -- the split-phase C-ABI (nw_iter_attract / nw_iter_directions / nw_iter_update)
with the caller's all-reduces between the phases -- is what run_search does for an executor without a communicator: the CPU tests
drive it over gloo with an executor built from the oracle, and the GPU tests with two gloo ranks sharing one GPU (RCCL refuses two
ranks on one device), so the N > 1 protocol is covered where no multi-GPU node is available.
"""
import ctypes
import numpy as np

from . import _lib as nw


class NativeComm(object):
    """This rank's RCCL communicator inside the library (nw_comm_init), created over a NativeContext.  `dist`: any initialised
    torch.distributed-like module -- used ONCE, to hand rank 0's unique id to the others (and for rank / world size)."""

    SUM, MAX = 0, 1
    _DT = {np.dtype('float32'): 0, np.dtype('float64'): 1, np.dtype('int64'): 2, np.dtype('int32'): 3}

    def __init__(self, native, dist):
        self.native, self.L, self.h = native, native.L, native.h
        self.rank, self.world = int(dist.get_rank()), int(dist.get_world_size())
        ident = np.zeros(128, np.uint8)
        if self.rank == 0:
            native.check(self.L.nw_comm_unique_id(nw.ptr(ident), ident.size))
        box = [ident.tobytes()]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
        ident = np.frombuffer(box[0], np.uint8).copy()
        native.check(self.L.nw_comm_init(self.h, nw.ptr(ident), ident.size, self.rank, self.world))

    def all_reduce_host(self, a, op=0):
        """in place over a C-contiguous host array (float32 / float64 / int64 / int32); blocking"""
        self.native.check(self.L.nw_comm_all_reduce(self.h, nw.ptr(a), a.size, self._DT[a.dtype], int(op)))
        return a

    def all_reduce_device(self, what, count, dtype, op=0):
        """in place over a library-owned device array (NW_ARR_*), asynchronous on the ctx's stream"""
        p, nb = ctypes.c_void_p(), ctypes.c_int64()
        self.native.check(self.L.nw_device_ptr(self.h, what, ctypes.byref(p), ctypes.byref(nb)))
        self.native.check(self.L.nw_comm_all_reduce(self.h, p, int(count), self._DT[np.dtype(dtype)], int(op)))

    def exchange(self, elems_per_row, dtype, owned_out=True):
        """one neighbour exchange of a sharded mesh's peers outside a block (nw_set_boundary with peers): NW_ARR_PEER_SEND goes out,
        NW_ARR_PEER_RECV comes in; asynchronous on the ctx's stream"""
        self.native.check(self.L.nw_comm_all_reduce(self.h, None, int(elems_per_row), self._DT[np.dtype(dtype)], 2 if owned_out else 3))

    def close(self):
        self.native.check(self.L.nw_comm_init(self.h, None, 0, 0, 0))


_COMM_FLAG = {'tiles': nw.NW_FLAG_COMM_TILES, 'replicated': nw.NW_FLAG_COMM_REPLICATED, 'halo': nw.NW_FLAG_COMM_HALO}


class _DevArray(object):
    """Minimal __cuda_array_interface__ carrier so torch can view library-owned device memory without a copy."""

    def __init__(self, ptr, shape, typestr):
        self.__cuda_array_interface__ = {'shape': tuple(shape), 'typestr': typestr, 'data': (int(ptr), False), 'version': 2}


class HipExecutor(object):
    """Phases of one iteration on this rank's nw_ctx + torch views of the buffers that get all-reduced.  Nothing here allocates device
    memory or launches a torch kernel per iteration: the boundary rows of a sharded mesh are packed / taken by the library itself
    (nw_set_boundary), the views of its exchange buffers are made once."""

    def __init__(self, cg):
        self.cg = cg
        self.L, self.h, self.native = cg._L, cg._h, cg._native
        stride = self.L.nw_info(nw.NW_INFO_SCALAR_STRIDE)          # doubles per slot (its ordered partial sums, k_reduce_scalars)
        self.n_point_scalars = self.L.nw_info(nw.NW_INFO_POINT_SCALARS) * stride
        self.n_scalars = self.L.nw_info(nw.NW_INFO_SCALARS) * stride
        self._views = {}
        self.write_back = True                      # end(): copy the rank's (M,3) result to the host mesh (a sharded mesh gathers the whole mesh instead)
        self.max_dist = 0.0
        self.comm = None                            # NativeComm: the library issues the block's collectives itself (run_search)
        self.blocks_run = 0

    def new_tensor(self, values):
        import torch
        return torch.tensor(values, dtype=torch.float64, device='cuda')

    def begin(self, data, lams, num_iters, sigma_inv, weights, prenormalized, pos, last_step):
        cg = self.cg
        cg._upload_points(sigma_inv, weights, prenormalized)
        lams_a = np.ascontiguousarray(lams, dtype=np.float32)
        flags = (nw.NW_FLAG_POSITIVITY if pos else 0) | (0 if last_step else nw.NW_FLAG_NO_LAST_STEP) | cg._regulariser_flag()
        posv = cg.mesh._vertices['position']
        # the result goes straight from the block's last update kernel to the host (as in cg.search) when end() copies it into the mesh
        self._direct = bool(self.write_back and posv.dtype == np.float32 and posv.strides[1] == 4 and posv.strides[0] >= 12)
        if self._direct:
            flags |= nw.NW_FLAG_RESULT_TO_HOST
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
        # the sums of the current iteration (first NW_INFO_POINT_SCALARS slots: point side + the status slot)
        return self._view(nw.NW_ARR_SCALARS, count, '<f8')

    def vertex_accumulator(self):
        # (M, 4) int64 fixed-point sums {A^T res, sum w}: integer all-reduce = exact, order independent
        return self._view(nw.NW_ARR_VACC, 4 * self.cg.M, '<i8')

    # -- 'halo' mode: boundary rows <-> one dense buffer over the global boundary list (packed / taken inside the phases) ------------
    def set_boundary(self, b_local, b_slot, n_boundary, owned_local, gv, n_global, peers=None):
        """b_local: local ids of this rank's vertices that are boundary vertices; b_slot: their rows in the global boundary
        list (length n_boundary); owned_local: uint8 (M_local) ownership flags; gv: global id of every local vertex.
        peers = (peer_rank, ghost_off, ghost_local, owned_off, owned_local) (HaloPartition.set_holders): the boundary rows then go
        between the ranks that share them (owner-wise exchange) and the dense list is not used."""
        ow = np.ascontiguousarray(owned_local, dtype=np.uint8)
        g = np.ascontiguousarray(gv, dtype=np.int32)
        self.peers = None
        if peers is not None:
            pr, go, gl, oo, ol = peers
            pr = np.ascontiguousarray(pr, np.int32)
            go, oo = np.ascontiguousarray(go, np.int64), np.ascontiguousarray(oo, np.int64)
            gl, ol = np.ascontiguousarray(gl, np.int32), np.ascontiguousarray(ol, np.int32)
            self.native.check(self.L.nw_set_boundary(self.h, None, None, 0, 0, nw.ptr(ow), nw.ptr(g), int(n_global),
                                                     pr.size, nw.ptr(pr), nw.ptr(go), nw.ptr(gl), nw.ptr(oo), nw.ptr(ol)))
            self.peers = (pr, go, oo)
            self.n_boundary = 0
        else:
            bl = np.ascontiguousarray(b_local, dtype=np.int32)
            bs = np.ascontiguousarray(b_slot, dtype=np.int32)
            self.native.check(self.L.nw_set_boundary(self.h, nw.ptr(bl), nw.ptr(bs), bl.size, int(n_boundary), nw.ptr(ow), nw.ptr(g), int(n_global),
                                                     -1, None, None, None, None, None))
            self.n_boundary = int(n_boundary)
        self.n_global = int(n_global)
        self._views = {}

    # -- owner-wise exchange: the segments of the library's send / receive buffers, peer by peer ---------------------------------------
    def peer_segments(self, kind):
        """[(peer rank, tensor to send, tensor to receive into)] of one neighbour exchange.  kind: 'acc_to_owners' (the copies' partial
        accumulator rows out, rows for the owned vertices in), 'acc_to_copies' (the owners' sums out / in), 'rows_to_copies' (the owners'
        float32 rows -- positions after update(), normals after refresh_normals_local() -- out / in)"""
        pr, go, oo = self.peers
        rows = int(max(go[-1], oo[-1], 1))
        if kind == 'rows_to_copies':
            e, snd, rcv = 3, self._view(nw.NW_ARR_PEER_SEND, 3 * rows, '<f4'), self._view(nw.NW_ARR_PEER_RECV, 3 * rows, '<f4')
        elif kind == 'acc_to_copies':                         # the sums go back as the four float32 the kernels convert them to
            e, snd, rcv = 4, self._view(nw.NW_ARR_PEER_SEND, 4 * rows, '<f4'), self._view(nw.NW_ARR_PEER_RECV, 4 * rows, '<f4')
        else:
            e, snd, rcv = 4, self._view(nw.NW_ARR_PEER_SEND, 4 * rows, '<i8'), self._view(nw.NW_ARR_PEER_RECV, 4 * rows, '<i8')
        so, ro = (go, oo) if kind == 'acc_to_owners' else (oo, go)
        return [(int(pr[k]), snd[e * int(so[k]):e * int(so[k + 1])], rcv[e * int(ro[k]):e * int(ro[k + 1])]) for k in range(pr.size)]

    def peer_step(self, what, step):
        """a step of the owner-wise exchange by hand (nw_halo_rows): the one between the two accumulator exchanges is the caller's"""
        self.native.check(self.L.nw_halo_rows(self.h, {'acc': nw.NW_ARR_VACC, 'pos': nw.NW_ARR_POS, 'nrm': nw.NW_ARR_NRM}[what], int(step)))

    def boundary_accumulator(self):
        """(n_boundary, 4) int64: this rank's partial sums of the boundary vertices it holds (filled by attract())"""
        return self._view(nw.NW_ARR_HALO_ACC, 4 * max(self.n_boundary, 1), '<i8')[:4 * self.n_boundary]

    def boundary_rows(self):
        """(n_boundary, 3) float32: rows of the boundary vertices this rank OWNS, zero elsewhere (filled by update(): the new positions)"""
        return self._view(nw.NW_ARR_HALO_ROWS, 3 * max(self.n_boundary, 1), '<f4')[:3 * self.n_boundary]

    def gather_owned(self, what='pos'):
        """(n_global * 3) float32: the owners' rows of the whole mesh, zero elsewhere -> one all-reduce per block"""
        self.native.check(self.L.nw_halo_gather_owned(self.h, nw.NW_ARR_POS if what == 'pos' else nw.NW_ARR_NRM))
        return self._view(nw.NW_ARR_HALO_FULL, 3 * self.n_global, '<f4')

    supports_reach = True

    def set_reference(self, full_positions, d0=None):
        """(M_global,3) float32: where the whole mesh was when the shares were cut (the drift of the halo is measured from it); d0 (N,):
        this rank's nearest distances then (shares cut with per-localization halos: max_dist becomes the largest growth of one)"""
        a = np.ascontiguousarray(full_positions, np.float32)
        d = None if d0 is None else np.ascontiguousarray(d0, np.float32)
        self.native.check(self.L.nw_halo_set_reference(self.h, nw.ptr(a), nw.ptr(d), 0 if d is None else d.size))

    def block_stats(self, max_dist):
        """(4,) float32 on the device: {largest nearest distance, this rank's quantum, max drift^2 of gather_owned()'s (all-reduced)
        array against the reference, 0} -> one all-reduce(MAX) per block"""
        self.native.check(self.L.nw_halo_block_stats(self.h, float(max_dist)))
        return self._view(nw.NW_ARR_HALO_STATS, 4, '<f4')

    def refresh_normals_local(self, extent):
        """block-boundary refresh (_membrane_mesh.pyx:1524-1527) of this rank's share on the device; the owners' normals of the boundary
        vertices are left in boundary_rows() for the all-reduce, take_normals() then gives every holder the owner's"""
        self.native.check(self.L.nw_refresh_normals(self.h, None, float(extent)))
        self.native.check(self.L.nw_halo_rows(self.h, nw.NW_ARR_NRM, 0))

    def take_normals(self):
        self.native.check(self.L.nw_halo_rows(self.h, nw.NW_ARR_NRM, 1))
        self.native.check(self.L.nw_reset_history(self.h))          # a new optimiser per block (_membrane_mesh.pyx:1510)
        self.cg.tests, self.cg.ress, self.cg.prefs = [], [], []

    def host_copy_rows(self, src, contiguous, rows, valid_u8):
        """(M,3) float32 host array -> a contiguous copy and the strided vertex records, by the library's copy threads"""
        self.native.check(self.L.nw_host_copy_rows(self.h, nw.ptr(src), src.shape[0], nw.ptr(contiguous), ctypes.c_void_p(rows.ctypes.data), rows.strides[0],
                                                   nw.ptr(valid_u8)))

    def local_quantum(self):
        """the quantum this rank would choose for its own localizations (after begin(), or before a block once points and mesh are set)"""
        q = ctypes.c_double(0.0)
        self.native.check(self.L.nw_accumulator_quantum(self.h, ctypes.byref(q)))
        return q.value

    def set_quantum(self, value):
        """fix the quantum of the fixed-point accumulator (ranks that add their accumulators must agree)"""
        q = ctypes.c_double(float(value))
        self.native.check(self.L.nw_accumulator_quantum(self.h, ctypes.byref(q)))

    def end(self):
        cg = self.cg
        logs = (nw.IterLog * max(self._num_iters, 1))()
        lc = ctypes.c_int(0)
        out = None
        if self._direct:
            out = cg._result_buffer()
            posv = cg.mesh._vertices['position']
            self.native.check(self.L.nw_set_write_back(self.h, ctypes.c_void_p(posv.ctypes.data), posv.strides[0]))
        code = self.L.nw_search_end(self.h, nw.ptr(out) if out is not None else None, logs, ctypes.byref(lc))
        if self._direct:
            self.native.check(self.L.nw_set_write_back(self.h, None, 0))
        self.native.check(code)
        cg.max_dist = 0.0
        cg._consume_logs(logs, lc.value)
        cg._accumulate_stage_ms()
        self.max_dist = cg.max_dist
        if not self.write_back:
            return None
        if out is not None:
            cg.fs = out
            cg.f = out.ravel()
            cg.mesh._initialize_curvature_vectors()
        else:
            cg._finish()
        return cg.fs


def run_search(ex, dist, mode, data, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True, quantum=None):
    """One search() call of `num_iters` iterations over all ranks of `dist` (a torch.distributed-like module with an
    initialised default group).  Every rank calls this collectively with its own executor.
    quantum: the accumulator quantum all ranks agreed on earlier (a scene object carries it from block to block, piggy-backed on a
    collective it needs anyway); None = agree now (one blocking all-reduce MAX of the ranks' own values)."""
    if mode not in ('tiles', 'replicated', 'halo'):
        raise ValueError(mode)
    if getattr(ex, 'comm', None) is not None:
        return _run_search_native(ex, mode, data, lams, num_iters, sigma_inv, weights, pos, last_step, quantum)
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
    if mode != 'tiles' and hasattr(ex, 'set_quantum'):
        # the ranks add their integer accumulators: one common quantum (the coarsest any rank chooses for its own localizations)
        if quantum is None:
            t = ex.new_tensor([ex.local_quantum()])
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            quantum = float(t[0])
        ex.set_quantum(quantum)
    n_red = ex.n_point_scalars if mode == 'replicated' else ex.n_scalars
    timer = getattr(ex, 'collective_timer', None)        # optional (bench.py): device time spent inside the collectives

    def all_reduce(t):
        if timer is None:
            dist.all_reduce(t)
        else:
            with timer:
                dist.all_reduce(t)

    peers = mode == 'halo' and getattr(ex, 'peers', None) is not None

    def iteration():
        ex.attract()                                         # 'halo': leaves this rank's partial sums of the boundary vertices packed
        if mode == 'replicated':
            all_reduce(ex.vertex_accumulator())
        elif peers:
            peer_exchange(dist, ex, 'acc_to_owners', timer)  # the copies' partial rows to the vertices' owners ...
            ex.peer_step('acc', 1)                           # ... added there ...
            peer_exchange(dist, ex, 'acc_to_copies', timer)  # ... and the sums back to the copies (directions() takes them)
        elif mode == 'halo' and ex.n_boundary > 0:
            all_reduce(ex.boundary_accumulator())            # (|B|, 4) int64
        ex.directions()                                      # 'halo': takes the summed boundary rows first
        all_reduce(ex.scalars(n_red))
        ex.update()                                          # 'halo': leaves the new positions of the boundary vertices it OWNS packed
        if peers:
            peer_exchange(dist, ex, 'rows_to_copies', timer)
        elif mode == 'halo' and ex.n_boundary > 0:
            all_reduce(ex.boundary_rows())                   # (|B|, 3): owner-only non-zero rows -> every holder takes the owner's value

    for _ in range(int(num_iters)):
        iteration()
    ex.blocks_run = getattr(ex, 'blocks_run', 0) + 1
    return ex.end()                                          # (the last update's rows are taken here)


def peer_exchange(dist, ex, kind, timer=None):
    """One neighbour exchange of the owner-wise scheme over an external process group: every peer's segment goes out and comes in as
    point-to-point messages (batch_isend_irecv).  Device buffers are staged through the host when the group cannot send them (gloo)."""
    import torch
    segs = ex.peer_segments(kind)
    if not segs:
        return
    stage = bool(segs[0][1].is_cuda) and str(dist.get_backend()) != 'nccl'
    ops, taken = [], []
    for peer, snd, rcv in segs:
        if snd.numel():
            ops.append(dist.P2POp(dist.isend, snd.cpu() if stage else snd, peer))
        if rcv.numel():
            buf = torch.empty(rcv.shape, dtype=rcv.dtype) if stage else rcv
            ops.append(dist.P2POp(dist.irecv, buf, peer))
            taken.append((rcv, buf))
    if not ops:
        return

    def go():
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        if stage:
            for rcv, buf in taken:
                rcv.copy_(buf)
    if timer is None:
        go()
    else:
        with timer:
            go()


def _run_search_native(ex, mode, data, lams, num_iters, sigma_inv, weights, pos, last_step, quantum):
    """The block of run_search for an executor with a NativeComm: set-up collectives through the communicator, then ONE nw_search call
    that runs the phases and the collectives between them on the library's own stream (captured as the block's hipGraph)."""
    comm, cg = ex.comm, ex.cg
    w_eff = weights if weights is not None else sigma_inv
    prenorm = None
    if not np.isscalar(w_eff):
        cached = getattr(ex, '_prenorm_cache', None)
        if cached is not None and cached[0] is w_eff:
            prenorm = cached[1]
        else:
            w_arr = np.asarray(w_eff, dtype=np.float32).ravel()
            t = comm.all_reduce_host(np.array([float(w_arr.astype(np.float64).sum()), float(w_arr.size)], np.float64))
            prenorm = (w_arr / np.float32(float(t[0]) / float(t[1]))).astype(np.float32)
            ex._prenorm_cache = (w_eff, prenorm)
    cg._upload_points(sigma_inv, weights, prenorm)
    if mode != 'tiles':
        if quantum is None:
            quantum = float(comm.all_reduce_host(np.array([ex.local_quantum()], np.float64), NativeComm.MAX)[0])
        ex.set_quantum(quantum)
    cg.max_dist = 0.0
    out = cg.search(data, lams, num_iters=num_iters, weights=weights, sigma_inv=sigma_inv, pos=pos, last_step=last_step,
                    comm_flags=_COMM_FLAG[mode], prenormalized=prenorm, to_host=ex.write_back)
    ex.max_dist = cg.max_dist
    ex.blocks_run = getattr(ex, 'blocks_run', 0) + 1
    return out


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

    `comm` (NativeComm over the optimiser's NativeContext): the library runs the collectives on its own stream.  Without it the
    collectives go through `dist` (an external process group) and kernels and collectives must share ONE stream: torch's default
    stream has the handle 0, which the C-ABI reads as "use your own stream", so the optimiser must then be constructed on a dedicated
    `torch.cuda.Stream` (`stream=s.cuda_stream`) and that same torch stream is made current around every collective here."""

    def __init__(self, cg, dist=None, mode='tiles', torch_stream=None, comm=None):
        self.cg = cg
        self.dist = dist
        self.mode = mode
        self.torch_stream = torch_stream
        self.comm = comm                       # NativeComm over cg's NativeContext: the library issues the collectives (production)
        if dist is not None and comm is None and torch_stream is None:
            raise ValueError('multi-GPU runs over an external process group need the torch stream the optimiser was constructed on (see class docstring)')
        self.ex = HipExecutor(cg) if (dist is not None or comm is not None) else None
        if self.ex is not None:
            self.ex.comm = comm

    def search(self, data, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True):
        if self.ex is None:
            return self.cg.search(data, lams=lams, num_iters=num_iters, sigma_inv=sigma_inv, weights=weights, pos=pos, last_step=last_step)
        if type(lams) is float or np.isscalar(lams):
            lams = [float(lams)]
        if self.comm is not None:
            return run_search(self.ex, None, self.mode, data, lams, num_iters, sigma_inv, weights, pos, last_step)
        import torch
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
THIN_CUTS_NM = 25.0          # HaloScene: half-width of the slab around a candidate cut in which localizations are counted (bisect_tiles)


def bisect_tiles(points, n_ranks, thin_cuts=None):
    """Recursive bisection of the cloud (the cuts of partition_by_tiles kept as a tree).  Returns (parts, classify): `parts` = one
    index array per rank, `classify(xyz)` = the rank whose tile contains each position -- a partition of SPACE, so every mesh vertex
    has exactly one owner, found with the same cuts on every rank.

    thin_cuts (nm): instead of always cutting across the longest axis, try all three axes at their balanced position and take the one with
    the fewest localizations within thin_cuts of the plane -- the shortest cut through the structure, i.e. the fewest shared vertices."""
    cuts = []

    def split(part, k, base):
        if k == 1:
            return [part], ('leaf', base)
        p = points[part]
        kl = k // 2
        cut = (part.size * kl) // k
        ax = int(np.argmax(p.max(0) - p.min(0))) if part.size else 0
        if thin_cuts and part.size > 1 and 0 < cut < part.size:
            best = None
            for a in range(3):
                col = p[:, a]
                srt = np.partition(col, [cut - 1, cut])
                plane_a = 0.5 * (float(srt[cut - 1]) + float(srt[cut]))
                near = int(np.count_nonzero(np.abs(col - plane_a) < thin_cuts))
                if best is None or near < best[0]:
                    best = (near, a)
            ax = best[1]
        order = np.argsort(p[:, ax], kind='stable')
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
    # bounding boxes of the tiles' localizations: like the tiles they depend on the cloud alone (a pass over every localization)
    classify.boxes = [(points[p].min(0), points[p].max(0)) if p.size else None for p in parts]
    return parts, classify


def faces_within_reach(cent, pts, reach, classes=16, voxel=None):
    """Mask of the centroids `cent` (F,3) that lie within reach_i of at least one point pts[i] -- a superset, never less: the points are
    put into classes of similar reach (geometric steps) and, per class, into voxels of a quarter of the class's reach; a centroid is taken
    when it is within (the class's largest reach + half a voxel diagonal) of an occupied voxel's centre.  One k-d tree over the
    occupied voxels' centres and one bounded nearest-neighbour query of the candidate centroids per class.  `voxel`: upper limit of the
    voxel edge (the slack it adds -- 0.87 voxel edges -- widens the halo of every localization of the class)."""
    from scipy.spatial import cKDTree
    F = cent.shape[0]
    out = np.zeros(F, bool)
    if pts.shape[0] == 0 or F == 0:
        return out
    reach = np.maximum(np.asarray(reach, np.float64), 1e-30)
    pts = np.asarray(pts, np.float64)
    rmax = float(reach.max())
    lo, hi = pts.min(0) - rmax, pts.max(0) + rmax
    cand = np.nonzero(((cent >= lo) & (cent <= hi)).all(1))[0]          # nothing outside the tile's box + largest reach can qualify
    if cand.size == 0:
        return out
    c = np.asarray(cent[cand], np.float64)
    rmin = float(reach.min())
    edges = rmin * (rmax / rmin) ** (np.arange(1, classes + 1) / float(classes)) if rmax > rmin else np.array([rmax])
    edges[-1] = rmax
    cls = np.minimum(np.searchsorted(edges, reach, side='left'), edges.size - 1)
    todo = np.ones(cand.size, bool)
    for k in range(edges.size - 1, -1, -1):                              # widest class first: it settles most candidates
        sel = cls == k
        if not sel.any() or not todo.any():
            continue
        g = max(float(edges[k]) / 4.0, 1e-30)
        if voxel is not None and voxel > 0:
            g = min(g, float(voxel))
        vox = np.unique(np.floor(pts[sel] / g).astype(np.int64), axis=0)
        centres = (vox + 0.5) * g
        bound = float(edges[k]) + 0.5 * g * np.sqrt(3.0) * (1.0 + 1e-9) + 1e-9 * (abs(float(edges[k])) + float(np.abs(c).max()))
        idx = np.nonzero(todo)[0]
        d, _ = cKDTree(centres).query(c[idx], k=1, distance_upper_bound=bound, workers=-1)
        hit = np.isfinite(d)
        todo[idx[hit]] = False
    out[cand[~todo]] = True
    return out


class HaloPartition(object):
    """Host-side decomposition of ONE mesh for the 'halo' mode, computed identically on every rank (no communication).

    For rank r: its localizations (tile r of the bisection); the faces whose centroid lies within `halo` of the tile's bounding
    box (every centroid closer than `halo` to one of its localizations is among them, so the local nearest-face query is the
    global one as long as no nearest distance exceeds `halo`); V_r = their vertices, on which the rank runs the complete
    iteration; W_r = further 1-ring neighbours of V_r, ghosts that only carry positions / normals for the curvature prior.
    A vertex is owned by the tile that contains it.  Boundary vertices = present (in V or W) on more than one rank."""

    def __init__(self, pos, nrm, nbr, faces, points, n_ranks, halo, tiles=None, detail_ranks=None, membership_ranks=None, reach=None, reach_voxel=None):
        """detail_ranks: the ranks whose share is worked out in full (index maps, local faces, local ring table); None = all.
        membership_ranks: the ranks whose MEMBERSHIP (which vertices they hold) is computed here; None = all.  Who else holds a vertex
        decides the boundary list and the peers' rows, which all ranks must agree on: a process that computes only its own membership
        leaves `holders` partial and must call set_holders() with the sum over the ranks (one all-reduce of M int64: the ranks' bits are
        disjoint) before using `boundary` / `peers`.

        reach: {rank: (n_r,) array} -- PER-LOCALIZATION radii for the ranks whose membership is computed here (in the order of
        parts[rank]): the rank then holds the faces whose centroid lies within reach_i of its localization i (faces_within_reach),
        instead of everything within `halo` of its tile's bounding box.  With reach_i = (nearest distance of i now) + margin the local
        query stays the global one while no nearest distance has grown, and no centroid has moved, by more than the margin together --
        a halo that pays for the mesh's movement, not for the height of the few localizations far above the surface."""
        pos = np.asarray(pos, np.float32)
        faces = np.asarray(faces, np.int32)
        nbr = np.asarray(nbr, np.int32)
        M = pos.shape[0]
        self.M, self.n_ranks, self.halo = M, int(n_ranks), float(halo)
        # `tiles` = bisect_tiles(points, n_ranks) computed earlier: the tiles of a cloud do not depend on the mesh
        self.parts, classify = tiles if tiles is not None else bisect_tiles(points, n_ranks)
        self.owner = classify(pos)
        cent = ((pos[faces[:, 0]] + pos[faces[:, 1]]) + pos[faces[:, 2]]) / np.float32(3.0)
        fowner = self.owner[faces]                                  # (F, 3): owners of a face's vertices
        if self.n_ranks > 62:
            raise ValueError('HaloPartition: at most 62 ranks (the holders of a vertex are kept as the bits of an int64)')
        count = np.zeros(M, np.int32)
        holders = np.zeros(M, np.int64)                              # bit r: rank r holds the vertex (as one of its own, a copy, or a ring ghost)
        self.ranks = []
        nbr_ok = nbr >= 0
        nbr_safe = np.where(nbr_ok, nbr, 0)
        boxes = getattr(classify, 'boxes', None)
        for r in range(self.n_ranks):
            if membership_ranks is not None and r not in membership_ranks:
                self.ranks.append(dict(pidx=self.parts[r]))
                continue
            if boxes is not None:
                box = boxes[r]
            else:
                p = points[self.parts[r]]
                box = (p.min(0), p.max(0)) if p.shape[0] else None
            if reach is not None and r in reach:
                fmask = faces_within_reach(cent, points[self.parts[r]], np.asarray(reach[r], np.float64), voxel=reach_voxel)
            elif box is not None:
                lo, hi = box[0] - self.halo, box[1] + self.halo
                fmask = ((cent >= lo) & (cent <= hi)).all(1)
            else:
                fmask = np.zeros(faces.shape[0], bool)
            # the faces of every owned vertex belong to the owner as well (its prior and its update must be complete there)
            fmask |= (fowner == r).any(1)
            inV = self.owner == r
            if fmask.any():
                inV[faces[fmask].ravel()] = True
            # W = 1-ring of V outside V: ghosts
            inW = np.zeros(M, bool)
            rows_ok = nbr_ok[inV]
            inW[nbr_safe[inV][rows_ok]] = True
            inW &= ~inV
            count += inV
            count += inW
            holders |= (inV | inW).astype(np.int64) << r
            if detail_ranks is not None and r not in detail_ranks:
                self.ranks.append(dict(pidx=self.parts[r]))
                continue
            fsel = np.nonzero(fmask)[0]
            gV = np.nonzero(inV)[0]
            gW = np.nonzero(inW)[0]
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
            self.ranks.append(dict(pidx=self.parts[r], gv=gv, nV=int(gV.size), faces=g2l[faces[fsel]].astype(np.int32),
                                   nbr=nbr_l, valid=valid, owned=owned))
        self.count = count
        self.holders = holders
        self.boundary = None
        if membership_ranks is None:
            self.set_holders(holders)

    def set_count(self, count):
        """count[v] = number of ranks holding vertex v (summed over the ranks) -> the boundary list and every detailed rank's rows in it"""
        self.count = np.asarray(count)
        self.boundary = np.nonzero(self.count > 1)[0]
        slot = -np.ones(self.M, np.int64)
        slot[self.boundary] = np.arange(self.boundary.size)
        for d in self.ranks:
            if 'gv' not in d:
                continue
            s = slot[d['gv']]
            d['b_local'] = np.nonzero(s >= 0)[0]
            d['b_slot'] = s[s >= 0]


    def set_holders(self, holders):
        """holders[v] = the ranks holding vertex v as bits of an int64 (a process that computed its own membership only: the SUM over the
        ranks -- the bits are disjoint).  -> the boundary list (set_count) and, for every detailed rank, the rows of the OWNER-WISE
        exchange: `peers` = (peer ranks, ghost_off, ghost_local, owned_off, owned_local) -- for peer q the local ids of this rank's copies
        of vertices q owns, and of the vertices this rank owns that q holds, both in ascending global id: rank r's ghost segment for q
        and q's owned segment for r list the same vertices in the same order (both follow from the same masks and the same owner map)."""
        holders = np.asarray(holders, np.int64)
        self.holders = holders
        count = np.zeros(self.M, np.int32)
        for r in range(self.n_ranks):
            count += ((holders >> r) & 1).astype(np.int32)
        self.set_count(count)
        for r, d in enumerate(self.ranks):
            if 'gv' not in d:
                continue
            gv = d['gv']
            owned = np.asarray(d['owned'], bool)
            own_of = self.owner[gv]
            gl = np.nonzero(~owned)[0]
            gl = gl[np.lexsort((gv[gl], own_of[gl]))]                      # by owner, ascending global id inside
            g_owner = own_of[gl]
            ol = np.nonzero(owned)[0]
            ol = ol[np.argsort(gv[ol], kind='stable')]
            others = holders[gv[ol]] & ~(np.int64(1) << r)
            peers, go, oo, gparts, oparts = [], [0], [0], [], []
            for q in range(self.n_ranks):
                if q == r:
                    continue
                gq = gl[g_owner == q]
                oq = ol[((others >> q) & 1) == 1]
                if gq.size == 0 and oq.size == 0:
                    continue
                peers.append(q)
                gparts.append(gq)
                oparts.append(oq)
                go.append(go[-1] + gq.size)
                oo.append(oo[-1] + oq.size)
            d['peers'] = (np.asarray(peers, np.int32), np.asarray(go, np.int64),
                          np.concatenate(gparts).astype(np.int32) if gparts else np.zeros(0, np.int32),
                          np.asarray(oo, np.int64), np.concatenate(oparts).astype(np.int32) if oparts else np.zeros(0, np.int32))

    @staticmethod
    def exchange_bytes(peers):
        """bytes one rank SENDS per iteration with the owner-wise exchange: its copies' partial accumulator rows (32 B) to their owners,
        and for every copy another rank holds of a vertex it owns the sum (16 B: four float32) and the new position (12 B)"""
        return int(peers[1][-1]) * 32 + int(peers[3][-1]) * 28


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


def margin_after_block(worst, drift, step, budget, cut_margin, halo, min_margin, per_point, blocks_on_shares, blocks_since_shrink):
    """New shares before the next block, and with which margin?  -> (cut, margin or None).  worst = largest growth of a nearest distance
    (or the largest nearest distance) since the shares were cut, drift = the mesh's largest movement since then, step = the drift the last
    block added, budget = what the shares were cut for.
      * cut as soon as another block like the last one (one and a half times its movement, for room) could go beyond the budget -- a fit
        slows down as it converges (C3 moves 26, 14, 9, 3, 2 ... nm per block of 5); the margin of the new shares is at least five steps
        (a fit's movement roughly halves from block to block; three were measured too few) and TWICE the margin that has just run out:
        cutting shares costs 0.1-0.5 s of host time, hundreds of blocks' worth of device time, and late in a fit growth + drift still rise
        by 1-2 nm per block -- a margin of a few steps was used up every two or three blocks (19 cuts in 60 blocks, 10 s for 0.1 s of
        iterations).  Doubling makes the cuts of a long fit few; never more than `halo`;
      * shares that have lasted a hundred blocks with room to spare are cut again with a smaller margin (fewer vertices held), never less
        than half."""
    if worst + drift + 1.5 * step > budget:
        if not per_point:
            return True, None
        return True, min(halo, max(min_margin, 5.0 * step, 2.0 * cut_margin))
    if per_point and blocks_on_shares >= 100 and blocks_since_shrink >= 100:
        want = max(min_margin, 5.0 * step, 2.0 * (worst + drift))
        if want < 0.5 * cut_margin:
            return True, max(want, 0.5 * cut_margin)
    return False, None


class HaloExceeded(RuntimeError):
    """a block of a sharded mesh ran beyond what its shares guarantee (HaloScene.search cuts new shares and runs it again, once)"""


class HaloScene(object):
    """Front end of the 'halo' mode: every rank holds the whole mesh on the HOST (it is small: 24 bytes per vertex) and its own share of
    everything on the DEVICE, and keeps it there:

      * the tiles of the cloud are cut once (they do not depend on the mesh), a rank's localizations are uploaded once;
      * the partition of the mesh (HaloPartition: shares, owners, boundary list), the rank's sub-mesh on the device and the optimiser
        over it are built once per topology -- and again when the mesh has moved too far for the margin the shares were cut with, or
        after mesh_changed() (a remesh);
      * search() = one block on the resident state: run_search(..., 'halo') and then ONE all-reduce of the owners' rows, float32 on the
        device, that gives every rank the whole new mesh (copied to the host mesh like search() does, mesh_conj_grad.py:288-289);
      * refresh_normals() = the block-boundary refresh (_membrane_mesh.pyx:1524-1527) for an unchanged topology on the device: every rank
        refreshes its share, the owners' normals of the boundary vertices go round, the optimiser history restarts.

    The sharded nearest-face query is exact as long as (growth of any nearest distance since the shares were cut) + (drift of the mesh
    since then) stays within the margin the shares were cut with (per-localization halos; with one radius for all: largest nearest
    distance + drift <= halo); checked after every block -- a block that went beyond runs again on new shares (search()), and new shares
    are cut BEFORE a block whenever another block like the last one could go beyond, with a margin of at least five times the last
    block's movement and TWICE the margin that has just run out (cutting shares costs hundreds of blocks' worth of time: the cuts of a
    long fit must be few), never more than `halo`; optimize_layout() cuts once with the small margin a short timed run needs.

    make_executor(local_mesh, local_points) -> executor (HipExecutor over a ShrinkwrapMeshConjGrad in production; the CPU tests pass
    an oracle-backed one)."""

    def __init__(self, mesh, points, dist, halo, make_executor=None, native=None, torch_stream=None, comm=None, per_point=None, min_margin=None,
                 exchange='peers', allow_recut_every_block=False):
        self.mesh, self.points, self.dist, self.halo = mesh, np.ascontiguousarray(points, np.float32), dist, float(halo)
        self.allow_recut_every_block = bool(allow_recut_every_block)
        # per_point: shares cut with PER-LOCALIZATION halos -- a rank holds every face within (nearest distance now + margin) of each of
        # its localizations instead of everything within `halo` of its tile's bounding box: the halo then pays for the mesh's movement
        # (the margin: at most `halo`, a few times the last block's movement where a caller asks for the set-up, doubled whenever it runs out),
        # not for the height of the few localizations far above the surface.  Default: on for the HIP executor.
        self.per_point = (make_executor is None) if per_point is None else bool(per_point)
        # exchange: 'peers' = the boundary rows go between the ranks that share them (owner-wise: copies' partial sums to the owner, the
        # owner's sum and new position back); 'dense' = two all-reduces per iteration over the global list of boundary vertices
        if exchange not in ('peers', 'dense'):
            raise ValueError("exchange must be 'peers' or 'dense'")
        self.exchange = exchange
        self.margin = float(halo)
        self.min_margin = float(min_margin) if min_margin is not None else 0.05 * float(halo)
        self._blocks_total = 0
        self._last_shrink = -10 ** 9
        self.comm = comm                  # NativeComm over `native`: every collective of the run goes through the library's communicator
        if comm is not None and native is None:
            raise ValueError('HaloScene(comm=...) needs the NativeContext the communicator was created on (native=...)')
        self.rank, self.world = (comm.rank, comm.world) if comm is not None else (dist.get_rank(), dist.get_world_size())
        self.drift, self.max_dist = 0.0, 0.0
        self.make_executor = make_executor
        self.native = native
        self.torch_stream = torch_stream
        self.last_partition = None
        self.ex = None
        self._tiles = None
        self._local_points = None
        self._local_arrays = {}          # id(global per-localization array) -> (the array, this rank's rows): the same object every block
        self._quantum = None
        self._host_full = None
        self._pos0_t = None
        self.host_ms = {}                # wall time of the host-side steps of the last block / set-up
        self.repartitions = 0
        self._last_step = 0.0                         # movement of the mesh over the previous block (nm)
        self._blocks_since_partition = 0

    # -- set-up (once per topology) ---------------------------------------------------------------------------------------------
    def mesh_changed(self):
        """the host mesh was edited (remesh, surgery, positions set by hand): partition and upload again before the next block.

        Cutting shares is HOST work of 0.2-0.5 s per rank at a million localizations (nearest distances + the reach classes' k-d trees)
        against ~1.5 ms for the block it serves: a caller that edits the mesh after EVERY block -- the recipe's pattern, 39 iterations
        with remesh_frequency 5, recipe_modules/surface_fitting.py:17,30 -- would spend 99 % of its time cutting.  The third edit in a row
        with at most one block in between is refused (RuntimeError) unless the scene was built with `allow_recut_every_block=True`: such
        fits run on ONE GPU (ShrinkwrapMembrane does), or as independent tiles."""
        since = self._blocks_total - getattr(self, '_last_edit_block', -10 ** 9)
        self._edits_in_a_row = (getattr(self, '_edits_in_a_row', 0) + 1) if since <= 1 else 1
        self._last_edit_block = self._blocks_total
        if self._edits_in_a_row >= 3 and not getattr(self, 'allow_recut_every_block', False):
            raise RuntimeError("HaloScene: the mesh was edited after each of the last three blocks -- re-cutting a sharded mesh's shares costs hundreds of blocks' worth of "
                               "host time (0.2-0.5 s per rank against ~1.5 ms per block); fit a mesh that is remeshed every block on one GPU, or pass allow_recut_every_block=True")
        self.last_partition = None

    def _hip_executor(self, local_mesh, local_points):
        from .mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext
        if self.native is None:
            import torch
            self.native = NativeContext(torch.cuda.current_device(), self.torch_stream.cuda_stream if self.torch_stream is not None else None)
        self.native.mesh_key = None
        cg = ShrinkwrapMeshConjGrad(local_mesh, local_points, native=self.native)       # (the localizations stay resident: same array object)
        prof = getattr(self, '_profiling', None)
        if prof is not None:
            cg.set_profiling(prof)
        ex = HipExecutor(cg)
        ex.write_back = False
        ex.comm = self.comm
        return ex

    def _stream(self):
        import contextlib
        if self.torch_stream is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.torch_stream)

    def _setup(self):
        import time
        t0 = time.perf_counter()
        mesh = self.mesh
        if self._tiles is None:
            # (cuts across the axis on which the fewest localizations lie near the plane: the shortest cut through the structure -- 8 ranks
            # of the genus-2 network hold 1.22 x the mesh instead of 1.30 with cuts across the longest axis; a vesicle's are the same)
            self._tiles = bisect_tiles(self.points, self.world, thin_cuts=THIN_CUTS_NM)
            self._local_points = np.ascontiguousarray(self.points[self._tiles[0][self.rank]])
        if hasattr(mesh, 'neighbor_vertex_table'):
            nbr = mesh.neighbor_vertex_table()
        else:
            nbr = mesh._halfedges['vertex'][mesh._vertices['neighbors']]
            nbr[mesh._vertices['neighbors'] == -1] = -1
        pos = np.ascontiguousarray(mesh._vertices['position'], np.float32)
        nrm = np.ascontiguousarray(mesh.vertex_normals, np.float32)
        t1 = time.perf_counter()
        # every rank works out its OWN share only; who else holds a vertex (boundary list, peers' rows) comes from one all-reduce of the holder bits
        reach, d0 = None, None
        if self.per_point:
            from scipy.spatial import cKDTree
            f = np.asarray(mesh.faces)
            cent = ((pos[f[:, 0]] + pos[f[:, 1]]) + pos[f[:, 2]]) / np.float32(3.0)
            d0 = None
            cg_old = getattr(self.ex, 'cg', None) if self.ex is not None else None
            if cg_old is not None and getattr(cg_old, 'loopcount', 0) > 0 and cg_old._points_f32.shape[0] == self._local_points.shape[0]:
                # the distances of the last query on the device (one update old: ANY reference radius is valid -- the growth is measured
                # against the same numbers the halos were cut with)
                try:
                    d0 = np.ascontiguousarray(cg_old.d[:, 0], np.float32)
                except Exception:
                    d0 = None
            if d0 is None:
                d0 = cKDTree(cent).query(self._local_points, workers=-1)[0].astype(np.float32)      # this rank's nearest distances now
            reach = {self.rank: d0.astype(np.float64) * (1.0 + 1e-6) + self.margin}
        part = HaloPartition(pos, nrm, nbr, mesh.faces, self.points, self.world, self.halo, tiles=self._tiles, detail_ranks=(self.rank,),
                             membership_ranks=(self.rank,), reach=reach, reach_voxel=0.25 * self.margin)      # (a quarter: the voxels' slack, 0.87 edges, widens every halo -- half a margin held 1.5 % more of the mesh at no saving of time)
        # who else holds a vertex: the ranks' holder bits (disjoint) summed
        if self.comm is not None:
            part.set_holders(self.comm.all_reduce_host(np.ascontiguousarray(part.holders, np.int64)))
        else:
            import torch
            hold = torch.from_numpy(part.holders)
            if self.make_executor is None:
                hold = hold.cuda()
            self.dist.all_reduce(hold)
            part.set_holders(hold.cpu().numpy())
        t2 = time.perf_counter()
        self.last_partition = part
        d = part.ranks[self.rank]
        gv = d['gv']
        self._gv = gv
        self._local_mesh = ArrayMesh(pos[gv], nrm[gv], d['nbr'], d['faces'], d['valid'])
        old = self.ex
        keep = {k: getattr(old, k) for k in ('collective_timer', '_prenorm_cache') if old is not None and hasattr(old, k)}
        self.ex = (self.make_executor or self._hip_executor)(self._local_mesh, self._local_points)
        for k, v in keep.items():
            setattr(self.ex, k, v)
        if old is not None and hasattr(old, 'cg') and hasattr(self.ex, 'cg') and hasattr(old.cg, 'stage_ms_total') and getattr(self, '_profiling', None):
            self.ex.cg.stage_ms_total = old.cg.stage_ms_total          # HIP-event totals run on across a re-partition (bench.py reads them at the end)
        if self.exchange == 'peers':
            self.ex.set_boundary(d['b_local'], d['b_slot'], part.boundary.size, d['owned'], gv, part.M, peers=d['peers'])
        else:
            self.ex.set_boundary(d['b_local'], d['b_slot'], part.boundary.size, d['owned'], gv, part.M)
        self.exchange_bytes = HaloPartition.exchange_bytes(d['peers']) if self.exchange == 'peers' else int(part.boundary.size) * 44
        self.boundary_vertices = int(part.boundary.size)
        self._pos0 = pos.copy()                       # where the mesh was when the shares were cut (drift budget of the halo)
        if hasattr(self.ex, 'set_reference'):
            if self.per_point:
                self.ex.set_reference(self._pos0, d0)
            else:
                self.ex.set_reference(self._pos0)
        self._cut_margin = self.margin
        self._blocks_since_partition = 0
        self._pos0_t = None
        self._valid = mesh._vertices['halfedge'] != -1
        self._all_valid = bool(self._valid.all())
        self._valid_u8 = np.ascontiguousarray(self._valid, np.uint8)
        self.repartitions += 1
        t3 = time.perf_counter()
        self.host_ms['setup'] = (t3 - t0) * 1e3
        self.host_ms['setup_parts'] = {'tiles_ring_table_normals': (t1 - t0) * 1e3, 'partition_and_count_all_reduce': (t2 - t1) * 1e3, 'sub_mesh_optimiser_boundary': (t3 - t2) * 1e3}

    def _local(self, a):
        """this rank's rows of a per-localization (3N,) array -- the SAME object for the same input, so that the residency keys of the
        optimiser (and the weight normalisation cache of run_search) hold from block to block"""
        if a is None or np.isscalar(a):
            return a
        hit = self._local_arrays.get(id(a))
        if hit is None or hit[0] is not a:
            pidx = self._tiles[0][self.rank]
            hit = (a, np.ascontiguousarray(np.asarray(a, np.float32).reshape(-1, 3)[pidx].ravel()))
            if len(self._local_arrays) >= 4:
                self._local_arrays.clear()
            self._local_arrays[id(a)] = hit
        return hit[1]

    def _to_host(self, t):
        """(3M,) float32 tensor -> (M,3) array on the host; a device tensor goes through one pinned buffer, the copy is only ENQUEUED
        (the caller synchronises once for everything it reads back)"""
        if getattr(t, 'is_cuda', False):
            import torch
            if self._host_full is None or self._host_full.numel() != t.numel():
                self._host_full = torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
            self._host_full.copy_(t, non_blocking=True)
            return self._host_full.numpy().reshape(-1, 3)
        return t.numpy().reshape(-1, 3)

    # -- one block ----------------------------------------------------------------------------------------------------------------
    def search(self, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True):
        """One block.  The sharded query is only known to have been exact AFTER the block (the growth of the nearest distances and the
        drift of the mesh against the margin the shares were cut with): if it was not, the host mesh still holds the positions of the
        block's start -- new shares are cut from it with the whole `halo` as margin and the block runs once more; only a block that fails
        on fresh shares raises."""
        try:
            return self._search_once(lams, num_iters, sigma_inv, weights, pos, last_step)
        except HaloExceeded:
            if self._blocks_since_partition == 0 and self._cut_margin >= self.halo:
                raise
            self.margin = self.halo
            self.last_partition = None
            self.redone_blocks = getattr(self, 'redone_blocks', 0) + 1
            return self._search_once(lams, num_iters, sigma_inv, weights, pos, last_step)

    def _search_once(self, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True):
        import time
        if self.last_partition is None:
            self._setup()
        ex, mesh = self.ex, self.mesh
        if type(lams) is float or np.isscalar(lams):
            lams = [float(lams)]
        if self.comm is not None:
            # the library's communicator and stream: the block, then the owners' rows of the whole mesh and the block's statistics go round
            run_search(ex, None, 'halo', self._local_points, lams, num_iters, self._local(sigma_inv), self._local(weights), pos, last_step,
                       quantum=self._quantum)
            # (the block's tail -- the owners' rows of the whole mesh and the block's statistics going round, both staged in pinned host
            # memory -- ran inside nw_search, behind the last iteration: what is left is host work)
            t0 = time.perf_counter()
            L, h, chk = ex.L, ex.h, ex.native.check
            st4 = np.zeros(4, np.float32)
            chk(L.nw_get(h, nw.NW_ARR_HALO_STATS, nw.ptr(st4), st4.nbytes))
            worst, q, d2 = float(st4[0]), float(st4[1]), float(st4[2])
            newpos = None                                     # (the staged mesh: copied out by the library's host threads below)
            return self._finish_block(ex, mesh, newpos, worst, q, d2, t0)
        import torch
        with self._stream():
            run_search(ex, self.dist, 'halo', self._local_points, lams, num_iters, self._local(sigma_inv), self._local(weights), pos, last_step,
                       quantum=self._quantum)
            t0 = time.perf_counter()
            # ONE all-reduce of the owners' rows (float32, on the device) gives every rank the whole new mesh
            full = ex.gather_owned('pos')
            self.dist.all_reduce(full)
            # exactness of the sharded query, drift of the mesh and the common quantum of the NEXT block: one small MAX all-reduce per block
            if hasattr(ex, 'block_stats'):
                stats = ex.block_stats(ex.max_dist)           # one launch of the library (no torch arithmetic, no allocation)
            else:
                if self._pos0_t is None or self._pos0_t.device != full.device:
                    self._pos0_t = torch.from_numpy(self._pos0.ravel()).to(full.device)
                drift2 = (full - self._pos0_t).view(-1, 3).pow(2).sum(1).max()
                stats = ex.new_tensor([float(ex.max_dist), ex.local_quantum() if hasattr(ex, 'local_quantum') else 0.0, 0.0])
                stats[2] = drift2
            self.dist.all_reduce(stats, op=self.dist.ReduceOp.MAX)
            newpos = self._to_host(full)
            worst, q, d2 = stats.tolist()[:3]             # (the one synchronisation of the block's tail: the whole mesh has landed too)
        return self._finish_block(ex, mesh, newpos, worst, q, d2, t0)

    def _finish_block(self, ex, mesh, newpos, worst, q, d2, t0):
        import time
        t1 = time.perf_counter()
        self._quantum = q if q > 0 else None
        drift = float(np.sqrt(max(d2, 0.0)))
        budget = self._cut_margin if self.per_point else self.halo
        if worst + drift > budget:
            raise HaloExceeded("halo mode: %s %.3g and the mesh has moved %.3g since the shares were cut, beyond the %s %.3g: the sharded query "
                               "is not guaranteed exact (increase `halo`)" % ("a nearest distance has grown by" if self.per_point else "a localization's nearest face centroid is at",
                                                                               worst, drift, "margin" if self.per_point else "halo radius", budget))
        posv = mesh._vertices['position']
        n_rows = ex.n_global if newpos is None else newpos.shape[0]
        # the (M,3) result: a fresh 2.4 MB array costs ~600 page faults per block -- two buffers take turns (the caller holds at most the
        # previous block's result when the next one is written; keep a copy of anything older)
        pool = getattr(self, '_out_pool', None)
        if pool is None or pool[0].shape[0] != n_rows:
            pool = self._out_pool = [np.empty((n_rows, 3), np.float32) for _ in range(2)]
            self._out_turn = 0
        self._out_turn ^= 1
        out = pool[self._out_turn]
        if newpos is None:
            direct = posv.dtype == np.float32 and posv.strides[1] == 4 and posv.strides[0] >= 12
            ex.native.check(ex.L.nw_host_copy_rows(ex.h, None, n_rows, nw.ptr(out), ctypes.c_void_p(posv.ctypes.data) if direct else None, posv.strides[0] if direct else 0,
                                                   None if self._all_valid else nw.ptr(self._valid_u8)))
            if not direct:
                posv[self._valid] = out[self._valid]
        elif hasattr(ex, 'host_copy_rows') and posv.dtype == np.float32 and posv.strides[1] == 4 and posv.strides[0] >= 12:
            ex.host_copy_rows(newpos, out, posv, None if self._all_valid else self._valid_u8)      # the library's copy threads
        else:
            out[:] = newpos
            if self._all_valid:
                posv[:] = newpos
            else:
                posv[self._valid] = newpos[self._valid]
        # New shares before the next block?  The query of a block is exact if (largest nearest distance + drift since the shares were
        # cut) stays within the halo radius THROUGH the block, which is only known afterwards: so cut again as soon as another block
        # like the last one (one and a half times its movement, for margin) could exceed it.
        step = max(drift - (self.drift if self._blocks_since_partition > 0 else 0.0), 0.0)
        self._last_step = step
        self._blocks_since_partition += 1
        self.max_dist, self.drift = worst, drift
        self._blocks_total += 1
        cut, margin = margin_after_block(worst, drift, step, budget, self._cut_margin, self.halo, self.min_margin, self.per_point,
                                         self._blocks_since_partition, self._blocks_total - self._last_shrink)
        if cut:
            self.last_partition = None                # cut new shares around the moved mesh before the next block
            if margin is not None:
                if margin < self._cut_margin:
                    self._last_shrink = self._blocks_total
                self.margin = margin
        t2 = time.perf_counter()
        self.host_ms['block_tail_collectives_and_copy'] = (t1 - t0) * 1e3
        self.host_ms['block_tail_host_mesh'] = (t2 - t1) * 1e3
        return out

    def _wanted_margin(self):
        """margin for shares cut now: a fit's movement roughly halves from block to block, so everything still to come is about one more
        step of the last block's size; five of them leave room for the nearest distances that grow meanwhile and for the re-cut rule
        (growth + drift + 1.5 steps must stay within the margin THROUGH the next block).  Three were measured too few: new shares after
        every block of the fit's first twenty iterations, each followed by a cold query."""
        return min(self.halo, max(self.min_margin, 5.0 * self._last_step))

    def optimize_layout(self):
        """One-off set-up a caller can take out of a timed region (bench.py, after its warm-up): shares cut NOW with the margin the fit
        needs from here on (the first shares were cut with the whole `halo`, for a mesh that still moves by tens of nm per block), then the
        library's own set-up (projection sort, cell tuner, work list)."""
        if self.per_point and self._blocks_since_partition > 0 and not getattr(self, '_layout_cut', False):
            # (also when the margin stays what it was: the drift and the growth the margin pays for are counted from the cut)
            self.margin = self._wanted_margin()
            self.last_partition = None
            self._layout_cut = True                 # (once: a second call -- after the new shares' first, cold, block -- is for the library's own set-up)
        if self.last_partition is None:
            self._setup()
        if hasattr(self.ex, 'cg'):
            self.ex.cg.optimize_layout()

    def refresh_normals(self, to_host=True):
        """vertex normals of the next block from the current positions, topology unchanged; restarts the optimiser's history (the
        reference builds a new optimiser per block)"""
        if self.last_partition is None:               # the shares are about to be cut again from the host mesh: refresh it there
            self.mesh.update_geometry()
            return
        ex = self.ex
        if self.comm is not None:
            p = self.mesh._vertices['position']
            ex.refresh_normals_local(float((p.max(0).astype(np.float64) - p.min(0).astype(np.float64)).max()))
            if getattr(ex, 'peers', None) is not None:
                self.comm.exchange(3, np.float32, owned_out=True)      # the owners' normals to the copies
            elif ex.n_boundary > 0:
                self.comm.all_reduce_device(nw.NW_ARR_HALO_ROWS, 3 * ex.n_boundary, np.float32)
            ex.take_normals()
            if to_host:
                ex.native.check(ex.L.nw_halo_gather_owned(ex.h, nw.NW_ARR_NRM))
                self.comm.all_reduce_device(nw.NW_ARR_HALO_FULL, 3 * ex.n_global, np.float32)
                nrm = np.empty((ex.n_global, 3), np.float32)
                ex.native.check(ex.L.nw_get(ex.h, nw.NW_ARR_HALO_FULL, nw.ptr(nrm), nrm.nbytes))
                self.mesh._vertices['normal'][:] = nrm
            return
        with self._stream():
            if hasattr(ex, 'refresh_normals_local'):
                p = self.mesh._vertices['position']
                ext = float((p.max(0).astype(np.float64) - p.min(0).astype(np.float64)).max())
                ex.refresh_normals_local(ext)
                if getattr(ex, 'peers', None) is not None:
                    peer_exchange(self.dist, ex, 'rows_to_copies')
                elif ex.n_boundary > 0:
                    self.dist.all_reduce(ex.boundary_rows())
                ex.take_normals()
                if to_host:
                    t = ex.gather_owned('nrm')
                    self.dist.all_reduce(t)
                    nrm = self._to_host(t)
                    if getattr(t, 'is_cuda', False):
                        import torch
                        torch.cuda.current_stream().synchronize()
                    self.mesh._vertices['normal'][:] = nrm
            else:                                     # executor without a device refresh (the CPU tests'): host substrate's definition
                self.mesh.update_geometry()
                ex.set_normals(np.ascontiguousarray(np.asarray(self.mesh.vertex_normals, np.float32)[self._gv]))

    def set_profiling(self, level):
        self._profiling = level
        if self.ex is not None and hasattr(self.ex, 'cg'):
            self.ex.cg.set_profiling(level)
