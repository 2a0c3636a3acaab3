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
        cg._consume_logs(logs, lc.value)
        cg._accumulate_stage_ms()
        cg._finish()
        return cg.fs


def run_search(ex, dist, mode, data, lams, num_iters, sigma_inv, weights=None, pos=False, last_step=True):
    """One search() call of `num_iters` iterations over all ranks of `dist` (a torch.distributed-like module with an
    initialised default group).  Every rank calls this collectively with its own executor."""
    if mode not in ('tiles', 'replicated'):
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
    n_red = ex.n_scalars if mode == 'tiles' else ex.n_point_scalars
    for _ in range(int(num_iters)):
        ex.attract()
        if mode == 'replicated':
            dist.all_reduce(ex.vertex_accumulator())
        ex.directions()
        dist.all_reduce(ex.scalars(n_red))
        ex.update()
    return ex.end()


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
