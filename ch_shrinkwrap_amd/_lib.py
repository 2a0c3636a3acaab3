"""
ctypes binding of include/nanowrap.h (libnanowrap_hip.so).  There is NO CPU fallback: if the HIP library is
missing or no GPU is visible, construction of the optimiser raises.
"""
import os
import ctypes
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('NW_LIB_PATH') or os.path.join(HERE, 'libnanowrap_hip.so')      # (NW_LIB_PATH: developer knob, A/B of two builds on one box)

NW_OK = 0
NW_ERR_BADARG, NW_ERR_HIP, NW_ERR_NAN, NW_ERR_SINGULAR, NW_ERR_NONFINITE, NW_ERR_NOMEM, NW_ERR_INTERNAL, NW_ERR_REMOTE, NW_ERR_HANDOFF, NW_ERR_NONMANIFOLD = -1, -2, -3, -4, -5, -6, -7, -8, -9, -10
NW_WEIGHTS_FROM_SIGMA_INV, NW_WEIGHTS_SCALAR, NW_WEIGHTS_ARRAY, NW_WEIGHTS_PRENORMALIZED = 0, 1, 2, 3
NW_FLAG_POSITIVITY, NW_FLAG_NO_LAST_STEP, NW_FLAG_WFUNC, NW_FLAG_RESULT_TO_HOST = 1, 2, 4, 8
NW_FLAG_COMM_TILES, NW_FLAG_COMM_REPLICATED, NW_FLAG_COMM_HALO = 16, 32, 64
NW_FLAG_ROWS_ASYNC = 128
(NW_ARR_S, NW_ARR_RES, NW_ARR_VIDX, NW_ARR_W, NW_ARR_DIST, NW_ARR_FACE, NW_ARR_POS, NW_ARR_FDEF, NW_ARR_PI,
 NW_ARR_MESHPOS, NW_ARR_VACC, NW_ARR_SCALARS, NW_ARR_NBR, NW_ARR_NRM, NW_ARR_VALID, NW_ARR_HALO_ACC, NW_ARR_HALO_ROWS,
 NW_ARR_HALO_FULL, NW_ARR_HALO_STATS, NW_ARR_PEER_SEND, NW_ARR_PEER_RECV) = range(21)
NW_N_SCALARS = 32

# every symbol include/nanowrap.h declares (tests/test_abi.py checks the exports against the header)
NW_INFO_POINT_SCALARS, NW_INFO_SCALARS, NW_INFO_SCALAR_STRIDE = 0, 1, 2
SYMBOLS = ['nw_abi_version', 'nw_create', 'nw_destroy', 'nw_last_error', 'nw_set_stream', 'nw_synchronize',
           'nw_set_points', 'nw_set_mesh', 'nw_refresh_normals', 'nw_reset_history', 'nw_search', 'nw_search_begin',
           'nw_iter_attract', 'nw_iter_directions', 'nw_iter_update', 'nw_search_end', 'nw_info',
           'nw_apply_A', 'nw_apply_At', 'nw_get', 'nw_write_back', 'nw_device_ptr', 'nw_lfunc', 'nw_curvature', 'nw_set_profiling', 'nw_stage_ms', 'nw_debug', 'nw_set_data', 'nw_accumulator_quantum', 'nw_optimize_layout', 'nw_set_write_back',
           'nw_set_boundary', 'nw_halo_rows', 'nw_halo_gather_owned', 'nw_host_copy_rows',
           'nw_comm_unique_id', 'nw_comm_init', 'nw_comm_all_reduce', 'nw_halo_set_reference', 'nw_halo_block_stats',
           'nw_remesh_device', 'nw_host_free']


class IterLog(ctypes.Structure):
    _fields_ = [('test', ctypes.c_double), ('res_norm', ctypes.c_double), ('prefs_norm', ctypes.c_double),
                ('cpred', ctypes.c_double), ('wpred', ctypes.c_double), ('c', ctypes.c_double * 3),
                ('H', ctypes.c_double * 9), ('G', ctypes.c_double * 3), ('mean_dist', ctypes.c_double), ('max_dist', ctypes.c_double),
                ('n_search', ctypes.c_int32), ('nn_max_ring', ctypes.c_int32), ('status', ctypes.c_int32),
                ('executed', ctypes.c_int32)]


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('%s not found: build it with `python -m ch_shrinkwrap_amd.build` (hipcc, gfx950). '
                           'There is no CPU fallback for the NanoWrap hot path.' % LIB_PATH)
    # PyTorch brings its own copy of the HIP runtime (torch/lib/libamdhip64.so), this library is linked against the system's: one process gets
    # whichever is loaded FIRST for both.  With the system's first, torch later finds "No HIP GPUs" (seen when a test touched the library
    # before torch.cuda); with torch's first everything works -- which is the order bench.py and the multi-GPU layer have anyway.  So:
    # torch's copy first, whenever torch is installed (it is this package's plumbing for streams and torch.distributed) -- without importing
    # torch, which takes a second: both copies have the SONAME libamdhip64.so.7, so loading torch's by path makes it the process's.
    import sys
    if 'torch' not in sys.modules:
        try:
            import importlib.util
            spec = importlib.util.find_spec('torch')
            hip = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so') if spec and spec.origin else None
            if hip and os.path.exists(hip):
                ctypes.CDLL(hip, mode=ctypes.RTLD_GLOBAL)
        except (ImportError, OSError, ValueError):
            pass
    L = ctypes.CDLL(LIB_PATH)
    vp, i32, i64, f32, u32 = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32
    L.nw_abi_version.argtypes = []
    L.nw_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.nw_destroy.argtypes = [vp]
    L.nw_destroy.restype = None
    L.nw_last_error.argtypes = [vp]
    L.nw_last_error.restype = ctypes.c_char_p
    L.nw_set_stream.argtypes = [vp, vp]
    L.nw_synchronize.argtypes = [vp]
    L.nw_set_points.argtypes = [vp, vp, i64, vp, f32, i32, vp, f32]
    L.nw_set_mesh.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32]
    L.nw_refresh_normals.argtypes = [vp, vp, ctypes.c_double]
    L.nw_reset_history.argtypes = [vp]
    L.nw_search.argtypes = [vp, vp, i32, i32, u32, vp, ctypes.POINTER(IterLog), ctypes.POINTER(i32)]
    L.nw_search_begin.argtypes = [vp, vp, i32, i32, u32]
    L.nw_iter_attract.argtypes = [vp]
    L.nw_iter_directions.argtypes = [vp]
    L.nw_iter_update.argtypes = [vp]
    L.nw_search_end.argtypes = [vp, vp, ctypes.POINTER(IterLog), ctypes.POINTER(i32)]
    L.nw_info.argtypes = [i32]
    L.nw_apply_A.argtypes = [vp, vp, vp]
    L.nw_apply_At.argtypes = [vp, vp, vp]
    L.nw_get.argtypes = [vp, i32, vp, i64]
    L.nw_write_back.argtypes = [vp, vp, vp, i64]
    L.nw_set_write_back.argtypes = [vp, vp, i64]
    L.nw_set_boundary.argtypes = [vp, vp, vp, i64, i64, vp, vp, i64, ctypes.c_int32, vp, vp, vp, vp, vp]
    L.nw_halo_rows.argtypes = [vp, i32, i32]
    L.nw_halo_gather_owned.argtypes = [vp, i32]
    L.nw_host_copy_rows.argtypes = [vp, vp, i64, vp, vp, i64, vp]
    L.nw_comm_unique_id.argtypes = [vp, i64]
    L.nw_comm_init.argtypes = [vp, vp, i64, i32, i32]
    L.nw_comm_all_reduce.argtypes = [vp, vp, i64, i32, i32]
    L.nw_halo_set_reference.argtypes = [vp, vp, vp, i64]
    L.nw_halo_block_stats.argtypes = [vp, ctypes.c_double]
    L.nw_set_data.argtypes = [vp, vp]
    L.nw_device_ptr.argtypes = [vp, i32, ctypes.POINTER(vp), ctypes.POINTER(i64)]
    L.nw_lfunc.argtypes = [vp, i32, vp, vp, vp]
    L.nw_curvature.argtypes = [vp, vp, vp, vp, f32, f32, f32, f32] + [vp] * 12
    L.nw_set_profiling.argtypes = [vp, i32]
    L.nw_stage_ms.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(i64)]
    L.nw_debug.argtypes = [vp, i32, vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_int)]
    L.nw_optimize_layout.argtypes = [vp]
    L.nw_accumulator_quantum.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    for s in SYMBOLS:
        if s not in ('nw_destroy', 'nw_last_error', 'nw_host_free'):
            getattr(L, s).restype = i32
    if L.nw_abi_version() != 6:
        raise RuntimeError('libnanowrap_hip.so ABI version mismatch')
    _lib = L
    return L


_hip = None


def pinned_empty(shape, dtype):
    """ndarray over page-locked host memory (hipHostMalloc): a device-to-host copy into it is a plain DMA, not a staged one.  The memory lives
    as long as the process (block-sized buffers allocated once per scene).  Falls back to ordinary memory if the HIP runtime cannot be loaded."""
    global _hip
    n = int(np.prod(shape)) * np.dtype(dtype).itemsize
    try:
        if _hip is None:
            _hip = ctypes.CDLL('libamdhip64.so')
            _hip.hipHostMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
            _hip.hipHostMalloc.restype = ctypes.c_int
        p = ctypes.c_void_p()
        if _hip.hipHostMalloc(ctypes.byref(p), max(n, 1), 0) != 0 or not p.value:
            raise OSError('hipHostMalloc failed')
        buf = (ctypes.c_char * max(n, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    except Exception:
        return np.empty(shape, dtype)


def ptr(a):
    """host ndarray (C-contiguous) or raw integer device pointer -> c_void_p"""
    if a is None:
        return None
    if isinstance(a, (int, np.integer)):
        return ctypes.c_void_p(int(a))
    return a.ctypes.data_as(ctypes.c_void_p)


class NanoWrapError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, 'nanowrap status %d: %s' % (code, msg))
        self.code = code


def check(L, ctx, code):
    """Map a status to the exception the reference would have raised at the same place."""
    if code == NW_OK:
        return
    msg = L.nw_last_error(ctx)
    msg = msg.decode() if msg else ''
    if code == NW_ERR_NAN:
        raise AssertionError(msg)                       # mesh_conj_grad.py:514,548,580 are `assert`s
    if code == NW_ERR_SINGULAR:
        raise np.linalg.LinAlgError(msg)                # numpy.linalg.solve, conj_grad.py:219
    if code == NW_ERR_BADARG:
        raise ValueError(msg)
    if code == NW_ERR_REMOTE:
        msg = msg or 'another rank raised a status in this iteration (its own exception names it); this rank stopped with it'
    raise NanoWrapError(code, msg)
