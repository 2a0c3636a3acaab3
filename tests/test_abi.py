"""CPU test: libnanowrap_hip.so loads and exports every function include/nanowrap.h declares (no GPU calls)."""
import os
import re
import ctypes

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, 'include', 'nanowrap.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(nw_[a-zA-Z0-9_]+)\s*\(', txt)))


def test_library_exports_header_symbols():
    from ch_shrinkwrap_amd import build, _lib
    build.build_hip_library()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), 'libnanowrap_hip.so does not export %s' % n
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.load().nw_abi_version() == 6
    assert _lib.load().nw_info(_lib.NW_INFO_POINT_SCALARS) == 14          # 13 point-side sums + the status slot that travels with them
    assert len(names) <= 42


def test_log_struct_layout_matches_header():
    from ch_shrinkwrap_amd import _lib
    # 5 + 3 + 9 + 3 + 2 doubles (mean_dist, max_dist), 4 int32
    assert ctypes.sizeof(_lib.IterLog) == 22 * 8 + 4 * 4


def test_product_path_never_imports_the_oracle():
    """The shipped package must not reference oracle/ (it is test infrastructure)."""
    pkg = os.path.join(ROOT, 'ch_shrinkwrap_amd')
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, fn)).read()
                assert 'nanowrap_oracle' not in src and 'from oracle' not in src and 'import oracle' not in src, fn


def test_no_gpu_means_loud_failure():
    import numpy as np
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    v, f = icosphere(1, 10.0)
    with pytest.raises(RuntimeError):
        ShrinkwrapMeshConjGrad(TriMesh(v, f), np.zeros((10, 3), 'f4'))


def test_kernels_stay_within_their_resource_budget():
    """The per-iteration kernels are tuned to an occupancy that only the register allocator enforces: no scratch, no spilled VGPRs,
    VGPRs and LDS within ch_shrinkwrap_amd/build.py's KERNEL_BUDGETS (read from the gfx950 code object's notes; build() fails likewise)."""
    from ch_shrinkwrap_amd import build
    build.build_hip_library()
    res = build.check_kernel_budgets()                  # raises on a violation
    nn = res['k_nn_wave<false>']
    assert nn['vgpr'] <= 80 and nn['scratch'] == 0 and nn['vgpr_spill'] == 0 and nn['sgpr_spill'] == 0
    for k in build.KERNEL_BUDGETS:
        assert res[k]['scratch'] == 0, k
    # a kernel over its budget is caught (budget lowered by one register for the check)
    worst = dict(build.KERNEL_BUDGETS)
    try:
        build.KERNEL_BUDGETS['k_nn_wave<false>'] = (res['k_nn_wave<false>']['vgpr'] - 1, 16 * 1024)
        with pytest.raises(RuntimeError):
            build.check_kernel_budgets()
    finally:
        build.KERNEL_BUDGETS.clear(); build.KERNEL_BUDGETS.update(worst)


def test_device_remesher_checks_its_arguments_before_it_touches_a_gpu():
    """nw_remesh_device (ABI 6): sizes, NULL pointers and a non-positive target are refused with NW_ERR_BADARG before any HIP call -- also
    where there is no GPU; with valid arguments and no GPU the call fails loudly (NW_ERR_HIP), it does not fall back to the host remesher."""
    import numpy as np
    from ch_shrinkwrap_amd import _lib
    L = _lib.load()
    v = np.zeros((4, 3), np.float32)
    f = np.array([[0, 1, 2], [0, 3, 1], [0, 2, 3], [1, 3, 2]], np.int32)
    ov, of, nv, nf = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
    L.nw_remesh_device.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    out = (ctypes.byref(ov), ctypes.byref(nv), ctypes.byref(of), ctypes.byref(nf), None)
    assert L.nw_remesh_device(0, None, 4, f.ctypes.data, 4, 5, 1.0, 0.5, 0, 16, *out) == _lib.NW_ERR_BADARG
    assert L.nw_remesh_device(0, v.ctypes.data, 2, f.ctypes.data, 4, 5, 1.0, 0.5, 0, 16, *out) == _lib.NW_ERR_BADARG      # fewer than three vertices
    assert L.nw_remesh_device(0, v.ctypes.data, 4, f.ctypes.data, 0, 5, 1.0, 0.5, 0, 16, *out) == _lib.NW_ERR_BADARG      # no face
    assert L.nw_remesh_device(0, v.ctypes.data, 4, f.ctypes.data, 4, 5, -1.0, 0.5, 0, 16, *out) == _lib.NW_ERR_BADARG     # the C-ABI wants an explicit target
    assert L.nw_remesh_device(0, v.ctypes.data, 4, f.ctypes.data, 4, -1, 1.0, 0.5, 0, 16, *out) == _lib.NW_ERR_BADARG
    assert L.nw_remesh_device(0, v.ctypes.data, 4, f.ctypes.data, 4, 5, 1.0, 0.5, -3, 16, *out) == _lib.NW_ERR_BADARG
    assert ov.value is None and of.value is None
    import torch
    if not torch.cuda.is_available():
        assert L.nw_remesh_device(0, v.ctypes.data, 4, f.ctypes.data, 4, 5, 1.0, 0.5, 0, 16, *out) == _lib.NW_ERR_HIP
