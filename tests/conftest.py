import os
import sys
import subprocess
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'reference: needs /root/reference (build container only)')
    # the oracle's C library is test infrastructure: build it on demand (gcc only, < 1 s)
    lib = os.path.join(ROOT, 'oracle', 'libnw_oracle.so')
    src = os.path.join(ROOT, 'oracle', 'nw_oracle.c')
    if (not os.path.exists(lib)) or os.path.getmtime(lib) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), 'libnw_oracle.so'])
    # the host-side remesher / half-edge helper is product code but plain C++: make sure a fresh checkout has it before the first TriMesh
    from ch_shrinkwrap_amd import build as _b
    _b.build_host_library()


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


@pytest.fixture(scope='session')
def golden():
    return load_golden


def rel_rms(a, b):
    """vertex RMS difference relative to the bounding-box diagonal of `b`."""
    a = np.asarray(a, 'f8').reshape(-1, 3)
    b = np.asarray(b, 'f8').reshape(-1, 3)
    diag = np.linalg.norm(b.max(0) - b.min(0))
    return float(np.sqrt(((a - b) ** 2).sum(1).mean()) / diag)
