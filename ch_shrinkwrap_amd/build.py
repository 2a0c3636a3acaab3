"""
Builds the native pieces in-tree (no pip, no JIT cache):
  * ch_shrinkwrap_amd/libnanowrap_hip.so  -- the HIP kernels + C-ABI (hipcc, --offload-arch=gfx950)
  * ch_shrinkwrap_amd/libnw_remesh.so     -- the block-boundary remesher (host C++, g++; include/nw_remesh.h)
The oracle (test infrastructure) is built by oracle/Makefile, see __graft_entry__.build().
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libnanowrap_hip.so')
SRC = os.path.join(HERE, 'csrc', 'nanowrap.hip')
SRC_SORT = os.path.join(HERE, 'csrc', 'nw_sort.hip')      # set-up radix sort (hipCUB), its own translation unit
OBJ_SORT = os.path.join(HERE, 'csrc', 'nw_sort.o')
OBJ_MAIN = os.path.join(HERE, 'csrc', 'nanowrap.o')
import glob
# every header of csrc/ is included by nanowrap.hip (directly or through nw_kernels.h): editing any of them must rebuild the library
DEPS = [SRC, SRC_SORT] + sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'nanowrap.h')]

# -ffp-contract=off : the parity-critical float32 arithmetic must round products before adding, exactly like
#                     the NumPy reference (explicit fma() is used where contraction is wanted);
# -munsafe-fp-atomics: float/double atomicAdd -> global_atomic_add_f32/f64 (no CAS loop).
HIPCC_FLAGS = ['-O3', '--offload-arch=gfx950', '-fPIC', '-shared', '-ffp-contract=off', '-munsafe-fp-atomics',
               '-fvisibility=hidden', '-Wall', '-Wno-unused-function']


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_hip_library(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

    def run(cmd):
        if verbose:
            print(' '.join(cmd))
        subprocess.check_call(cmd)

    if force or not os.path.exists(OBJ_SORT) or os.path.getmtime(OBJ_SORT) < os.path.getmtime(SRC_SORT):
        run([hipcc, '-O3', '--offload-arch=gfx950', '-fPIC', '-fvisibility=hidden', '-Wno-unused-value', '-c', '-o', OBJ_SORT, SRC_SORT])
    run([hipcc] + [f for f in HIPCC_FLAGS if f != '-shared'] + ['-c', '-o', OBJ_MAIN, SRC])
    run([hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB, OBJ_MAIN, OBJ_SORT])
    return LIB


HOST_LIB = os.path.join(HERE, 'libnw_remesh.so')
HOST_SRC = os.path.join(HERE, 'csrc', 'remesh.cpp')
HOST_DEPS = [HOST_SRC, os.path.join(os.path.dirname(HERE), 'include', 'nw_remesh.h')]
HOST_FLAGS = ['-O2', '-std=c++14', '-fPIC', '-shared', '-fvisibility=hidden', '-ffp-contract=off', '-Wall', '-pthread']


def build_host_library(force=False, verbose=False):
    if not force and os.path.exists(HOST_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(HOST_LIB) for d in HOST_DEPS):
        return HOST_LIB
    cmd = [os.environ.get('CXX', 'g++')] + HOST_FLAGS + ['-o', HOST_LIB, HOST_SRC]
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == '__main__':
    build_hip_library(force=True, verbose=True)
    build_host_library(force=True, verbose=True)
