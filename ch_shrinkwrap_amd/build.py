"""
Builds the native pieces in-tree (no pip, no JIT cache):
  * ch_shrinkwrap_amd/libnanowrap_hip.so  -- the HIP kernels + C-ABI (hipcc, --offload-arch=gfx950): csrc/nanowrap.hip, nw_sort.hip, nw_remesh_dev.hip
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
SRC_REMESH = os.path.join(HERE, 'csrc', 'nw_remesh_dev.hip')   # the block-boundary remesher as kernels (hipCUB scans), its own translation unit
OBJ_REMESH = os.path.join(HERE, 'csrc', 'nw_remesh_dev.o')
import glob
# every header of csrc/ is included by nanowrap.hip (directly or through nw_kernels.h): editing any of them must rebuild the library
DEPS = [SRC, SRC_SORT, SRC_REMESH] + sorted(glob.glob(os.path.join(HERE, 'csrc', '*.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'nanowrap.h')]

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
    if force or not os.path.exists(OBJ_REMESH) or os.path.getmtime(OBJ_REMESH) < max(os.path.getmtime(SRC_REMESH), os.path.getmtime(DEPS[-1])):
        run([hipcc, '-O3', '--offload-arch=gfx950', '-fPIC', '-fvisibility=hidden', '-ffp-contract=off', '-Wall', '-Wno-unused-value', '-Wno-unused-function',
             '-c', '-o', OBJ_REMESH, SRC_REMESH])
    run([hipcc] + [f for f in HIPCC_FLAGS if f != '-shared'] + ['-c', '-o', OBJ_MAIN, SRC])
    check_kernel_budgets(verbose=verbose)          # before the link: a kernel that spills or outgrows its occupancy never ships
    run([hipcc, '--offload-arch=gfx950', '-fPIC', '-shared', '-o', LIB, OBJ_MAIN, OBJ_SORT, OBJ_REMESH])
    return LIB


# ---- resource budget of the per-iteration kernels ---------------------------------------------------------------------------------
# The kernels are tuned to an occupancy (waves per SIMD = 512 // VGPRs, capped at 8) that nothing but the register allocator enforces:
# a compiler bump or an innocent edit can spill (scratch > 0: round 2's 20-byte spill of k_nn_wave was only noticed through WRITE_SIZE
# in a profile) or cross a VGPR step and silently halve the waves in flight.  The build reads the gfx950 code object's metadata notes
# (.vgpr_count, .private_segment_fixed_size, .sgpr_spill_count, .group_segment_fixed_size) out of csrc/nanowrap.o and fails on a
# violation; tests/test_abi.py asserts the same numbers.  Budget = (max VGPRs, max LDS bytes); scratch and VGPR spills must be 0.
LLVM_BIN = os.environ.get('NW_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
KERNEL_BUDGETS = {
    # kernel (demangled prefix)       VGPRs  LDS
    'k_nn_wave<false>':               (80, 10 * 1024),     # 6 waves per SIMD (amdgpu_waves_per_eu(6,8)): 12 workgroups of 128 per CU; LDS = the larger of the query's wave-private lists and the appended attraction workgroups' table (9.4 KB)
    'k_attract':                      (72, 20 * 1024),     # 7 workgroups per CU (the run sums keep 24 more registers alive; LDS-pipe-bound: 7 or 8 is the same), 18 KB of LDS each
    'k_face_centroids':               (64, 8 * 1024),
    'k_centroid_scatter':             (64, 0),
    'k_scan_final':                   (64, 1024),
    'k_prior_ring':                   (128, 1024),         # (only launched on its own with NW_RING_IN_NN=0: the ring half rides in the query launch)
    'k_prior_directions':             (96, 1024),          # streaming since the ring half left it: 5 waves per SIMD
    'k_subspace_point_sums':          (128, 1024),         # 4 waves per SIMD cover the launch in one round (rows of two localizations in flight)
    'k_solve_update':                 (128, 1024),
}


def kernel_resources(obj=None):
    """{demangled kernel name: {'vgpr', 'sgpr', 'scratch', 'lds', 'vgpr_spill', 'sgpr_spill'}} of the gfx950 code object inside `obj`."""
    import re
    import tempfile
    obj = obj or OBJ_MAIN
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, 'fat.bin'), os.path.join(td, 'dev.co')
        subprocess.check_call(['objcopy', '-O', 'binary', '--only-section=.hip_fatbin', obj, fat])
        subprocess.check_call([os.path.join(LLVM_BIN, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + fat,
                               '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co])
        notes = subprocess.check_output([os.path.join(LLVM_BIN, 'llvm-readelf'), '--notes', co]).decode()
    out = {}
    for entry in re.split(r'\n\s+- \.agpr_count', notes)[1:]:
        def field(f, default='0'):
            m = re.search(r'\.%s:\s+(\S+)' % f, entry)
            return m.group(1) if m else default
        sym = field('name', '')
        if not sym:
            continue
        # demangled by hand (no c++filt dependency): _Z<len><name>[I L b <0|1> E E]... -> name, name<false>, name<true>
        m = re.match(r'_Z(\d+)', sym)
        name = sym
        if m:
            n0 = m.end()
            name = sym[n0:n0 + int(m.group(1))]
            t = re.match(r'ILb([01])EE', sym[n0 + int(m.group(1)):])
            if t:
                name += '<true>' if t.group(1) == '1' else '<false>'
        out[name] = {'vgpr': int(field('vgpr_count')), 'sgpr': int(field('sgpr_count')), 'scratch': int(field('private_segment_fixed_size')),
                     'lds': int(field('group_segment_fixed_size')), 'vgpr_spill': int(field('vgpr_spill_count')), 'sgpr_spill': int(field('sgpr_spill_count'))}
    return out


def check_kernel_budgets(obj=None, verbose=False):
    """Raise RuntimeError if a budgeted kernel is missing, uses scratch, spills VGPRs, or exceeds its VGPR / LDS budget."""
    res = kernel_resources(obj)
    bad = []
    for k, (max_vgpr, max_lds) in KERNEL_BUDGETS.items():
        r = res.get(k)
        if r is None:
            bad.append('%s: not in the code object' % k)
            continue
        if verbose:
            print('  %-28s %3d VGPRs (<= %3d)  %5d B LDS (<= %5d)  scratch %d' % (k, r['vgpr'], max_vgpr, r['lds'], max_lds, r['scratch']))
        if r['scratch'] or r['vgpr_spill']:
            bad.append('%s: %d bytes of scratch, %d VGPRs spilled' % (k, r['scratch'], r['vgpr_spill']))
        if r['vgpr'] > max_vgpr:
            bad.append('%s: %d VGPRs, budget %d' % (k, r['vgpr'], max_vgpr))
        if r['lds'] > max_lds:
            bad.append('%s: %d bytes of LDS, budget %d' % (k, r['lds'], max_lds))
    if bad:
        raise RuntimeError('kernel resource budget violated:\n  ' + '\n  '.join(bad))
    return res


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
