"""
GPU tests added in round 5 (run with -m gpu on an MI355X), all through the C-ABI:
  * the mesh's vertex records written BEHIND a block (NW_FLAG_ROWS_ASYNC, trimesh.TriMesh._vertices) hold what the synchronous
    write-back leaves, and nobody can read them half-written;
  * a result above the 4 MB limit of the direct output comes back in slices announced through the flag word, and the host tail of such
    a block is bounded (c5 / C4 had grown a 2.3 ms tail per block);
  * the ring half of the curvature prior computed inside the query launch equals the same half as a launch of its own.
"""
import ctypes
import os
import subprocess
import sys
import time

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _imports():
    from ch_shrinkwrap_amd.trimesh import TriMesh
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    return TriMesh, ShrinkwrapMeshConjGrad


def test_vertex_records_written_behind_a_block_are_the_synchronous_ones():
    """Two identical fits of 4 blocks, one with the records written while the caller goes on, one with the round-4 write-back: same
    results, same records (valid rows updated, slots of deleted vertices untouched); a read of `mesh._vertices` between two blocks sees
    the block that has just returned, complete."""
    TriMesh, CG = _imports()
    from ch_shrinkwrap_amd import synth, mesh_conj_grad
    c = synth.make_config('c3', scale=0.05, seed=11)
    pts, s = c['points'], 1.0 / c['sigma'].ravel()
    got = {}
    for deferred in (True, False):
        mesh_conj_grad._ROWS_ASYNC = deferred
        try:
            mesh = TriMesh(c['vertices'].copy(), c['faces'], max_vertices=c['vertices'].shape[0] + 7)     # 7 unused slots at the end
            sentinel = mesh._vertices['position'][-7:].copy()
            cg = CG(mesh, pts)
            outs, recs = [], []
            for block in range(4):
                out = cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
                if deferred:
                    assert (mesh.__dict__.get('_rows_pending') is not None) == True
                if block == 1:
                    recs.append(mesh._vertices['position'].copy())          # a read in between: waits for the rows, sees block 1
                    assert mesh.__dict__.get('_rows_pending') is None
                    assert np.array_equal(recs[-1][:-7], out[:-7])
                outs.append(out.copy())
            cg.synchronize()
            final = mesh._vertices['position'].copy()
            assert np.array_equal(final[-7:], sentinel), 'rows of unused vertex slots were written'
            assert np.array_equal(final[:-7], outs[-1][:-7])
            got[deferred] = (outs, final, recs)
        finally:
            mesh_conj_grad._ROWS_ASYNC = True
    for a, b in zip(got[True][0], got[False][0]):
        assert np.array_equal(a, b)
    assert np.array_equal(got[True][1], got[False][1])
    assert np.array_equal(got[True][2][0], got[False][2][0])


def test_large_result_comes_back_in_announced_slices_and_its_host_tail_is_bounded():
    """361 000 vertices (4.3 MB of positions: above the direct-output limit).  The result must equal the device's estimate, the block must
    take the sliced path exactly once, and what a block with its result costs beyond the same block WITHOUT one (`to_host=False`: same
    kernels, nothing brought back) stays below 1.5 ms -- round 4's c5 blocks had grown a 2.3 ms tail."""
    TriMesh, CG = _imports()
    from ch_shrinkwrap_amd.trimesh import geodesic_sphere
    from ch_shrinkwrap_amd.synth import sphere_cloud
    from ch_shrinkwrap_amd import _lib as nw
    v, f = geodesic_sphere(190, 300.0 * 1.05)
    assert v.shape[0] * 12 > (4 << 20)
    pts = sphere_cloud(400000, 300.0, 10.0, seed=3)
    s = np.full(pts.size, 0.1, 'f4')
    mesh = TriMesh(v, f)
    cg = CG(mesh, pts)
    for block in range(3):                       # cold query, projection re-sort, recorded graph
        cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s)
    cg.synchronize()
    n0 = (ctypes.c_int64 * 2)()
    cg._native.check(cg._L.nw_debug(cg._h, 2, n0, None, 0, None))
    t0 = time.perf_counter()
    out = cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s)
    cg.synchronize()
    t_with = time.perf_counter() - t0
    n1 = (ctypes.c_int64 * 2)()
    cg._native.check(cg._L.nw_debug(cg._h, 2, n1, None, 0, None))
    assert (n1[0] - n0[0], n1[1] - n0[1]) == (0, 1), 'a result above 4 MB takes the sliced path, once'
    dev = np.empty((cg.M, 3), 'f4')
    cg._native.check(cg._L.nw_get(cg._h, nw.NW_ARR_POS, nw.ptr(dev), dev.nbytes))
    assert np.array_equal(out, dev)
    assert np.array_equal(mesh._vertices['position'], dev)
    t0 = time.perf_counter()
    cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s, to_host=False)
    cg.synchronize()
    t_without = time.perf_counter() - t0
    print('block of 5 at 361k vertices: %.3f ms with its result on the host, %.3f ms without' % (t_with * 1e3, t_without * 1e3))
    assert t_with - t_without < 1.5e-3, (t_with, t_without)


def test_ring_half_inside_the_query_launch_equals_its_own_launch():
    """NW_RING_IN_NN=0 computes the ring half of the curvature prior as a launch of its own instead of in workgroups appended to the
    query's grid: bit-identical fits (the knob is read once per process: two child processes)."""
    code = r'''
import sys, numpy as np, zlib
sys.path.insert(0, %r)
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
from ch_shrinkwrap_amd import synth
c = synth.make_config('c2', scale=0.1, seed=4)
pts, s = c['points'], 1.0 / c['sigma'].ravel()
cg = ShrinkwrapMeshConjGrad(TriMesh(c['vertices'].copy(), c['faces']), pts)
for b in range(3):
    out = cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s)
print('CRC', zlib.crc32(np.ascontiguousarray(out).tobytes()), float(np.abs(out).sum()))
''' % ROOT
    crcs = []
    for knob in ('1', '0'):
        env = dict(os.environ, NW_RING_IN_NN=knob)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        crcs.append([l for l in r.stdout.splitlines() if l.startswith('CRC')][-1])
    assert crcs[0] == crcs[1], crcs
