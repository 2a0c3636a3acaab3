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
    # (the bound is on what the path costs, not on one block's luck with the host's scheduler: the best of three of each)
    for rep in range(2):
        t0 = time.perf_counter()
        cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s)
        cg.synchronize()
        t_with = min(t_with, time.perf_counter() - t0)
    t_without = np.inf
    for rep in range(3):
        t0 = time.perf_counter()
        cg.search(pts, lams=[10.0], num_iters=5, sigma_inv=s, to_host=False)
        cg.synchronize()
        t_without = min(t_without, time.perf_counter() - t0)
    print('block of 5 at 361k vertices: %.3f ms with its result on the host, %.3f ms without' % (t_with * 1e3, t_without * 1e3))
    assert t_with - t_without < 1.5e-3, (t_with, t_without)


def test_stages_inside_the_query_launch_equal_their_own_launches():
    """The ring half of the curvature prior and the attraction step ride in the query launch (workgroups appended to k_nn_wave's grid; the
    attraction step of a work item waits for the item's nearest faces through a word in memory).  NW_RING_IN_NN=0 / NW_ATTRACT_IN_NN=0 make
    each a launch of its own: bit-identical fits (the knobs are read once per process: child processes)."""
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
    for knobs in ({}, {'NW_RING_IN_NN': '0'}, {'NW_ATTRACT_IN_NN': '0'}, {'NW_RING_IN_NN': '0', 'NW_ATTRACT_IN_NN': '0'}):
        env = dict(os.environ, **knobs)
        r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        crcs.append([l for l in r.stdout.splitlines() if l.startswith('CRC')][-1])
    assert len(set(crcs)) == 1, crcs


def test_attraction_step_toggled_at_run_time_gives_the_same_bits():
    """nw_debug(what = 3) moves the attraction step out of the query launch and back (bench.py times its stages apart): the same fit either
    way, block by block, also across the switch."""
    TriMesh, CG = _imports()
    from ch_shrinkwrap_amd import synth
    c = synth.make_config('c3', scale=0.1, seed=6)
    pts, s = c['points'], 1.0 / c['sigma'].ravel()
    outs = {}
    for plan in ('inside', 'apart', 'mixed'):
        cg = CG(TriMesh(c['vertices'].copy(), c['faces']), pts)
        res = []
        for block in range(6):
            if plan == 'apart' or (plan == 'mixed' and block % 2 == 1):
                cg.separate_attraction(True)
            else:
                cg.separate_attraction(False)
            res.append(cg.search(pts, lams=c['lams'], num_iters=5, sigma_inv=s).copy())
        outs[plan] = res
    for a, b, m in zip(outs['inside'], outs['apart'], outs['mixed']):
        assert np.array_equal(a, b) and np.array_equal(a, m)


def test_curvature_tables_built_on_the_device_equal_the_host_substrates(monkeypatch):
    """nw_curvature with NULL tables builds nbr_next / nbr_area from the faces and positions on the device: every one of the twelve outputs
    must equal, bit for bit, the call with the tables the host substrate builds (remesh.ring_tables) -- on a closed mesh, on a mesh with a
    boundary (open fans) and with unused vertex slots; and the fast path of the block boundary (nothing uploaded: the block left the mesh
    on the device) must give what a fresh upload of the same mesh gives."""
    from ch_shrinkwrap_amd import membrane_mesh as mm
    from ch_shrinkwrap_amd.trimesh import icosphere
    from ch_shrinkwrap_amd.synth import sphere_cloud
    names = ('_k_0', '_k_1', '_e_0', '_e_1', '_H', '_K', '_dH', '_dK', '_E', '_pE', '_dE_neighbors')
    v, f = icosphere(4, 100.0)
    v = (v * (1.0 + 0.05 * np.sin(v[:, :1] * 0.07))).astype('f4')
    open_f = f[: f.shape[0] // 2]                                     # half a sphere: a boundary loop
    for faces in (f, open_f):
        out = {}
        for host in ('1', '0'):
            monkeypatch.setenv('NW_HOST_TABLES', host)
            m = mm.MembraneMesh(v.copy(), faces, kc=1.0)
            d = m.curvature_grad_c(dN=0.1)
            out[host] = [d.copy()] + [getattr(m, n).copy() for n in names]
        for a, b in zip(out['1'], out['0']):
            assert np.array_equal(a, b, equal_nan=True)
    # the block boundary's fast path
    monkeypatch.setenv('NW_HOST_TABLES', '0')
    pts = sphere_cloud(20000, 100.0, 5.0, seed=2)
    m = mm.MembraneMesh(v.copy(), f, kc=1.0, step_size=20.0, max_iter=5, remesh_frequency=0, delaunay_remesh_frequency=0)
    m.shrink_wrap(pts, np.full(pts.shape, 5.0, 'f4'))
    assert m._native.mesh_key is not None and m._native.mesh_key[0] == id(m)
    d_fast = m.curvature_grad_c(dN=0.1)
    fast = [d_fast.copy()] + [getattr(m, n).copy() for n in names]
    m._native.mesh_key = None                                         # forget that the device holds it: a fresh upload of positions and normals
    d_up = m.curvature_grad_c(dN=0.1)
    up = [d_up.copy()] + [getattr(m, n).copy() for n in names]
    for a, b in zip(fast, up):
        assert np.array_equal(a, b, equal_nan=True)


def test_driver_with_device_built_tables_follows_the_host_tables_fit(monkeypatch):
    """The driver's block loop lets the library build the 1-ring table and the vertex normals (nw_set_mesh with nbr = nrm = NULL).  The ring
    table is the host substrate's bit for bit (test_hip_parity); the normals are the device's fixed-point sums instead of the host's
    float64 ones -- the same vectors to ~1e-7 --, so two 3-block fits stay within 1e-5 of the bounding box of each other."""
    from conftest import rel_rms
    from ch_shrinkwrap_amd import membrane_mesh as mm, synth
    c = synth.make_config('c2', scale=0.1, seed=5)
    res = {}
    for host in ('1', '0'):
        monkeypatch.setenv('NW_HOST_TABLES', host)
        m = mm.MembraneMesh(c['vertices'].copy(), c['faces'], kc=1.0, step_size=20.0, max_iter=15, remesh_frequency=5, delaunay_remesh_frequency=0)
        m.remesher = None
        m._warned_fixed_topology = True
        m.shrink_wrap(c['points'], c['sigma'])
        res[host] = m.vertices.copy()
        if host == '0':
            assert m.cg._device_tables and m.cg._vertex_neighbors is None            # no host table was built for the optimiser
    assert rel_rms(res['0'], res['1']) <= 1e-5
