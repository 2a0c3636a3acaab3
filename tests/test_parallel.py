"""
Multi-rank protocol of ch_shrinkwrap_amd.parallel.run_search.

CPU (world_size 2, gloo): the orchestration code that the GPU path uses is driven with an executor built from the ORACLE's
stage functions (test infrastructure), and the sharded result is compared with the single-process oracle run:
  * mode 'replicated': localizations split by spatial tiles, mesh replicated -> all-reduce of the vertex accumulator + 13 scalars
  * mode 'tiles'     : two disjoint vesicles, one per rank -> all-reduce of all 24 scalars; must equal the single-process run
                       on the union scene (ONE global subspace solve, conj_grad.py:202-219).
  * mode 'halo'      : ONE mesh sharded by spatial tiles (owned vertices + halo, ghosts); all-reduce of the boundary rows of the
                       accumulator, of the scalars, and the owners' boundary positions; must equal the single-process run.
GPU: the library's own RCCL communicator (nw_comm_init) with ONE rank -- all a one-GPU box can run of it -- must reproduce nw_search,
recorded as a hipGraph or launched one by one; two gloo ranks sharing the GPU drive the split-phase C-ABI (RCCL refuses two ranks on a device).
"""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import rel_rms
from ch_shrinkwrap_amd import parallel
from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
from ch_shrinkwrap_amd.synth import sphere_cloud

N_SC = 24


class OracleExecutor(object):
    """CPU stand-in for HipExecutor: same phases, same reduction buffers, arithmetic from oracle/nanowrap_oracle.py."""

    n_point_scalars = 13
    n_scalars = N_SC

    def __init__(self, pos, nrm, nbr, faces, points):
        self.pos0, self.nrm, self.nbr, self.faces, self.points = pos, nrm, nbr, faces, points
        self.M = pos.shape[0]
        self.own3 = np.ones(3 * self.M, bool)
        self.max_dist = 0.0

    # -- 'halo' mode hooks (same contract as parallel.HipExecutor: the phases pack / take the boundary rows themselves) ---------------
    n_boundary = 0

    peers = None

    def set_boundary(self, b_local, b_slot, n_boundary, owned_local, gv, n_global, peers=None):
        self.b_local, self.b_slot, self.n_boundary = np.asarray(b_local), torch.from_numpy(np.asarray(b_slot)), int(n_boundary)
        self.own3 = np.repeat(np.asarray(owned_local, bool), 3)
        self.owned = np.asarray(owned_local, bool)
        self.b_owned = self.owned[self.b_local]
        self.gv, self.n_global = np.asarray(gv), int(n_global)
        self.halo = True
        self.peers = None
        if peers is not None:                                  # owner-wise exchange (same contract as parallel.HipExecutor)
            pr, go, gl, oo, ol = [np.asarray(a) for a in peers]
            self.peers = (pr, go, oo)
            self.px_ghost, self.px_owned = torch.from_numpy(gl.astype(np.int64)), torch.from_numpy(ol.astype(np.int64))
            rows = int(max(go[-1], oo[-1], 1))
            self.px_send, self.px_recv = torch.zeros((rows, 4), dtype=torch.float32), torch.zeros((rows, 4), dtype=torch.float32)
            self.px_send3, self.px_recv3 = torch.zeros((rows, 3), dtype=torch.float32), torch.zeros((rows, 3), dtype=torch.float32)
            self.n_boundary = 0

    def peer_segments(self, kind):
        pr, go, oo = self.peers
        snd, rcv = (self.px_send3, self.px_recv3) if kind == 'rows_to_copies' else (self.px_send, self.px_recv)
        so, ro = (go, oo) if kind == 'acc_to_owners' else (oo, go)
        return [(int(pr[k]), snd[int(so[k]):int(so[k + 1])], rcv[int(ro[k]):int(ro[k + 1])]) for k in range(pr.size)]

    def peer_step(self, what, step):
        assert what == 'acc' and step == 1
        no = self.px_owned.numel()
        self.vacc.view(-1, 4).index_add_(0, self.px_owned, self.px_recv[:no])       # the copies' partial sums, added to the owner's rows
        self.px_send[:no] = self.vacc.view(-1, 4)[self.px_owned]

    def boundary_accumulator(self):
        return self.buf_acc

    def boundary_rows(self):
        return self.buf_rows

    def _pack_acc(self):
        if self.peers is not None:
            self.px_send[:self.px_ghost.numel()] = self.vacc.view(-1, 4)[self.px_ghost]
            return
        self.buf_acc = torch.zeros((self.n_boundary, 4), dtype=torch.float32)
        self.buf_acc[self.b_slot] = self.vacc.view(-1, 4)[self.b_local]

    def _take_acc(self):
        if self.peers is not None:
            self.vacc.view(-1, 4)[self.px_ghost] = self.px_recv[:self.px_ghost.numel()]
            return
        self.vacc.view(-1, 4)[self.b_local] = self.buf_acc[self.b_slot]

    def _pack_rows(self):
        if self.peers is not None:
            self.px_send3[:self.px_owned.numel()] = torch.from_numpy(self.f.reshape(-1, 3)[self.px_owned.numpy()].astype('f4'))
            self.rows_pending = True
            return
        self.buf_rows = torch.zeros((self.n_boundary, 3), dtype=torch.float32)
        rows = self.f.reshape(-1, 3)[self.b_local] * self.b_owned[:, None]
        self.buf_rows[self.b_slot] = torch.from_numpy(rows.astype('f4'))
        self.rows_pending = True

    def _take_rows(self):
        if getattr(self, 'rows_pending', False):
            f = self.f.reshape(-1, 3)
            if self.peers is not None:
                f[self.px_ghost.numpy()] = self.px_recv3[:self.px_ghost.numel()].numpy()
            else:
                f[self.b_local] = self.buf_rows[self.b_slot].numpy()
            self.f = f.ravel()
            self.rows_pending = False

    def gather_owned(self, what='pos'):
        assert what == 'pos'
        full = torch.zeros((self.n_global, 3), dtype=torch.float32)
        full[torch.from_numpy(self.gv[self.owned])] = torch.from_numpy(self.f.reshape(-1, 3)[self.owned].astype('f4'))
        return full.view(-1)

    def set_normals(self, nrm):
        self.nrm = nrm
        self.pos0 = self.f.reshape(-1, 3).copy()             # a new optimiser starts from the mesh's positions

    d0 = None

    def set_reference(self, full_positions, d0=None):
        """per-localization halos (same contract as parallel.HipExecutor.set_reference): with d0 the block's statistic is the largest
        GROWTH of a nearest distance over the one the shares were cut with"""
        self.d0 = None if d0 is None else np.asarray(d0, np.float32)

    def new_tensor(self, values):
        return torch.tensor(values, dtype=torch.float64)

    def begin(self, data, lams, num_iters, sigma_inv, weights, prenormalized, pos, last_step):
        self.lam = float(lams[0])
        self.max_dist = 0.0                                  # per block, like the device-side maximum of the logs
        self.f = self.pos0.copy().ravel()
        self.S = np.zeros((3 * self.M, 3), 'f4')
        self.it = 0
        self.sigma_inv = sigma_inv
        self.wn = prenormalized if prenormalized is not None else (weights if weights is not None else sigma_inv)
        self.sc = torch.zeros(N_SC, dtype=torch.float64)
        self.vacc = torch.zeros(4 * self.M, dtype=torch.float32)

    def attract(self):
        from oracle import nanowrap_oracle as O
        p = self.points
        if getattr(self, 'halo', False):
            self._take_rows()
        self.sc.zero_()
        self.vacc.zero_()
        if p.shape[0] == 0:
            self.wm = None
            if getattr(self, 'halo', False):
                self._pack_acc()
            return
        v_idx, w, dmean, _ = O.weight_matrix(self.f.reshape(-1, 3), self.faces, p)
        Af = O.apply_A(self.f, v_idx, w, p)
        res = (self.wn * (p.ravel() - Af)).astype('f4')
        d3 = np.repeat(dmean, 3)
        res = (res * (1.0 / (d3 * self.sigma_inv / 2.0 + 1))).astype('f4')
        self.wm, self.res = (v_idx, w), res
        self.max_dist = max(self.max_dist, float(dmean.max() if self.d0 is None else (dmean.astype('f4') - self.d0).max()))
        va = np.zeros((self.M, 4), 'f4')
        va[:, :3] = O.apply_At(res, v_idx, w, self.M).reshape(-1, 3)
        va[:, 3] = O.apply_At(np.ones_like(res), v_idx, w, self.M).reshape(-1, 3)[:, 0]
        self.vacc += torch.from_numpy(va.ravel())
        r2 = float((res.astype('f8') ** 2).sum())
        self.sc[0], self.sc[1], self.sc[2], self.sc[3] = r2, r2, float(dmean.sum()), float(p.shape[0])
        if getattr(self, 'halo', False):
            self._pack_acc()

    def vertex_accumulator(self):
        return self.vacc

    def directions(self):
        from oracle import nanowrap_oracle as O
        if getattr(self, 'halo', False):
            self._take_acc()
        va = self.vacc.numpy().reshape(self.M, 4)
        sw = va[:, 3]
        pi = np.sqrt((sw * sw + sw * sw) + sw * sw)
        fdef = O.ncc_prior(self.f.reshape(-1, 3).astype('f4'), self.nrm, self.nbr, pi).ravel()
        p64 = self.f.astype('f8') - fdef
        self.S[:, 0] = va[:, :3].ravel()
        self.S[:, 1] = -1.0 * p64.astype('f4')
        ns = 2 if self.it == 0 else 3
        self.ns = ns
        own = self.own3                    # vertex-side sums: every vertex once, on the rank that owns it
        S = self.S.astype('f8')[own]
        p64o = p64[own]
        idx = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
        for k, (a, b) in enumerate(idx):
            self.sc[13 + k] = float((S[:, a] * S[:, b]).sum()) if max(a, b) < ns else 0.0
        for k in range(3):
            self.sc[19 + k] = float((S[:, k] * p64o).sum()) if k < ns else 0.0
        self.sc[22] = float((p64o * p64o).sum())
        self.sc[23] = float((p64o.astype('f4').astype('f8') ** 2).sum())
        if self.wm is not None:
            v_idx, w = self.wm
            AS = np.stack([O.apply_A(self.S[:, k].copy(), v_idx, w, self.points).astype('f8') if k < ns else np.zeros(self.points.size)
                           for k in range(3)], 1)
            for k, (a, b) in enumerate(idx):
                self.sc[4 + k] = float((AS[:, a] * AS[:, b]).sum())
            for k in range(3):
                self.sc[10 + k] = float((AS[:, k] * self.res.astype('f8')).sum())

    def scalars(self, count):
        return self.sc[:count]

    def update(self):
        sc = self.sc.numpy()
        ns = self.ns
        l2 = self.lam * self.lam
        tri = {(0, 0): 0, (0, 1): 1, (0, 2): 2, (1, 1): 3, (1, 2): 4, (2, 2): 5}
        H = np.zeros((ns, ns), 'f4')
        G = np.zeros(ns, 'f4')
        for a in range(ns):
            G[a] = np.float32(np.float64(np.float32(sc[10 + a])) + l2 * (-sc[19 + a]))
            for b in range(ns):
                k = tri[(min(a, b), max(a, b))]
                H[a, b] = np.float32(np.float64(np.float32(sc[4 + k])) + l2 * np.float64(np.float32(sc[13 + k])))
        c = np.linalg.solve(H, G)
        fnew = (self.f + np.dot(self.S[:, :ns], c)).astype('f4')
        self.S[:, 2] = fnew - self.f
        self.f = fnew
        self.it += 1
        if getattr(self, 'halo', False):
            self._pack_rows()

    def end(self):
        if getattr(self, 'halo', False):
            self._take_rows()
        self.pos0 = self.f.reshape(-1, 3).copy()             # the mesh's positions: where the next search() starts (mesh_conj_grad.py:170)
        return self.f.reshape(-1, 3).copy()


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene(two_vesicles):
    v, f = icosphere(3, 60.0)
    pts = sphere_cloud(6000, 50.0, 5.0, seed=21)
    rng = np.random.default_rng(3)
    sigma = rng.uniform(3.0, 8.0, size=pts.shape).astype('f4')
    if two_vesicles:
        off = np.array([200.0, 0, 0], 'f4')
        v2 = (v * 1.1 + off).astype('f4')
        pts2 = (sphere_cloud(4000, 52.0, 5.0, seed=22) + off).astype('f4')
        sig2 = rng.uniform(4.0, 12.0, size=pts2.shape).astype('f4')
        return [(v, f, pts, sigma), (v2, f, pts2, sig2)]
    return [(v, f, pts, sigma)]


def _worker(rank, world, port, mode, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        if mode == 'replicated':
            (v, f, pts, sigma), = _scene(False)
            mesh = TriMesh(v, f)
            parts = parallel.partition_by_tiles(pts, world)
            mine = parts[rank]
            ex = OracleExecutor(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, pts[mine])
            out = parallel.run_search(ex, dist, 'replicated', pts[mine], [7.0], 4, 1.0 / sigma[mine].ravel())
        elif mode in ('halo', 'halo_tight', 'halo_reach', 'halo_reach3'):
            three = mode == 'halo_reach3'                     # three ranks: a rank has two peers, vertices near the cuts' meeting line have two copies
            mode = 'halo_reach' if three else mode
            (v, f, pts, sigma), = _scene(False)
            mesh = TriMesh(v, f)

            def make(local_mesh, local_points):
                lm = local_mesh
                nb = lm._halfedges['vertex'][lm._vertices['neighbors']]
                nb[lm._vertices['neighbors'] == -1] = -1
                return OracleExecutor(lm.vertices.copy(), lm.vertex_normals.copy(), np.ascontiguousarray(nb, np.int32), lm.faces, local_points)
            # 'halo': a radius with room for the fit's movement -> both blocks on the shares cut at the start;
            # 'halo_tight': largest nearest distance + movement comes close to the radius after the first block -> new shares are cut
            # 'halo_reach': per-localization halos (every face within a localization's own nearest distance + margin)
            # (its margin pays for growth + drift only: 9 nm is 'tight' for this fit -- new shares after the first block)
            scene = parallel.HaloScene(mesh, pts, dist, halo={'halo': 50.0, 'halo_tight': 35.0, 'halo_reach': 9.0}[mode], make_executor=make,
                                       per_point=(mode == 'halo_reach'), min_margin=4.0,
                                       exchange='dense' if mode == 'halo' else 'peers')       # both transports of the boundary rows
            s_inv = 1.0 / sigma.ravel()
            scene.search([7.0], 4, s_inv)
            first = (scene.max_dist, scene.drift)
            scene.refresh_normals()                           # second block on the RESIDENT shares: new normals, same partition
            out = scene.search([7.0], 3, s_inv)
            if not three:
                assert scene.repartitions == {'halo': 1, 'halo_tight': 2, 'halo_reach': 2}[mode], (scene.repartitions, first, scene.max_dist, scene.drift)
            assert getattr(scene, 'redone_blocks', 0) == 0
            part = scene.last_partition
            d = part.ranks[rank]                        # (a rank works out its own share only; the boundary list comes from the all-reduced counts)
            assert all('gv' not in o for r, o in enumerate(part.ranks) if r != rank)
            if three:                                   # both other ranks are peers, and some owned vertex is held by both of them
                pr, go, gl, oo, ol = d['peers']
                assert sorted(pr) == [r for r in range(3) if r != rank]
                q.put((rank, (out, part.boundary.size, int(d['nV']), int(d['owned'].sum()), int(ol.size - np.unique(ol).size))))
                return
            q.put((rank, (out, part.boundary.size, int(d['nV']), int(d['owned'].sum()))))
            return
        else:
            v, f, pts, sigma = _scene(True)[rank]
            mesh = TriMesh(v, f)
            ex = OracleExecutor(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, pts)
            out = parallel.run_search(ex, dist, 'tiles', pts, [7.0], 4, 1.0 / sigma.ravel())
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _run(mode, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.timeout(300)
def test_replicated_mesh_sharded_points_gloo():
    from oracle import nanowrap_oracle as O
    res = _run('replicated')
    (v, f, pts, sigma), = _scene(False)
    mesh = TriMesh(v, f)
    ref = O.search(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, pts, [7.0], 4, 1.0 / sigma.ravel())
    assert np.array_equal(res[0], res[1])                       # replicated state stays bitwise in sync
    assert rel_rms(res[0], ref.positions) <= 1e-5


@pytest.mark.timeout(300)
def test_tiles_two_vesicles_gloo():
    from oracle import nanowrap_oracle as O
    res = _run('tiles')
    (v1, f1, p1, s1), (v2, f2, p2, s2) = _scene(True)
    # single-process run on the union scene: one mesh with two components, one global subspace solve
    V = np.concatenate([v1, v2], 0)
    F = np.concatenate([f1, f2 + v1.shape[0]], 0).astype('i4')
    mesh = TriMesh(V, F)
    P = np.concatenate([p1, p2], 0)
    S = np.concatenate([s1, s2], 0)
    ref = O.search(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, P, [7.0], 4, 1.0 / S.ravel())
    got = np.concatenate([res[0], res[1]], 0)
    assert rel_rms(got, ref.positions) <= 1e-5
    # and it is NOT what two independent fits would give (the coefficient vector c is shared)
    m1 = TriMesh(v1, f1)
    ind = O.search(m1.vertices.copy(), m1.vertex_normals.copy(), m1.neighbor_vertex_table(), m1.faces, p1, [7.0], 4, 1.0 / s1.ravel())
    assert rel_rms(res[0], ind.positions) > 1e-5


@pytest.mark.timeout(300)
def test_halo_sharded_mesh_three_ranks_owner_wise_gloo():
    """Three ranks: every rank exchanges with two peers, and vertices where the tiles meet are held by all three -- the owner's row of such
    a vertex takes two partial sums and goes back twice."""
    from oracle import nanowrap_oracle as O
    res = _run('halo_reach3', world=3)
    (v, f, pts, sigma), = _scene(False)
    mesh = TriMesh(v, f)
    ref = O.search(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, pts, [7.0], 4, 1.0 / sigma.ravel())
    mesh._vertices['position'][:] = ref.positions.astype('f4')
    mesh.update_geometry()
    ref = O.search(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, pts, [7.0], 3, 1.0 / sigma.ravel())
    outs = [res[r][0] for r in range(3)]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    assert rel_rms(outs[0], ref.positions) <= 1e-5
    assert sum(res[r][3] for r in range(3)) == v.shape[0]         # every vertex owned once
    assert sum(res[r][4] for r in range(3)) > 0                   # some owned vertex has copies on both other ranks


@pytest.mark.timeout(300)
@pytest.mark.parametrize('mode', ['halo', 'halo_tight', 'halo_reach'])
def test_halo_sharded_mesh_gloo(mode):
    from oracle import nanowrap_oracle as O
    res = _run(mode)
    (v, f, pts, sigma), = _scene(False)
    mesh = TriMesh(v, f)
    ref = O.search(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, pts, [7.0], 4, 1.0 / sigma.ravel())
    mesh._vertices['position'][:] = ref.positions.astype('f4')
    mesh.update_geometry()
    ref = O.search(mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table(), mesh.faces, pts, [7.0], 3, 1.0 / sigma.ravel())
    (out0, nb, nV0, nown0), (out1, nb1, nV1, nown1) = res[0], res[1]
    nV, nown = [nV0, nV1], [nown0, nown1]
    assert nb == nb1
    assert np.array_equal(out0, out1)                             # every rank ends with the same whole mesh
    assert rel_rms(out0, ref.positions) <= 1e-5
    M = v.shape[0]
    assert sum(nown) == M and 0 < nb < M                          # every vertex owned once; a true sharding with a boundary
    assert max(nV) < M                                            # no rank holds the whole mesh


def test_halo_partition_invariants():
    v, f = icosphere(4, 60.0)
    mesh = TriMesh(v, f)
    pts = sphere_cloud(9000, 50.0, 5.0, seed=4)
    for n in (2, 3, 4, 8):
        part = parallel.HaloPartition(mesh.vertices, mesh.vertex_normals, mesh.neighbor_vertex_table(), mesh.faces, pts, n, halo=20.0)
        owned_total = np.zeros(v.shape[0], int)
        held = np.zeros(v.shape[0], int)
        for r, d in enumerate(part.ranks):
            gv = d['gv']
            assert np.unique(gv).size == gv.size
            owned_total[gv[d['owned'].astype(bool)]] += 1
            held[gv] += 1
            nb = d['nbr']
            # 1-rings of the computed vertices are complete and local; ghosts carry none
            assert (nb[:d['nV']] >= -1).all() and (nb[:d['nV']] < gv.size).all() and (nb[d['nV']:] == -1).all()
            full = mesh.neighbor_vertex_table()[gv[:d['nV']]]
            assert np.array_equal(np.where(nb[:d['nV']] >= 0, gv[np.maximum(nb[:d['nV']], 0)], -1), full)
            # every face centroid within the halo of one of the rank's localizations is among its faces
            cent = mesh.vertices[mesh.faces].mean(1)
            mine = set(map(tuple, np.sort(gv[d['faces']], 1)))
            p = pts[d['pidx']]
            near = np.nonzero((np.abs(cent[:, None, :] - p[None, ::37, :]).max(2) <= 20.0).any(1))[0]
            assert all(tuple(np.sort(mesh.faces[k])) in mine for k in near)
        assert (owned_total == 1).all()
        assert np.array_equal(np.nonzero(held > 1)[0], part.boundary)


def test_faces_within_reach_is_a_superset_of_the_exact_set():
    """Per-localization halos: every centroid within reach_i of localization i must be taken (the classes and voxels may add some,
    never lose one) -- radii spanning two orders of magnitude, points far outside the centroids' box, a single point, no point."""
    from scipy.spatial import cKDTree
    rng = np.random.default_rng(11)
    cent = rng.normal(0.0, 40.0, size=(20000, 3)).astype('f4')
    tree = cKDTree(cent.astype('f8'))
    for n, scale in ((3000, 1.0), (1, 1.0), (500, 30.0)):
        pts = (rng.normal(0.0, 45.0, size=(n, 3)) * np.array([1.0, 1.0, scale])).astype('f4')
        reach = np.exp(rng.uniform(np.log(0.3), np.log(40.0), size=n))
        for voxel in (None, 0.5):
            mask = parallel.faces_within_reach(cent, pts, reach, voxel=voxel)
            exact = np.zeros(cent.shape[0], bool)
            for lst in tree.query_ball_point(pts.astype('f8'), reach):
                exact[lst] = True
            assert not (exact & ~mask).any()
            if voxel is not None and n == 3000:
                assert mask.sum() <= 1.6 * exact.sum() + 50           # ... and not wastefully more
    assert not parallel.faces_within_reach(cent, np.zeros((0, 3), 'f4'), np.zeros(0)).any()


def test_per_localization_halos_hold_what_the_query_needs_and_far_fewer_vertices():
    """Shares cut with reach_i = nearest distance + margin: (1) every face whose centroid lies within reach_i of a rank's localization is
    among the rank's faces -- the local nearest-face query is the global one while growth + drift stay within the margin; (2) the
    partition invariants hold; (3) the shares are much smaller than with one radius for all (which has to cover the localization
    furthest from the surface)."""
    from scipy.spatial import cKDTree
    v, f = icosphere(5, 60.0)
    mesh = TriMesh(v, f)
    pts = sphere_cloud(20000, 60.0, 6.0, seed=4)
    pts[::50] *= 1.5                                              # a few localizations far above the surface
    cent = mesh.vertices[mesh.faces].mean(1)
    tree = cKDTree(cent)
    d0 = tree.query(pts)[0]
    args = (mesh.vertices, mesh.vertex_normals, mesh.neighbor_vertex_table(), mesh.faces, pts)
    for n in (2, 4, 8):
        tiles = parallel.bisect_tiles(pts, n)
        margin = 3.0
        reach = {r: d0[tiles[0][r]] + margin for r in range(n)}
        part = parallel.HaloPartition(*args, n, halo=0.0, tiles=tiles, reach=reach, reach_voxel=margin / 2)
        box = parallel.HaloPartition(*args, n, halo=float(d0.max()) + margin, tiles=tiles)
        owned_total = np.zeros(v.shape[0], int)
        for r, d in enumerate(part.ranks):
            gv = d['gv']
            owned_total[gv[d['owned'].astype(bool)]] += 1
            mine = np.zeros(mesh.faces.shape[0], bool)
            key = {tuple(t) for t in np.sort(gv[d['faces']], 1)}
            sub = tiles[0][r][::7]
            for i, lst in zip(sub, tree.query_ball_point(pts[sub], d0[sub] + margin)):
                assert all(tuple(np.sort(mesh.faces[k])) in key for k in lst)
        assert (owned_total == 1).all()
        held = sum(d['gv'].size for d in part.ranks)
        held_box = sum(d['gv'].size for d in box.ranks)
        assert held < 0.8 * held_box, (n, held, held_box)


@pytest.mark.timeout(900)
@pytest.mark.timeout(600)
@pytest.mark.parametrize('name', ['c3', 'c4'])
def test_eight_rank_partition_of_the_headline_mesh_at_full_size(name):
    """BASELINE configs[2] (C3: 10^6 localizations, 198 812 vertices) and configs[3] (C4: 5 10^6 localizations, 809 955 vertices) at full
    size, 8 ranks, the mesh on the cloud's surface (where a fit spends its time), margin 5 nm: what the ranks hold together, how many
    vertices are shared, the largest share, and what a rank sends per iteration.  (Round 3, one halo radius of 100 nm for all: C3 2.08 x
    the mesh, 70 % boundary vertices, largest share 2.09 x M/8; C4 1.79 / 65 % / 2.24.)  The tiles' cuts go across the axis on which the
    fewest localizations lie near the plane: the same cuts as across the longest axis for C3's vesicle, much shorter ones through C4's
    network (1.30 / 29 % / 1.49 with cuts across the longest axis)."""
    from scipy.spatial import cKDTree
    from ch_shrinkwrap_amd import synth
    cfg = synth.make_config(name)
    pts = cfg['points']
    v0, f0 = cfg['surface']
    mesh = TriMesh(v0, f0)
    pos, faces = mesh.vertices, mesh.faces
    cent = ((pos[faces[:, 0]] + pos[faces[:, 1]]) + pos[faces[:, 2]]) / np.float32(3.0)
    d0 = cKDTree(cent).query(pts, workers=-1)[0]
    tiles = parallel.bisect_tiles(pts, 8, thin_cuts=parallel.THIN_CUTS_NM)          # (HaloScene's choice)
    margin = 5.0
    reach = {r: d0[tiles[0][r]] + margin for r in range(8)}
    part = parallel.HaloPartition(pos, mesh.vertex_normals, mesh.neighbor_vertex_table(), faces, pts, 8, 0.0, tiles=tiles, reach=reach, reach_voxel=margin / 4)      # (HaloScene's choice)
    M = pos.shape[0]
    held = [d['gv'].size for d in part.ranks]
    print('8 ranks of %s, margin %.0f nm: held %.3f x M, boundary %.1f %%, largest share %.3f x M/8' % (name, margin, sum(held) / M, 100.0 * part.boundary.size / M, max(held) / (M / 8)))
    limit = (1.35, 0.30, 1.40)                                                 # (what VERDICT r03 #3 asked for)
    assert sum(held) / M <= limit[0]
    assert part.boundary.size / M <= limit[1]
    assert max(held) / (M / 8) <= limit[2]
    # owner-wise exchange: what a rank sends per iteration (32 B per copy it holds, 28 B per copy others hold of its vertices) against the
    # two dense buffers every rank would all-reduce (44 B per boundary vertex of the whole mesh)
    sent = [parallel.HaloPartition.exchange_bytes(d['peers']) for d in part.ranks]
    npeers = [d['peers'][0].size for d in part.ranks]
    print('   exchange per rank and iteration: %.2f MB on average, %.2f MB at most (dense list: %.2f MB); %d-%d peers per rank' % (
        np.mean(sent) / 1e6, max(sent) / 1e6, 44 * part.boundary.size / 1e6, min(npeers), max(npeers)))
    assert np.mean(sent) <= 0.21 * 44 * part.boundary.size and max(sent) <= 0.31 * 44 * part.boundary.size
    if name == 'c3':
        assert max(sent) <= 0.6e6


def test_a_rank_that_works_out_only_its_own_share_agrees_with_the_full_partition():
    """HaloScene gives every rank its OWN share only (detail_ranks / membership_ranks) and all-reduces the holders' counts: the shares and
    the boundary list must be those of the partition computed in full on one process."""
    v, f = icosphere(4, 60.0)
    mesh = TriMesh(v, f)
    pts = sphere_cloud(9000, 50.0, 5.0, seed=4)
    args = (mesh.vertices, mesh.vertex_normals, mesh.neighbor_vertex_table(), mesh.faces, pts)
    for n in (2, 3, 8):
        tiles = parallel.bisect_tiles(pts, n)
        full = parallel.HaloPartition(*args, n, halo=20.0, tiles=tiles)
        own = [parallel.HaloPartition(*args, n, halo=20.0, tiles=tiles, detail_ranks=(r,), membership_ranks=(r,)) for r in range(n)]
        holders = sum(p.holders for p in own)                     # what the all-reduce does (the ranks' bits are disjoint)
        assert np.array_equal(holders, full.holders)
        for r, p in enumerate(own):
            assert p.boundary is None
            p.set_holders(holders)
            assert np.array_equal(p.count, full.count) and np.array_equal(p.boundary, full.boundary)
            assert all('gv' not in d for q, d in enumerate(p.ranks) if q != r)
            for k, a in full.ranks[r].items():
                if k == 'peers':
                    assert all(np.array_equal(x, y) for x, y in zip(p.ranks[r][k], a)), (n, r, k)
                else:
                    assert np.array_equal(p.ranks[r][k], a), (n, r, k)
        # the owner-wise exchange: rank r's ghost segment for q and q's owned segment for r are the same vertices in the same order;
        # every copy of a vertex appears once, with the vertex's owner; the segments are in ascending global id
        seg = {}
        for r, d in enumerate(full.ranks):
            pr, go, gl, oo, ol = d['peers']
            gv = d['gv']
            assert np.unique(gl).size == gl.size == int((np.asarray(d['owned']) == 0).sum())          # every copy once
            for k, q in enumerate(pr):
                g, o = gv[gl[go[k]:go[k + 1]]], gv[ol[oo[k]:oo[k + 1]]]
                assert (np.diff(g) > 0).all() and (np.diff(o) > 0).all()
                assert (full.owner[g] == q).all() and (full.owner[o] == r).all()
                seg[(r, int(q))] = (g, o)
        for (r, q), (g, o) in seg.items():
            assert (q, r) in seg and np.array_equal(g, seg[(q, r)][1]) and np.array_equal(o, seg[(q, r)][0]), (n, r, q)


def test_margin_policy_cuts_a_long_fit_rarely():
    """A long fit on a fixed topology: growth + drift of a converged mesh still rise by ~1.8 nm per block (measured on C3).  The margin
    doubles when it runs out, so the cuts stay few (a cut costs hundreds of blocks' worth of time) -- the rule "five times the last step"
    alone cut new shares every two or three blocks."""
    halo, min_margin = 60.0, 3.0
    cut_margin, cuts, on_shares, since_shrink = 5.0, 0, 0, 10 ** 9       # shares cut with a small margin (a caller's optimize_layout)
    worst = drift = 0.0
    for block in range(400):
        worst += 0.7
        drift += 1.05
        on_shares += 1
        since_shrink += 1
        cut, margin = parallel.margin_after_block(worst, drift, 1.05, cut_margin, cut_margin, halo, min_margin, True, on_shares, since_shrink)
        assert worst + drift <= cut_margin                           # never beyond what the shares guarantee (the rule cuts BEFORE that block)
        if cut:
            cuts += 1
            if margin < cut_margin:
                since_shrink = 0
            cut_margin, worst, drift, on_shares = margin, 0.0, 0.0, 0
    assert cuts <= 20 and cut_margin == halo                         # 5 -> 10 -> 20 -> 40 -> 60, then once per ~30 blocks
    # one radius for all (no per-localization margin): the rule only says "cut"
    assert parallel.margin_after_block(50.0, 8.0, 2.0, 60.0, 60.0, 60.0, 3.0, False, 5, 5) == (True, None)
    # a hundred quiet blocks on shares with a large margin: a smaller one, never less than half
    assert parallel.margin_after_block(1.0, 2.0, 0.1, 60.0, 60.0, 60.0, 3.0, True, 120, 10 ** 9) == (True, 30.0)
    assert parallel.margin_after_block(1.0, 2.0, 0.1, 60.0, 60.0, 60.0, 3.0, True, 50, 10 ** 9) == (False, None)


@pytest.mark.parametrize('thin', [None, 25.0])
def test_bisect_tiles_cuts_space_consistently(thin):
    """the tiles partition the localizations in balanced counts, and classify() puts every localization into its own tile -- with cuts across
    the longest axis and with the shortest cuts (HaloScene's)"""
    rng = np.random.default_rng(5)
    pts = np.concatenate([sphere_cloud(4000, 50.0, 5.0, seed=2), sphere_cloud(3000, 30.0, 4.0, seed=3) + np.array([140.0, 10.0, 0.0], 'f4'),
                          (rng.normal(size=(2000, 3)) * np.array([60.0, 4.0, 4.0]) + np.array([70.0, 0.0, 0.0])).astype('f4')]).astype('f4')
    for n in (1, 2, 3, 5, 8):
        parts, classify = parallel.bisect_tiles(pts, n, thin_cuts=thin)
        assert len(parts) == n and np.array_equal(np.sort(np.concatenate(parts)), np.arange(pts.shape[0]))
        assert max(p.size for p in parts) - min(p.size for p in parts) <= 2
        owner = classify(pts)
        for r, p in enumerate(parts):
            assert (owner[p] == r).all(), (n, r)


def test_partition_by_tiles_is_a_partition():
    pts = sphere_cloud(5000, 50.0, 5.0, seed=2)
    for n in (1, 2, 3, 4, 8):
        parts = parallel.partition_by_tiles(pts, n)
        allidx = np.sort(np.concatenate(parts))
        assert len(parts) == n and np.array_equal(allidx, np.arange(5000))
        assert max(p.size for p in parts) - min(p.size for p in parts) <= 5000 // n // 2 + 2


class _OneRank(object):
    """the host channel of a one-rank run (NativeComm only asks for rank, world size and a broadcast it does not need)"""

    @staticmethod
    def get_rank():
        return 0

    @staticmethod
    def get_world_size():
        return 1


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tiles', 'replicated'])
def test_library_communicator_equals_search_on_one_gpu(mode):
    """nw_comm_init with one rank (RCCL itself): nw_search with NW_FLAG_COMM_* runs kernels -> ncclAllReduce -> kernels on the library's own
    stream; with one rank the sums are this rank's, so the result must be nw_search's, bit for bit."""
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext
    v, f = icosphere(4, 120.0)
    pts = sphere_cloud(20000, 100.0, 10.0, seed=9)
    rng = np.random.default_rng(1)
    sigma = rng.uniform(5.0, 15.0, size=pts.shape).astype('f4')
    s = 1.0 / sigma.ravel()
    m1 = TriMesh(v, f)
    a = ShrinkwrapMeshConjGrad(m1, pts).search(pts, lams=[10.0], num_iters=5, sigma_inv=s).copy()
    m2 = TriMesh(v, f)
    native = NativeContext(0)
    comm = parallel.NativeComm(native, _OneRank)
    try:
        cg = ShrinkwrapMeshConjGrad(m2, pts, native=native)
        scene = parallel.TiledScene(cg, mode=mode, comm=comm)
        b = scene.search(pts, [10.0], 5, s).copy()
        c = scene.search(pts, [10.0], 3, s).copy()          # second call continues from the mesh
        # the set-up collectives of a run: host and device buffers through the same communicator
        h = comm.all_reduce_host(np.array([1.5, -2.0], np.float64))
        assert np.array_equal(h, [1.5, -2.0])
        assert np.array_equal(comm.all_reduce_host(np.arange(5, dtype=np.int64), parallel.NativeComm.MAX), np.arange(5))
    finally:
        comm.close()
    assert np.array_equal(b, a)
    assert cg.loopcount == 3 and len(cg.tests) == 8
    assert np.array_equal(m2.vertices, c)


@pytest.mark.gpu
@pytest.mark.parametrize('mode', ['tiles', 'replicated', 'halo'])
def test_blocks_recorded_with_their_collectives_equal_launch_by_launch(mode):
    """A block of nw_search with a communicator is recorded -- the phases' launches AND the ncclAllReduce calls between them -- as one
    hipGraph of the library's own stream and replayed for later blocks (profiling level 0; level 4 keeps the block's last iteration
    live); with per-launch profiling (level 2) the same block is issued launch by launch.  Six blocks of 5 (the cell-size tuner changes
    the grid at the third: a new recording), the normals refreshed between blocks in 'halo' mode: bit-identical in all three forms.
    One rank: what a one-GPU box can run of it (RCCL enqueues no kernel for a one-rank all-reduce -- this exercises the capture and
    the host side, not RCCL's kernels as graph nodes)."""
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext
    v, f = icosphere(4, 120.0)
    pts = sphere_cloud(20000, 100.0, 10.0, seed=9)
    s = 1.0 / np.random.default_rng(1).uniform(5.0, 15.0, size=pts.shape).astype('f4').ravel()

    def fit(level):
        mesh = TriMesh(v, f)
        native = NativeContext(0)
        comm = parallel.NativeComm(native, _OneRank)
        outs = []
        try:
            if mode == 'halo':
                scene = parallel.HaloScene(mesh, pts, None, halo=200.0, native=native, comm=comm)
                scene.set_profiling(level)
                for b in range(6):
                    outs.append(scene.search([10.0], 5, s).copy())
                    scene.refresh_normals()
                assert scene.repartitions == 1
            else:
                cg = ShrinkwrapMeshConjGrad(mesh, pts, native=native)
                cg.set_profiling(level)
                scene = parallel.TiledScene(cg, mode=mode, comm=comm)
                for b in range(6):
                    outs.append(scene.search(pts, [10.0], 5, s).copy())
                assert len(cg.tests) == 30
        finally:
            comm.close()
        return outs

    a, b, c = fit(0), fit(2), fit(4)
    for x, y, z in zip(a, b, c):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert not np.array_equal(a[-1], a[-2])


@pytest.mark.gpu
def test_a_status_raised_on_one_rank_travels_with_the_sums():
    """A NaN localization raises NW_ERR_NAN inside an iteration; the status slot of the normal-equation sums carries it (summed over the
    ranks it reaches every one of them: they stop in the same iteration instead of solving with another rank's stale sums).  One rank:
    the slot is set, the block stops where it would on one GPU, the estimate stays at the last good iterate."""
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext
    v, f = icosphere(3, 120.0)
    pts = sphere_cloud(5000, 100.0, 10.0, seed=3)
    mesh = TriMesh(v, f)
    native = NativeContext(0)
    comm = parallel.NativeComm(native, _OneRank)
    try:
        cg = ShrinkwrapMeshConjGrad(mesh, pts, native=native)
        scene = parallel.TiledScene(cg, mode='tiles', comm=comm)
        ok = np.full(pts.size, 0.1, np.float32)
        good = scene.search(pts, [10.0], 2, ok).copy()
        bad = ok.copy()
        bad[3 * 1234 + 1] = np.nan                           # the localizations are fine, so the upload passes: the NaN shows up inside the iteration
        with pytest.raises(AssertionError):
            scene.search(pts, [10.0], 3, bad)
        assert np.array_equal(np.asarray(mesh._vertices['position']), good)
        again = scene.search(pts, [10.0], 1, ok)             # the context stays usable
        assert np.isfinite(again).all() and not np.array_equal(again, good)
    finally:
        comm.close()


# ---- the HIP executor at world_size 2 (two fresh processes sharing cuda:0, gloo carrying the device tensors) -------------------
def _gpu_worker(rank, world, port, mode, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
        ts = torch.cuda.Stream()
        if mode == 'tiles':
            v, f, pts, sigma = _scene(True)[rank]
            mesh = TriMesh(v, f)
            cg = ShrinkwrapMeshConjGrad(mesh, pts, stream=ts.cuda_stream)
            scene = parallel.TiledScene(cg, dist, mode='tiles', torch_stream=ts)
            out = scene.search(pts, [7.0], 4, 1.0 / sigma.ravel()).copy()
            out = scene.search(pts, [7.0], 3, 1.0 / sigma.ravel()).copy()
        elif mode == 'replicated':
            (v, f, pts, sigma), = _scene(False)
            mesh = TriMesh(v, f)
            mine = parallel.partition_by_tiles(pts, world)[rank]
            lp = np.ascontiguousarray(pts[mine])
            cg = ShrinkwrapMeshConjGrad(mesh, lp, stream=ts.cuda_stream)
            scene = parallel.TiledScene(cg, dist, mode='replicated', torch_stream=ts)
            s = 1.0 / sigma[mine].ravel()
            out = scene.search(lp, [7.0], 4, s).copy()
            out = scene.search(lp, [7.0], 3, s).copy()
        else:
            (v, f, pts, sigma), = _scene(False)
            mesh = TriMesh(v, f)
            # 'halo': the boundary rows go between the two ranks that share them (owner-wise exchange, the default); 'halo_dense': two
            # all-reduces over the global list of boundary vertices
            scene = parallel.HaloScene(mesh, pts, dist, halo=50.0 if world == 2 else 12.0, torch_stream=ts, exchange='dense' if mode == 'halo_dense' else 'peers')
            s_inv = 1.0 / sigma.ravel()
            out = scene.search([7.0], 4, s_inv)
            scene.refresh_normals()                           # on the device: shares stay resident, owners' normals go round
            out = scene.search([7.0], 3, s_inv)
            assert scene.repartitions == 1 or world == 3
            assert (scene.ex.peers is None) == (mode == 'halo_dense')
            if mode == 'halo3':                               # three ranks: two peers each
                assert sorted(scene.ex.peers[0]) == [r for r in range(3) if r != rank]
            if mode == 'halo':                                # every copy sends 32 B and gets 28 B back: less than the dense list's 44 B per boundary vertex
                pr, go, oo = scene.ex.peers
                assert list(pr) == [1 - rank] and go[-1] > 0 and oo[-1] > 0
                assert scene.exchange_bytes == 32 * int(go[-1]) + 28 * int(oo[-1]) < 44 * scene.last_partition.boundary.size
            assert np.array_equal(out, mesh._vertices['position'])
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _two_gpu_ranks(mode, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return res


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_owner_wise_exchange_gives_the_bits_of_the_dense_one():
    """The copies get the owner's sum as four float32 (16 B) and store the integer that converts to exactly that float (the quanta are
    powers of two): every holder computes with the numbers the dense all-reduce would have left -- the two transports must give the same
    mesh bit for bit, on two ranks and on three."""
    dense = _two_gpu_ranks('halo_dense')
    peers = _two_gpu_ranks('halo')
    assert np.array_equal(dense[0], dense[1]) and np.array_equal(peers[0], peers[1])
    assert np.array_equal(dense[0], peers[0])
    three = _two_gpu_ranks('halo3', world=3)
    assert np.array_equal(three[0], three[1]) and np.array_equal(three[0], three[2])


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.parametrize('mode', ['tiles', 'replicated', 'halo', 'halo_dense', 'halo3'])
def test_hip_executor_two_ranks_share_one_gpu(mode):
    """The N > 1 HIP path on hardware: two fresh processes, both on cuda:0, run HipExecutor (split-phase C-ABI, device buffers viewed
    by torch, collectives between the phases) in every mode; the result must equal the single-process nw_search fit of the same
    scene (two blocks: 4 + 3 iterations; for 'halo' the vertex normals are refreshed between the blocks on both sides)."""
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    world = 3 if mode == 'halo3' else 2          # 'halo3': three ranks on cuda:0 -- two peers each, vertices with two copies (owner-wise exchange)
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=400) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    if mode == 'tiles':
        (v1, f1, p1, s1), (v2, f2, p2, s2) = _scene(True)
        V = np.concatenate([v1, v2], 0)
        F = np.concatenate([f1, f2 + v1.shape[0]], 0).astype('i4')
        P, S = np.concatenate([p1, p2], 0), np.concatenate([s1, s2], 0)
        got = np.concatenate([res[0], res[1]], 0)
    else:
        (V, F, P, S), = _scene(False)
        assert all(np.array_equal(res[0], res[r]) for r in range(1, world))
        got = res[0]
    mesh = TriMesh(V, F)
    cg = ShrinkwrapMeshConjGrad(mesh, P)
    cg.search(P, lams=[7.0], num_iters=4, sigma_inv=1.0 / S.ravel())
    if mode in ('halo', 'halo_dense', 'halo3'):
        cg.refresh_normals()                             # the single-process form of the same block boundary, on the device
        cg = ShrinkwrapMeshConjGrad(mesh, P, native=cg._native, reuse_device_mesh=True)      # a new optimiser per block
    ref = cg.search(P, lams=[7.0], num_iters=3, sigma_inv=1.0 / S.ravel())
    rms = rel_rms(got, ref)
    print('HIP executor, 2 ranks on one GPU, mode %s: vertex RMS vs single-process nw_search %.3e' % (mode, rms))
    assert rms <= 1e-5


def test_a_sharded_mesh_refuses_to_be_recut_after_every_block():
    """cutting shares costs hundreds of blocks' worth of host time, and the recipe remeshes after every block -- a caller
    that edits the mesh after each of three blocks in a row is told so (unless it insists)."""
    from ch_shrinkwrap_amd.parallel import HaloScene
    sc = object.__new__(HaloScene)
    sc._blocks_total, sc.last_partition, sc.allow_recut_every_block = 0, 'p', False
    sc.mesh_changed()                       # a new mesh before the first block: fine
    sc._blocks_total += 1
    sc.mesh_changed()
    sc._blocks_total += 1
    with pytest.raises(RuntimeError):
        sc.mesh_changed()
    # an edit now and then is what the mode is for
    sc2 = object.__new__(HaloScene)
    sc2._blocks_total, sc2.last_partition, sc2.allow_recut_every_block = 0, 'p', False
    for _ in range(5):
        sc2.mesh_changed()
        sc2._blocks_total += 4
    assert sc2.last_partition is None
    sc3 = object.__new__(HaloScene)
    sc3._blocks_total, sc3.last_partition, sc3.allow_recut_every_block = 0, 'p', True
    for _ in range(5):
        sc3.mesh_changed()
        sc3._blocks_total += 1
