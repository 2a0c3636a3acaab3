#!/usr/bin/env python
"""
bench.py -- vertex-updates/s of the NanoWrap inner loop (force evaluation + subspace "CG" step) on MI355X.

Contract (see the task statement):  python bench.py --gpus N --steps K --warmup W  prints ONE JSON line.
  N > 1 started with plain `python`: this process parses the arguments and -- before torch is imported or the GPU touched -- starts N
  child processes of itself (one rank per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT in their
  environment), relays rank 0's JSON line and exits non-zero if any rank failed.  Started under `python -m torch.distributed.run` (the
  ranks already exist: WORLD_SIZE == N in the environment) it is one of the ranks.

  step      = ONE iteration of ShrinkwrapMeshConjGrad.search (/root/reference/ch_shrinkwrap/mesh_conj_grad.py:218-290):
              grid build + exact nearest-face query, weights, A f, residual, A^T scatter, curvature prior, search
              directions, A.S_k + normal equations, <=3x3 solve, position update.  Iterations are issued in blocks of
              `remesh_frequency` = 5 per search call, as the reference's outer loop does (_membrane_mesh.pyx:1515-1517);
              topology is held fixed between blocks (the remesher is a block-boundary step outside the metric).
  workload  = BASELINE.json configs[2] (the config the metric is quoted on): two-lobe vesicle, 1 000 000 localizations,
              sigma = 10 nm, 198 812-vertex / 397 620-face start mesh offset +20 nm.  Synthetic, seeded, resident in HBM
              before the timed region (upload and optimiser construction are outside it).
  N > 1, --mode tiles (default) = BASELINE.json configs[4]: N such vesicles, one per rank (spatial tiles with an empty boundary set);
              the scene keeps the reference's single global subspace solve, so every iteration all-reduces the 27 normal-equation
              sums over RCCL (ch_shrinkwrap_amd/parallel.py).  WEAK scaling: per-GPU work is fixed.
              (`--config c5 --gpus 1` runs that whole 8-vesicle scene as one mesh on ONE GPU: the single-process result the ranks
              must reproduce, and what one MI355X does with it.)
  N > 1, --mode halo = ONE mesh (the --config workload, c3 by default) sharded over the N ranks by spatial tiles of the cloud
              (BASELINE.json north_star): every rank holds its tile's localizations and the part of the mesh
              within one halo radius of it; per iteration the boundary rows of the accumulator, the normal-equation sums and the
              owners' new boundary positions are all-reduced.  STRONG scaling: the total work is fixed.  (--mode halo --gpus 1 is the
              plain single-GPU run of that mesh, the N = 1 point of the curve.)
  value     = (valid vertices of the whole scene) * K / (max over ranks of the wall time of the K timed steps).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
BLOCK = 5               # remesh_frequency of the headline config
PMC_FILE = 'r05_pmc_traffic.json'     # PMC passes of the headline workload (tools/profile_round.sh), committed under profiles/


def algorithmic_bytes(N, M, F, attract_in_query=False):
    """Compulsory traffic per ITERATION and per kernel, from the itemised list in SURVEY.md section 8(d)
    (4-byte elements, each named array read/written once per stage).  attract_in_query: the launch of k_nn_wave as the timed region runs it
    since round 5 -- the query's workgroups, then the ring half of the prior, then the attraction step of the same work items (workgroups
    appended to the grid): its bytes are the three stages' together."""
    per_kernel = {
        # NN query 12 r + 8 w per point; candidate reads >= 12 per face; + the ring half of _ncc, which rides in this launch since
        # round 5 (workgroups appended to its grid): adjacency 24 + 4, neighbour positions 12, neighbour normals 12 per vertex
        'k_nn_wave': 20 * N + 12 * F + 52 * M,
        # weights 28 r + 24 w, A f + residual 64 r + 12 w, A^T res 36 r, A^T 1 24 r per point;
        # weight gather 12 + A f 12 + A^T outputs 12 + 4 per vertex
        'k_attract': 188 * N + 40 * M,
        # 3 x (A S_k) 3*24 r + 3*12 w, reductions over AS/res 84 r per point; A S_k inputs 36 per vertex
        'k_subspace_point_sums': 192 * N + 36 * M,
        # _ncc 68 (of which 52 -- the 1-ring gathers -- are charged to the query launch, see above), S1 36, reductions over S/prefs 84 per vertex
        'k_prior_directions': 136 * M,
        'k_solve_update': 72 * M,
        # centroid input 12 per vertex; index read 12 + centroid write 12 + grid build 12 + 8 per face
        'grid_build': 12 * M + 44 * F,
    }
    if attract_in_query:
        per_kernel['k_nn_wave'] += per_kernel['k_attract']
    return per_kernel, 436 * N + 348 * M + 68 * F      # SURVEY.md section 8(d) total used by builder and judge


def cpu_baseline(cfg, iters=2):
    """The CPU oracle (NumPy + cKDTree(workers=-1) + C scatter; proven bit-identical to the reference on the golden
    vectors) timed on this box's host cores on the SAME workload for a bounded number of iterations."""
    from oracle import nanowrap_oracle as O
    from ch_shrinkwrap_amd.trimesh import TriMesh
    mesh = TriMesh(cfg['vertices'], cfg['faces'])
    s = 1.0 / cfg['sigma'].ravel()
    pos, nrm, nbr = mesh.vertices.copy(), mesh.vertex_normals.copy(), mesh.neighbor_vertex_table()
    t0 = time.perf_counter()
    r = O.search(pos, nrm, nbr, mesh.faces, cfg['points'], cfg['lams'], iters, s)
    dt = time.perf_counter() - t0
    M = int((mesh._vertices['halfedge'] != -1).sum())
    return dict(value=M * r.loopcount / dt, unit='vertex-updates/s', cores=os.cpu_count(), kind='port',
                sample='%d full-size iterations of the oracle (oracle/nanowrap_oracle.py) on the same %d-localization / %d-vertex '
                       'workload, %.1f s; cKDTree query uses all %d host threads, the rest is single-threaded NumPy/C like the reference'
                       % (r.loopcount, cfg['points'].shape[0], M, dt, os.cpu_count()),
                s_per_iter=dt / max(r.loopcount, 1))


WORKLOADS = {
    'c1': 'BASELINE configs[0]: sphere R=100 nm, icosphere start mesh at 1.2 R',
    'c2': 'BASELINE configs[1]: capped tube r=50 nm, L=1000 nm',
    'c3': 'BASELINE configs[2]: two-lobe vesicle (smooth union of two R=300 nm spheres)',
    'c4': 'BASELINE configs[3]: ERSim2 tube/sheet network with a fenestration (genus 2), twice life size',
    'c5': 'BASELINE configs[4] as ONE scene on one GPU: 8 two-lobe vesicles on a 2x2x2 lattice (what `--gpus 8` shares out, one vesicle per rank)',
}


def measured_copy_ceiling(torch, nbytes=1 << 30, reps=10):
    """Device-to-device copy rate of this GPU (read + write bytes / time): the practical HBM ceiling beside the 8 TB/s spec."""
    a = torch.empty(nbytes // 4, dtype=torch.float32, device='cuda')
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--config', default='c3')
    ap.add_argument('--mode', default='tiles', choices=('tiles', 'halo'),
                    help='N > 1: tiles = one vesicle per rank (BASELINE configs[4], weak scaling); halo = ONE mesh sharded over the ranks (strong scaling)')
    ap.add_argument('--halo', type=float, default=60.0, help='--mode halo: margin (nm) of the first shares -- every rank holds the faces within (nearest distance + margin) of each of its localizations; budget for the growth of a nearest distance + the drift of the mesh until new shares are cut (bench.py cuts them again, with three times the last block\'s movement, after its warm-up)')
    ap.add_argument('--exchange', choices=['peers', 'dense'], default='peers', help="--mode halo: how the boundary rows go round -- 'peers': between the ranks that share them (the copies' partial sums to the vertex's owner, its sum and new position back); 'dense': two all-reduces over the global list of boundary vertices")
    ap.add_argument('--collectives', choices=['library', 'group'], default=os.environ.get('NW_BENCH_COLLECTIVES', 'library'),
                    help="--gpus N over RCCL: 'library' = the library's own communicator issues a block's collectives on its stream, recorded in the block's hipGraph (nw_comm_init; default); "
                         "'group' = the fall-back that needs none of that: the split-phase C-ABI with torch.distributed's all-reduces between the phases (what the gloo tests drive). "
                         "NW_GRAPH_COMM=0 keeps 'library' but launches multi-rank blocks one by one instead of replaying a recording")
    ap.add_argument('--scale', type=float, default=1.0, help='shrink the workload (debug only; the reported config says so)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-graph-pass', action='store_true', help='skip the extra un-instrumented (graph-replay) pass (profiling runs)')
    ap.add_argument('--rank-timeout', type=float, default=600.0, help='--gpus N started with plain python: seconds after which ranks that are still running are ended and the run fails')
    ap.add_argument('--cpu-iters', type=int, default=0, help='oracle iterations for cpu_baseline (0 = about 10-20 s of CPU work: 10 up to 2M localizations, else 3)')
    return ap.parse_args(argv)


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """Parent of an N-rank run started with plain `python`: N fresh children of this script, one per rank.  Nothing here imports torch or
    touches the GPU (a process that has initialised HIP must not be replaced or forked into ranks).  Rank 0's standard output is relayed
    (its JSON line); every rank's standard error is inherited.  First failure ends the others; the exit code is that failure's."""
    import subprocess
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   NW_BENCH_CHILD='1')
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        env.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or 8) // args.gpus)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, universal_newlines=True))
    out0 = []
    rc = 0
    try:
        import threading
        reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.readlines()), daemon=True)
        reader.start()
        pending = set(range(args.gpus))
        deadline = time.monotonic() + max(args.rank_timeout, 1.0)
        while pending:
            if time.monotonic() > deadline:
                rc = 124
                sys.stderr.write('bench.py: ranks %s still running after %.0f s (--rank-timeout); stopping them\n' % (sorted(pending), args.rank_timeout))
                break
            for r in sorted(pending):
                code = procs[r].poll()
                if code is not None:
                    pending.discard(r)
                    if code != 0 and rc == 0:
                        rc = code
                        sys.stderr.write('bench.py: rank %d exited with code %d; stopping the other ranks\n' % (r, code))
            if rc != 0:
                break
            time.sleep(0.05)
        reader.join(timeout=10)
    finally:
        for p in procs:
            if p.poll() is None:
                p.terminate()                      # (exactly the children started above, by handle)
        for p in procs:
            try:
                p.wait(timeout=20)
            except Exception:
                p.kill()
    for line in out0:
        sys.stdout.write(line)
    sys.stdout.flush()
    if rc == 0 and not any(l.startswith('{') for l in out0):
        sys.stderr.write('bench.py: rank 0 printed no JSON line\n')
        rc = 1
    return rc


def main():
    args = parse_args()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if args.gpus > 1 and ('RANK' not in os.environ or world == 1):
        raise SystemExit(spawn_ranks(args))
    if args.gpus != world:
        raise SystemExit('bench.py: --gpus %d but the launcher started %d ranks' % (args.gpus, world))
    run_rank(args)


def run_rank(args):
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the NanoWrap hot path has no CPU fallback')
    # rehearsal hook: NW_BENCH_BACKEND=gloo lets several ranks share the one GPU of a development box (RCCL refuses duplicate
    # devices); the driver's multi-GPU runs use the default, nccl (= RCCL) with one GPU per rank
    backend = os.environ.get('NW_BENCH_BACKEND', 'nccl')
    if backend != 'nccl':
        local_rank = min(local_rank, torch.cuda.device_count() - 1)
    elif world > 1 and torch.cuda.device_count() < world and 'NW_BENCH_CHILD' in os.environ:
        raise SystemExit('bench.py: --gpus %d but only %d GPU(s) visible (RCCL needs one device per rank; NW_BENCH_BACKEND=gloo rehearses on one)'
                         % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    # developer hook: NW_BENCH_FORCE_DIST=1 takes the N > 1 code path (the library's own RCCL communicator, blocks with their collectives
    # recorded in the library's hipGraph) with ONE rank -- what a one-GPU box can run of it (with one rank RCCL enqueues no kernel for an
    # in-place all-reduce: this rehearses the host side and the capture, not RCCL's kernels)
    multi = world > 1 or bool(os.environ.get('NW_BENCH_FORCE_DIST'))
    if multi and world == 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(free_port()))
        os.environ.setdefault('RANK', '0')
        os.environ.setdefault('WORLD_SIZE', '1')
    if multi:
        if backend == 'nccl':
            # (the library records its blocks with their collectives: no buffer registration at capture time -- RCCL reads its parameters
            # once per process, so the default has to be there before torch's process group creates the first communicator)
            os.environ.setdefault('NCCL_GRAPH_REGISTER', '0')
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)

    from ch_shrinkwrap_amd import synth
    from ch_shrinkwrap_amd.trimesh import TriMesh
    from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
    from ch_shrinkwrap_amd import parallel

    halo = multi and args.mode == 'halo'
    # 'halo': every rank generates the SAME scene (seed 0) and takes its share; 'tiles': every rank its own vesicle (seed = rank)
    cfg = synth.make_config(args.config, scale=args.scale, seed=0 if halo else rank)
    if multi and not halo:
        # tile the vesicles on a 2x2x2 lattice (BASELINE.json configs[4]); ranks never share vertices
        off = np.array([(rank & 1), (rank >> 1) & 1, (rank >> 2) & 1], 'f4') * synth.C5_LATTICE
        cfg['points'] = (cfg['points'] + off[None, :]).astype('f4')
        cfg['vertices'] = (cfg['vertices'] + off[None, :]).astype('f4')
    pts, sigma = cfg['points'], cfg['sigma']
    s_inv = 1.0 / sigma.ravel()
    mesh = TriMesh(cfg['vertices'], cfg['faces'])
    N, M, F = pts.shape[0], int((mesh._vertices['halfedge'] != -1).sum()), mesh.faces.shape[0]

    # N > 1 over RCCL (the driver's runs): the LIBRARY owns the communicator and issues a block's collectives itself on its own stream
    # (parallel.NativeComm -> nw_comm_init; torch's process group only carries the communicator's id, the barriers and the timing).
    # NW_BENCH_BACKEND=gloo (ranks sharing one GPU, which RCCL refuses): the same protocol through the split-phase C-ABI with the process
    # group's all-reduces between the phases, kernels and collectives on one dedicated torch stream.
    from ch_shrinkwrap_amd.mesh_conj_grad import NativeContext
    native_comm = multi and backend == 'nccl' and args.collectives == 'library'
    tstream = torch.cuda.Stream() if (multi and not native_comm) else None
    native = NativeContext(local_rank, None) if native_comm else None
    comm = parallel.NativeComm(native, dist) if native_comm else None
    if halo:
        scene = parallel.HaloScene(mesh, pts, dist, halo=args.halo, torch_stream=tstream, native=native, comm=comm, exchange=args.exchange)
        scene.set_profiling(0)
        cg_of = lambda: scene.ex.cg                     # (a re-partition builds a new optimiser over the new share)
    else:
        if native_comm:
            cg = ShrinkwrapMeshConjGrad(mesh, pts, native=native)
        else:
            cg = ShrinkwrapMeshConjGrad(mesh, pts, device=local_rank, stream=tstream.cuda_stream if tstream is not None else None)
        runner = parallel.TiledScene(cg, dist if multi else None, torch_stream=tstream, comm=comm)
        cg_of = lambda: cg

    executed = [0]

    def run_steps(k):
        done = 0
        while done < k:
            n = min(BLOCK, k - done)
            if halo:
                scene.search(cfg['lams'], n, s_inv)
            else:
                runner.search(pts, cfg['lams'], n, s_inv)
            executed[0] += cg_of().loopcount         # iterations that really ran (the device-side stop condition can end a block early)
            done += n

    def set_profiling(level):
        if halo:
            scene.set_profiling(level)
        else:
            cg.set_profiling(level)

    def fence():
        cg_of().synchronize()                    # the library's stream AND its host copy threads (vertex records written behind a block)
        torch.cuda.synchronize()                 # (device-wide: the library's own stream included)
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    run_steps(args.warmup)
    # one-off set-up of the library that would otherwise fall into the first timed block: after its first completed block the
    # library re-sorts the localizations once by their foot point on the surface (radix sort + regather + new work list, ~2 ms)
    # tunes the cell size of the query (a few probe queries) and puts the heavy items of the query's work list first (one timed query)
    # timed region: HIP events on the library's stream around the launch of the dominant kernel (the NN query) in the LAST iteration
    # of every block of 5 -- K/5 live samples; an event pair costs the stream a few microseconds, and one per iteration took ~5 % off
    # the rate being measured.  That last iteration is launched from the host (events recorded inside hipGraph nodes read 0 on
    # ROCm 7.2); the other four iterations of the block are one replayed hipGraph.
    # The full per-stage breakdown is taken in a short extra pass AFTER the timed region.
    set_profiling(4)
    if args.warmup > 0:
        if halo:
            scene.optimize_layout()         # (also: new shares with the margin the fit needs from here on, see HaloScene.optimize_layout)
        else:
            cg_of().optimize_layout()
        # ... and two more untimed blocks after it: the set-up leaves the GPU idle for tens of milliseconds of host work (clocks drop)
        # and re-sorts the localizations (the first block afterwards runs ~25 % slower than the following ones); and a block's recording
        # (hipGraph) is keyed by the staging half its result goes to, which alternates -- two blocks record both, so that no recording falls
        # into the timed region (seen with NW_VERBOSE=3: the first timed block used to record the second half's graph, 0.13 ms of 5)
        run_steps(min(args.warmup, BLOCK))
        if not halo:
            run_steps(min(args.warmup, BLOCK))
        if halo:
            # new shares mean a new sub-mesh on the device: the block above was its first (a cold query), and the library's own set-up
            # (cell tuner, work-list order) needs a warm one -- once more, and one more untimed block behind it
            scene.optimize_layout()
            run_steps(min(args.warmup, BLOCK))
        set_profiling(4)                    # (drops that block's sample of the query kernel: only the timed region's are reported)
    fence()
    warmup_executed = executed[0]
    executed[0] = 0
    reparts0 = scene.repartitions if halo else 0
    t0 = time.perf_counter()
    run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    steps_done = executed[0]
    if steps_done != args.steps:
        raise SystemExit('bench: only %d of %d timed iterations executed (stop condition fired): the throughput would be overstated' % (steps_done, args.steps))
    nn_ms, nn_launches = cg_of().stage_ms_total['nn']
    reparts = (scene.repartitions - reparts0) if halo else 0
    host_ms = dict(scene.host_ms) if halo else {}
    # the same K steps once more WITHOUT events: every block is then one replayed hipGraph (what a caller who does not profile gets).
    # Reported beside the official number, never instead of it.
    dt_graph = None
    if not multi and not args.no_graph_pass:
        cg.set_profiling(0)
        run_steps(BLOCK)                       # captures (first un-instrumented block)
        fence()
        tg = time.perf_counter()
        run_steps(args.steps)
        fence()
        dt_graph = time.perf_counter() - tg
    set_profiling(2)
    # per-stage timings: every stage a launch of its own (in the timed region the attraction step rides in the query launch: nw_debug what = 3)
    fused_query = not multi or native_comm
    if hasattr(cg_of(), 'separate_attraction'):
        cg_of().separate_attraction(True)
    ctimer = None
    if multi and not native_comm:
        ctimer = parallel.CollectiveTimer()          # device time inside the collectives of the extra iterations
        (scene.ex if halo else runner.ex).collective_timer = ctimer
    run_steps(2 * BLOCK)
    fence()
    comm_ms, comm_n = ctimer.total_ms() if ctimer is not None else (0.0, 0)
    stage = dict(cg_of().stage_ms_total)
    n_extra = max(stage['update'][1], 1)
    rccl = None
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if halo:
            M_total = float(M)                       # one mesh: every rank holds the whole of it on its host
        else:
            mt = torch.tensor([float(M)], dtype=torch.float64, device='cuda')
            dist.all_reduce(mt, op=dist.ReduceOp.SUM)
            M_total = float(mt.item())
        devs = [None] * world
        p = torch.cuda.get_device_properties(torch.cuda.current_device())
        dist.all_gather_object(devs, dict(rank=rank, device=torch.cuda.current_device(), name=p.name, pci_bus_id=getattr(p, 'pci_bus_id', None)))
        rccl = dict(backend=dist.get_backend(), world_size_seen=dist.get_world_size(), devices=devs,
                    collectives_issued_by=('the library: ncclAllReduce on the nw_ctx stream between the phases, part of the block\'s hipGraph (nw_comm_init, include/nanowrap.h)'
                                           if native_comm else 'the process group, between the phases of the split-phase C-ABI'))
    else:
        M_total = float(M)

    if rank == 0:
        # sizes of what THIS rank's kernels work on (a sharded mesh: its share)
        if halo:
            d = scene.last_partition.ranks[rank] if scene.last_partition is not None else None
            Nl = scene._local_points.shape[0]
            Ml = int(scene.ex.cg.M)
            Fl = int(np.asarray(scene.ex.cg.faces).shape[0])
        else:
            Nl, Ml, Fl = N, M, F
        per_kernel, per_iter = algorithmic_bytes(Nl, Ml, Fl)
        per_kernel_timed, _ = algorithmic_bytes(Nl, Ml, Fl, attract_in_query=os.environ.get('NW_ATTRACT_IN_NN', '1') != '0')
        # dominant kernel by device time, from HIP events recorded around each launch on the library's stream
        kern = {'nn': 'k_nn_wave', 'attract': 'k_attract', 'as': 'k_subspace_point_sums', 'prior': 'k_prior_directions',
                'update': 'k_solve_update'}          # single-kernel stages (grid build and the NN fix-up are reported in stage_ms_per_iter)
        dom = max((k for k in kern), key=lambda k: stage[k][0] / max(stage[k][1], 1))
        if dom == 'nn' and nn_launches > 0:
            ms_tot, launches = nn_ms, nn_launches            # measured live over the timed region
        else:
            ms_tot, launches = stage[dom]                    # (extra pass: the NN query is no longer the dominant kernel)
        avg_ms = ms_tot / max(launches, 1)
        dom_bytes = per_kernel_timed[kern[dom]] if (dom == 'nn' and nn_launches > 0) else per_kernel[kern[dom]]
        achieved = dom_bytes / (avg_ms * 1e-3) / 1e9
        traffic = None
        tfile = os.path.join(ROOT, 'profiles', PMC_FILE)     # NOT measured by this run
        if os.path.exists(tfile) and args.config == 'c3' and args.scale == 1.0 and world == 1:
            try:
                traffic = json.load(open(tfile)).get(kern[dom])
            except Exception:
                traffic = None
        if not multi:
            par = 'single GPU'
        elif halo:
            par = ('halo%d: ONE mesh sharded by spatial tiles of the cloud (halo radius %.0f nm); per iteration RCCL all-reduces of the boundary rows of the '
                   'fixed-point accumulator (%d rows x 32 B), of the normal-equation sums (nw_n_scalars() slots x 32 ordered parts) and of the owners\' new boundary positions '
                   '(%d rows x 12 B); per block one all-reduce of the owners\' rows of the whole mesh (%d x 12 B)'
                   % (world, args.halo, scene.ex.n_boundary, scene.ex.n_boundary, M))
        else:
            par = ('tiles%d (one vesicle per GPU; the scene keeps ONE global subspace solve: one RCCL all-reduce of the normal-equation '
                   'sums (28 slots x 32 ordered parts = 7.2 KB) per iteration, no vertex data)' % world)
        out = {
            'metric': 'vertex-updates/s (force+CG step) + achieved HBM GB/s, 1M pts/200k verts' if args.config == 'c3' else
                      'vertex-updates/s (force+CG step) + achieved HBM GB/s, %s' % args.config,
            'value': M_total * args.steps / dt,
            'unit': 'vertex-updates/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'warmup_executed': warmup_executed,
            'ms_per_step': dt / args.steps * 1e3,
            'ms_per_step_graph_replay': (dt_graph / args.steps * 1e3) if dt_graph is not None else None,
            'higher_is_better': True,
            'scaling': 'strong' if halo else 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%s, %d localizations sigma=10 nm, %d vertices / %d faces, lams=[10], blocks of %d iterations, fixed topology%s'
                                   % (WORKLOADS[args.config], N, M, F, BLOCK, '' if args.scale == 1.0 else ' [SCALED x%.3g: debug run]' % args.scale),
                       'localizations_per_gpu': Nl, 'vertices_per_gpu': Ml, 'faces_per_gpu': Fl, 'block': BLOCK,
                       'one_off_setup': 'nw_optimize_layout after the warm-up, before the timed region: projection re-sort of the localizations, cell-size tuner (a few timed probe queries), heavy-first order of the query work list (one timed query); then two more untimed blocks of min(warmup, %d) iterations (a block\'s hipGraph is recorded per staging half of its result: both before the timed region): %d iterations ran before the timed region' % (BLOCK, warmup_executed),
                       'mode': args.mode if multi else 'single',
                       'parallelism': par},
            'roofline': {'bound': 'hbm', 'kernel': kern[dom], 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'traffic_source': ('profiles/%s (rocprofv3 --pmc passes of this workload, committed; not re-measured by this run)' % PMC_FILE if traffic is not None else None),
                         'algorithmic_bytes_per_launch': dom_bytes, 'avg_launch_ms': avg_ms, 'launches': launches,
                         'launch_holds': ('the nearest-face query, the ring half of the curvature prior (1-ring gathers) and -- behind a warm query -- the attraction step of the same work items: workgroups appended to k_nn_wave\'s grid run in the query\'s drain (round 5); algorithmic bytes = the three stages\' (SURVEY 8d items)' if dom == 'nn' else None),
                         'launches_note': 'HIP events on the library stream around the k_nn_wave launch of the last iteration of every block of %d of the timed region (that iteration is launched from the host while the replayed hipGraph of the block\'s other iterations is still running)' % BLOCK,
                         'measured_copy_peak': measured_copy_ceiling(torch)},
            'roofline_iteration': {'algorithmic_bytes': per_iter, 'device_ms': stage['total'][0] / n_extra,
                                   'achieved': per_iter / (stage['total'][0] / n_extra * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS,
                                   'unit': 'GB/s', 'note': 'sum of per-stage HIP-event spans of %d extra iterations run after the timed region%s' % (n_extra, ' (rank 0, its share)' if world > 1 else '')},
            'stage_ms_per_iter': {k: stage[k][0] / n_extra for k in stage},
            'stage_ms_note': 'HIP-event spans of %d extra iterations behind the timed region with EVERY stage a launch of its own (nw_debug what = 3); in the timed region the attraction step rides in the query launch, whose live average is roofline.avg_launch_ms' % n_extra,
            'nn_max_ring': cg_of().nn_max_ring, 'mean_dist_nm': cg_of().mean_dist,
        }
        out['roofline_iteration']['frac'] = out['roofline_iteration']['achieved'] / HBM_PEAK_GBS
        if multi:
            out['rccl'] = rccl
            if native_comm:
                out['collectives'] = {'per_iter': 3 if halo else 1, 'ms_per_iter': None,
                                      'note': 'issued inside nw_search (not bracketed by events: they are nodes of the block\'s hipGraph); their cost is the difference between ms_per_step and the device time of the stages'}
            else:
                out['collectives'] = {'ms_per_iter': comm_ms / n_extra, 'per_iter': comm_n / n_extra,
                                      'share_of_device_time': (comm_ms / n_extra) / max(stage['total'][0] / n_extra + comm_ms / n_extra, 1e-12),
                                      'note': 'rank 0, event-bracketed all-reduces of %d extra iterations after the timed region' % n_extra}
            if halo:
                out['halo'] = {'radius_nm': args.halo, 'per_localization_halos': bool(scene.per_point), 'margin_nm': float(getattr(scene, '_cut_margin', args.halo)),
                               'exchange': args.exchange,
                               'exchange_bytes': int(scene.exchange_bytes) + 28 * 32 * 8,
                               'exchange_bytes_note': ('bytes rank 0 SENDS per iteration: 32 B for every copy it holds of a vertex another rank owns (partial accumulator row to the owner), 28 B for every copy another rank holds of a vertex it owns (the sum as four float32 and the new position back), + the 7 KB of normal-equation sums' if args.exchange == 'peers' else 'bytes of the two dense buffers every rank all-reduces per iteration (44 B per boundary vertex of the whole mesh) + the 7 KB of normal-equation sums'),
                               'boundary_vertices': int(scene.boundary_vertices), 'repartitions_in_timed_region': reparts,
                               'max_nn_distance_nm': scene.max_dist, 'drift_since_partition_nm': scene.drift,
                               'host_ms_per_block': host_ms,
                               'vertices_held_over_owned': float(Ml) * world / max(M, 1)}
        # BASELINE.json north_star: ">= 40 % of the HBM-bandwidth roofline on the curvature+attraction kernel" (SURVEY 8d:
        # <= 0.17 ms/iter for them at C3): the attraction/scatter, curvature-prior, A.S and update kernels together, their
        # SURVEY-8d algorithmic bytes over their HIP-event spans
        ca = ('attract', 'prior', 'as', 'update')
        ca_bytes = sum(per_kernel[kern[k]] for k in ca)
        ca_ms = sum(stage[k][0] for k in ca) / n_extra
        out['roofline_attraction_curvature'] = {'kernels': [kern[k] for k in ca], 'algorithmic_bytes': ca_bytes, 'device_ms': ca_ms,
                                                'achieved': ca_bytes / (ca_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                                'frac': ca_bytes / (ca_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                'note': 'SURVEY 8d algorithmic bytes (the unfused itemisation both builder and judge use) over the four kernels\' HIP-event spans as launches of their own; frac_physical = the same spans against the bytes the PMC counters saw (file-sourced)'}
        # what the counters say these four kernels really move (committed PMC passes; lower bound = FETCH_SIZE + WRITE_SIZE, see the file's _bounds)
        if os.path.exists(tfile) and args.config == 'c3' and args.scale == 1.0 and world == 1:
            try:
                pj = json.load(open(tfile))
                phys = sum(pj[kern[k]] for k in ca)
                out['roofline_attraction_curvature'].update({'traffic': phys, 'traffic_upper': sum(pj['_bounds'][kern[k]][1] for k in ca),
                                                             'frac_physical': phys / (ca_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                             'traffic_source': 'profiles/%s (not re-measured by this run)' % PMC_FILE,
                                                             'wait_share': {kern[k]: pj.get('_wait_share', {}).get(kern[k]) for k in ca}})
            except Exception:
                pass
        # the iteration as the timed region runs it: the query launch holds the attraction step (live average) -- the per-stage sum minus the
        # two stages it replaces
        if nn_launches > 0 and stage['attract'][1] > 0:
            fused = stage['total'][0] / n_extra - stage['nn'][0] / max(stage['nn'][1], 1) - stage['attract'][0] / max(stage['attract'][1], 1) + nn_ms / nn_launches
            out['roofline_iteration']['device_ms_fused_query'] = fused
            out['roofline_iteration']['frac_fused_query'] = per_iter / (fused * 1e-3) / 1e9 / HBM_PEAK_GBS
        # The dominant kernel is not an HBM kernel (25 MB algorithmic per launch): it is bound by VALU instruction issue.  VALU
        # wave-instructions per launch from the committed SQ_INSTS_VALU pass (tools/profile_round.sh; file-sourced like `traffic`),
        # against two peaks: the guide's 256 CUs x 4 SIMD x 2.4 GHz / 2 cycles per wave64 instruction, and the ~4.1 cycles per
        # instruction this kernel's mix actually retires at (SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU of the same pass; a micro-benchmark,
        # tools/micro/valu_rate.hip, gives 2.6 cycles for pure-VGPR VOP2, 4.3 with an SGPR operand or VCC, 7.0 for v_pk_fma_f32).
        if traffic is not None and dom == 'nn':
            try:
                vi = json.load(open(tfile)).get('k_nn_wave_valu_wave_instructions')
            except Exception:
                vi = None
            if vi:
                peak = 256 * 4 * 2.4e9 * 0.5
                out['roofline']['valu_issue'] = {'wave_instructions_per_launch': vi, 'source': 'profiles/%s (not re-measured by this run)' % PMC_FILE,
                                                 'achieved': vi / (avg_ms * 1e-3), 'peak': peak, 'unit': 'wave-instructions/s',
                                                 'frac': vi / (avg_ms * 1e-3) / peak, 'frac_of_measured_issue_rate': vi / (avg_ms * 1e-3) / (256 * 4 * 2.4e9 / 4.1)}
        # a shared box is occasionally throttled (every kernel 5-50x slower for a whole call, seen twice in round 2): flag it
        cp = out['roofline'].get('measured_copy_peak', 0.0)
        out['device_health'] = 'ok' if cp >= 3000.0 else 'degraded: device-to-device copy ran at %.0f GB/s (normally ~5100)' % cp
        if not args.no_cpu_baseline and world == 1:
            cpu_iters = args.cpu_iters if args.cpu_iters > 0 else (10 if N <= 2000000 else 3)
            out['cpu_baseline'] = cpu_baseline(synth.make_config(args.config, scale=args.scale, seed=rank), cpu_iters)
            out['speedup_vs_cpu_baseline'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out))
        sys.stdout.flush()
    if multi:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
