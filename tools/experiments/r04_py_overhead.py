"""cProfile of the host side of a multi-rank block (ONE rank over the library's communicator) beside the plain search: where does the
time between two nw_search calls go?   python tools/experiments/r04_py_overhead.py [scale] [blocks]"""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, parallel
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad, NativeContext

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 100


class One(object):
    get_rank = staticmethod(lambda: 0)
    get_world_size = staticmethod(lambda: 1)


cfg = synth.make_config('c3', scale=scale)
pts, s_inv = cfg['points'], 1.0 / cfg['sigma'].ravel()
for name in ('single', 'tiles'):
    mesh = TriMesh(cfg['vertices'], cfg['faces'])
    native = NativeContext(0)
    comm = parallel.NativeComm(native, One) if name == 'tiles' else None
    cg = ShrinkwrapMeshConjGrad(mesh, pts, native=native)
    scene = parallel.TiledScene(cg, mode='tiles', comm=comm)
    for _ in range(4):
        scene.search(pts, cfg['lams'], 5, s_inv)
    cg.optimize_layout()
    for _ in range(4):
        scene.search(pts, cfg['lams'], 5, s_inv)
    t0 = time.perf_counter()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(blocks):
        scene.search(pts, cfg['lams'], 5, s_inv)
    pr.disable()
    dt = time.perf_counter() - t0
    out = io.StringIO()
    pstats.Stats(pr, stream=out).sort_stats('cumulative').print_stats(14)
    print('== %s: %.1f us per block' % (name, dt / blocks * 1e6))
    print('\n'.join(l for l in out.getvalue().splitlines() if l.strip() and ('{' in l or '.py' in l))[:3000])
    if comm is not None:
        comm.close()
