"""Developer tool: time the headline workload (C3) per stage for several grid settings in ONE process
(interleaved A/B as cdna_hip_programming.md section 5.4 rule 24 asks).  Usage: python tools/sweep.py "F:B" "F:B" ...
where F = NW_CELL_FACTOR (a multiplier on the cell-size rule; setting it switches the autotuner off) and B = NW_BRICK;
optional :S0 (NW_STAGE0) and :TB (NW_NN_BLOCK).  tools/cell_sweep.py sweeps the ABSOLUTE cell size instead."""
import os
import sys
import time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ch_shrinkwrap_amd import synth                                    # noqa: E402
from ch_shrinkwrap_amd.trimesh import TriMesh                          # noqa: E402
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad     # noqa: E402

cfgname = os.environ.get('SWEEP_CONFIG', 'c3')
scale = float(os.environ.get('SWEEP_SCALE', '1.0'))
cfg = synth.make_config(cfgname, scale=scale, seed=0)
pts, s = cfg['points'], 1.0 / cfg['sigma'].ravel()
settings = sys.argv[1:] or ['1.0:2']
rounds = int(os.environ.get('SWEEP_ROUNDS', '2'))
ref = None
for rnd in range(rounds):
    for st in settings:
        parts = st.split(':')
        f, b = parts[0], parts[1]
        os.environ['NW_CELL_FACTOR'] = f
        os.environ['NW_BRICK'] = b
        os.environ['NW_STAGE0'] = parts[2] if len(parts) > 2 else '1'
        os.environ['NW_NN_BLOCK'] = parts[3] if len(parts) > 3 else '256'
        mesh = TriMesh(cfg['vertices'], cfg['faces'])
        cg = ShrinkwrapMeshConjGrad(mesh, pts)
        for _ in range(2):
            cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)       # warm-up: 10 iterations
        cg.set_profiling(True)
        t0 = time.perf_counter()
        for _ in range(4):
            out = cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
        dt = (time.perf_counter() - t0) / 20
        sm = cg.stage_ms_total
        n = max(sm['update'][1], 1)
        if ref is None:
            ref = out.copy()
        dev = float(np.abs(out - ref).max())
        print('%-14s round %d  wall %.3f ms/iter | dev total %.3f  grid %.3f nn %.3f attract %.3f prior %.3f as %.3f update %.3f | max ring %d mean_d %.2f | maxdiff vs first %.2e'
              % (st, rnd, dt * 1e3, sm['total'][0] / n, sm['grid'][0] / n, sm['nn'][0] / n, sm['attract'][0] / n, sm['prior'][0] / n,
                 sm['as'][0] / n, sm['update'][0] / n, cg.nn_max_ring, cg.mean_dist, dev), flush=True)
        del cg
