"""How far does the recipe fit get when it is allowed to converge?  (round 4)
   python tools/experiments/r04_convergence.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from ch_shrinkwrap_amd import synth, evaluation as E
from ch_shrinkwrap_amd.membrane_mesh import ShrinkwrapMembrane


def fit(cfg, **kw):
    class Surf(object):
        vertices, faces = cfg['vertices'], cfg['faces']
    pts = cfg['points']
    table = {'x': pts[:, 0], 'y': pts[:, 1], 'z': pts[:, 2], 'error_x': cfg['sigma'][:, 0], 'error_y': cfg['sigma'][:, 1], 'error_z': cfg['sigma'][:, 2]}
    remesher = kw.pop('remesher', 'builtin')
    mod = ShrinkwrapMembrane(**kw)
    mod.remesher = remesher
    return mod.execute({'surf': Surf, 'filtered_localizations': table})


for name, scale in (('c2', 1.0), ('c4', 0.02)):
    cfg = synth.make_config(name, scale=scale, seed=0)
    truth = synth.truth_cloud(cfg)
    for iters in (39, 79, 159, 319):
        for rem in ('builtin', None):
            t0 = time.time()
            mesh = fit(cfg, max_iters=iters, remesh_frequency=5, curvature_weight=20.0, neck_first_iter=-1, remesher=rem)
            q = E.fit_quality(mesh, truth)
            print('%s x%.2f  %3d iterations  remesher %-8s: mse_rms %.2f  mse01 %.1f  mse10 %.1f  vertices %d  (%.1f s)' % (
                name, scale, iters, rem, q['mse_rms'], q['mse01'], q['mse10'], mesh.vertices.shape[0], time.time() - t0), flush=True)
