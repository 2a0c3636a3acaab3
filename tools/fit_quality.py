"""Fit quality in the reference's metric (ch_shrinkwrap_amd/evaluation.py: mse01, mse10, mse_rms) of complete fits on the BASELINE configurations.
usage: python tools/fit_quality.py c3 1.0 [remesh: 0 | 1 (host remesher) | 2 (device remesher)] [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ch_shrinkwrap_amd import synth, evaluation
from ch_shrinkwrap_amd.membrane_mesh import ShrinkwrapMembrane

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
remesh = int(sys.argv[3]) if len(sys.argv) > 3 else 1
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 39
cfg = synth.make_config(name, scale=scale, seed=0)
truth = synth.truth_cloud(cfg)


class Surf(object):
    vertices, faces = cfg['vertices'], cfg['faces']


pts = cfg['points']
table = {'x': pts[:, 0], 'y': pts[:, 1], 'z': pts[:, 2], 'error_x': cfg['sigma'][:, 0], 'error_y': cfg['sigma'][:, 1], 'error_z': cfg['sigma'][:, 2]}
q0 = evaluation.fit_quality(type('M', (), {'_vertices': {'position': cfg['vertices']}, 'faces': cfg['faces']})(), truth)
mod = ShrinkwrapMembrane(max_iters=iters, remesh_frequency=5, curvature_weight=20.0, neck_first_iter=-1, remesher=[None, 'builtin', 'device'][remesh])
t0 = time.time()
mesh = mod.execute({'surf': Surf, 'filtered_localizations': table})
dt = time.time() - t0
q = evaluation.fit_quality(mesh, truth)
print('%s x%g, %d localizations sigma 10 nm, %d truth points; start mesh (+20 nm): mse_rms %.2f nm; after %d iterations (%s, %d vertices, %.2f s): mse01 %.2f nm^2, mse10 %.2f nm^2, mse_rms %.2f nm'
      % (name, scale, pts.shape[0], truth.shape[0], q0['mse_rms'], iters, ['fixed topology', 'remeshed every 5 on the host', 'remeshed every 5 on the device'][remesh], mesh.vertices.shape[0], dt, q['mse01'], q['mse10'], q['mse_rms']))
