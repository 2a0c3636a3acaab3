"""
End-to-end example: fit a membrane to a synthetic single-molecule localization cloud on one MI355X.

    python examples/fit_network.py [scale] [device|builtin]       (scale 0.02 = 100 000 localizations, default; 1.0 = 5 000 000;
                                                                   the block boundary's remesher on the GPU, default, or on the host)

Mirrors what the PYME recipe module `ShrinkwrapMembrane` does upstream
(/root/reference/ch_shrinkwrap/recipe_modules/surface_fitting.py:46-115): a coarse start surface and a table of
localizations go into a namespace, `execute` runs `max_iters` (39, the module's default) NanoWrap iterations in blocks of
`remesh_frequency`, the mesh
is remeshed between blocks towards `minimum_edge_length`, and neck candidates are selected from the Gaussian curvature
after `neck_first_iter`.
"""
import sys
import time
import numpy as np

sys.path.insert(0, __file__.rsplit('/', 2)[0])
from ch_shrinkwrap_amd import synth                               # noqa: E402
from ch_shrinkwrap_amd.membrane_mesh import ShrinkwrapMembrane     # noqa: E402


def main(scale=0.02, remesher='device'):
    cfg = synth.make_config('c4', scale=scale, seed=0)           # ERSim2 tube/sheet network with a fenestration

    class Surf(object):                                           # anything with .vertices / .faces works as the input surface
        vertices, faces = cfg['vertices'], cfg['faces']
    pts = cfg['points']
    table = {'x': pts[:, 0], 'y': pts[:, 1], 'z': pts[:, 2],
             'error_x': cfg['sigma'][:, 0], 'error_y': cfg['sigma'][:, 1], 'error_z': cfg['sigma'][:, 2]}
    ns = {'surf': Surf, 'filtered_localizations': table}
    mod = ShrinkwrapMembrane(max_iters=39, remesh_frequency=5, curvature_weight=20.0, minimum_edge_length=max(5.0, 2.5 / np.sqrt(scale)),
                             neck_first_iter=9, remesher=remesher)
    t0 = time.time()
    mesh = mod.execute(ns)
    dt = time.time() - t0
    sdf = lambda p: 2.0 * synth.sdf_er_sim2(np.asarray(p, 'f8') * 0.5)
    d = sdf(mesh.vertices)
    print('%d localizations, start mesh %d vertices -> fitted mesh %d vertices / %d faces in %.2f s (%d blocks)'
          % (pts.shape[0], cfg['vertices'].shape[0], mesh.vertices.shape[0], mesh.faces.shape[0], dt, len(mesh.block_log)))
    print('distance of the fitted vertices to the true surface: rms %.2f nm (start: 20 nm offset, localization error 10 nm)' % np.sqrt((d * d).mean()))
    for b in mesh.block_log:
        print('  iteration %3d: remesh target %.2f nm -> mean edge %.2f nm' % (b['iteration'], b['target_length'], b['mean_length']))
    for b in mesh.neck_log:
        print('  iteration %3d: %d neck candidates' % (b['iteration'], b['candidates']))
    return mesh, d


if __name__ == '__main__':
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 0.02, sys.argv[2] if len(sys.argv) > 2 else 'device')
