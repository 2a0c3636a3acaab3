import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ch_shrinkwrap_amd.trimesh import TriMesh, icosphere
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad
from ch_shrinkwrap_amd.synth import sphere_cloud
from oracle import nanowrap_oracle as O
off = np.array([2.0e5, -1.5e5, 1.0e5], 'f4')
v, f = icosphere(3, 120.0)
v = (v + off).astype('f4')
pts = (sphere_cloud(5000, 100.0, 10.0, seed=3) + off).astype('f4')
mesh = TriMesh(v, f)
cent = O.face_centroids(mesh.vertices.copy(), mesh.faces)
d_ref, f_ref = O.nearest_faces(cent, pts)
cg = ShrinkwrapMeshConjGrad(mesh, pts)
os.environ['NW_VERBOSE'] = '1'
cg.search(pts, lams=[10.0], num_iters=1, sigma_inv=1.0 / np.full(pts.size, 10.0, 'f4'))
got = cg.nearest_face
bad = np.nonzero(got != f_ref)[0]
print('mismatches', bad)
for i in bad:
    p = pts[i].astype('f8')
    dg = np.linalg.norm(p - cent[got[i]].astype('f8')); dr = np.linalg.norm(p - cent[f_ref[i]].astype('f8'))
    print(i, 'got', got[i], 'd', repr(dg), 'ref', f_ref[i], 'd', repr(dr), 'diff', dg - dr, 'gpu dist', cg.d[i, 0])
    print('  point', pts[i], 'cent got', cent[got[i]], 'cent ref', cent[f_ref[i]])
