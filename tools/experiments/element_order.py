"""What does the ORDER of vertices and faces in the caller's arrays cost?  C3 start mesh as generated (surface-nets scan order), with
Morton-sorted vertex and face ids, and with randomly shuffled ids (what seven remeshing passes tend towards).  Per-stage HIP-event
times over iterations 15-45.  usage: element_order.py [config] [scale]"""
import sys, numpy as np
sys.path.insert(0, '.')
from ch_shrinkwrap_amd import synth
from ch_shrinkwrap_amd.trimesh import TriMesh
from ch_shrinkwrap_amd.mesh_conj_grad import ShrinkwrapMeshConjGrad

def morton(p, bits=10):
    q = ((p - p.min(0)) / (p.max(0) - p.min(0)).max() * ((1 << bits) - 1)).astype(np.uint64)
    code = np.zeros(p.shape[0], np.uint64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> np.uint64(b)) & np.uint64(1)) << np.uint64(3 * b + a)
    return code

def relabel(v, f, vorder):
    inv = np.empty(v.shape[0], np.int64); inv[vorder] = np.arange(v.shape[0])
    v2 = v[vorder]; f2 = inv[f].astype(np.int32)
    return v2, f2

name = sys.argv[1] if len(sys.argv) > 1 else 'c3'
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cfg = synth.make_config(name, scale=scale, seed=0)
v, f, pts = cfg['vertices'], cfg['faces'], cfg['points']
s = 1.0 / cfg['sigma'].ravel()
rng = np.random.default_rng(0)
variants = {}
variants['as generated'] = (v, f)
vo = np.argsort(morton(v), kind='stable'); v2, f2 = relabel(v, f, vo)
fo = np.argsort(morton(v2[f2].mean(1)), kind='stable'); variants['morton'] = (v2, f2[fo])
variants['morton faces only'] = (v, f[np.argsort(morton(v[f].mean(1)), kind='stable')])
variants['morton vertices only'] = (v2, f2)
vo = rng.permutation(v.shape[0]); v3, f3 = relabel(v, f, vo); variants['shuffled'] = (v3, f3[rng.permutation(f.shape[0])])
variants['shuffled, morton faces only'] = (v3, f3[np.argsort(morton(v3[f3].mean(1)), kind='stable')])
for tag, (vv, ff) in variants.items():
    mesh = TriMesh(vv.copy(), ff.copy())
    cg = ShrinkwrapMeshConjGrad(mesh, pts)
    for b in range(3):
        cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
    cg.optimize_layout()
    cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
    cg.set_profiling(2)
    for b in range(6):
        cg.search(pts, lams=cfg['lams'], num_iters=5, sigma_inv=s)
    st = cg.stage_ms_total
    n = max(st['update'][1], 1)
    print('%-28s' % tag, ' '.join('%s %.1f' % (k, 1e3 * st[k][0] / n) for k in ('total', 'grid', 'nn', 'attract', 'prior', 'as', 'update')), 'us per iteration')
