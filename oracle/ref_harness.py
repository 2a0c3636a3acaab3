"""
TEST INFRASTRUCTURE -- not part of the product path.

Imports the REFERENCE optimiser (`/root/reference/ch_shrinkwrap/{conj_grad,mesh_conj_grad}.py`) in THIS
container so that (1) the oracle restatement (oracle/nanowrap_oracle.py) can be validated stage by stage and
(2) golden vectors can be generated (tests/golden/make_golden.py).  Nothing here travels to the GPU box as
a dependency: `/root/reference` does not exist there and every caller guards on `available()`.

How: a scratch package dir is assembled under a temp path with SYMLINKS to the
reference's own files where they lie, plus the reference C extension compiled in place by
oracle/Makefile into oracle/_ref/.  The only missing third-party dependency is PYME (plain
ModuleNotFoundError, `delaunay_utils.py:5`, `mesh_conj_grad.py:440`); none of its contents are touched on
the kd-tree branch, so empty placeholder modules are registered in sys.modules for the import to succeed.
No reference source is copied into the repository.
"""
import os
import sys
import glob
import types
import tempfile
import warnings

REF_ROOT = '/root/reference/ch_shrinkwrap'
_HERE = os.path.dirname(os.path.abspath(__file__))
_REF_SO_DIR = os.path.join(_HERE, '_ref')

_mods = None


def available():
    return os.path.isdir(REF_ROOT) and bool(glob.glob(os.path.join(_REF_SO_DIR, 'conj_grad_utils*.so')))


def load():
    """Returns (mesh_conj_grad module, conj_grad module, conj_grad_utils module) of the reference."""
    global _mods
    if _mods is not None:
        return _mods
    if not available():
        raise RuntimeError('reference not available (need /root/reference and oracle/_ref; run `make -C oracle ref`)')
    scratch = tempfile.mkdtemp(prefix='nw_refpkg_')
    pkg = os.path.join(scratch, 'ch_shrinkwrap')
    os.mkdir(pkg)
    for fn in ('__init__.py', 'conj_grad.py', 'mesh_conj_grad.py', 'delaunay_utils.py', 'sdf.py', 'util.py'):
        os.symlink(os.path.join(REF_ROOT, fn), os.path.join(pkg, fn))
    so = glob.glob(os.path.join(_REF_SO_DIR, 'conj_grad_utils*.so'))[0]
    os.symlink(so, os.path.join(pkg, os.path.basename(so)))
    for name in ('PYME', 'PYME.experimental', 'PYME.experimental.isosurface', 'PYME.experimental.octree'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules['PYME'].experimental = sys.modules['PYME.experimental']
    sys.modules['PYME.experimental'].isosurface = sys.modules['PYME.experimental.isosurface']
    sys.modules['PYME.experimental'].octree = sys.modules['PYME.experimental.octree']
    sys.path.insert(0, scratch)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            import ch_shrinkwrap.mesh_conj_grad as mcg
            import ch_shrinkwrap.conj_grad as cg
            import ch_shrinkwrap.conj_grad_utils as cgu
    finally:
        sys.path.remove(scratch)
    _mods = (mcg, cg, cgu)
    return _mods


def load_shapes():
    """The reference's `ch_shrinkwrap.shape` module (CSG shapes over `sdf.py`).  Its module-level
    `from PYME.simulation.locify import points_from_sdf` (shape.py:16) is only used by `Shape.points()`; a placeholder
    that raises is registered so that the signed-distance functions can be evaluated."""
    load()
    for name in ('PYME.simulation', 'PYME.simulation.locify'):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    sys.modules['PYME'].simulation = sys.modules['PYME.simulation']
    sys.modules['PYME.simulation'].locify = sys.modules['PYME.simulation.locify']

    def points_from_sdf(*a, **k):
        raise RuntimeError('PYME is not available')
    sys.modules['PYME.simulation.locify'].points_from_sdf = points_from_sdf
    import importlib.util
    import ch_shrinkwrap
    spec = importlib.util.spec_from_file_location('ch_shrinkwrap.shape', os.path.join(REF_ROOT, 'shape.py'))
    mod = importlib.util.module_from_spec(spec)
    sys.modules['ch_shrinkwrap.shape'] = mod
    spec.loader.exec_module(mod)
    return mod


def load_evaluation_utils():
    """The reference's `ch_shrinkwrap.evaluation_utils` (points_from_mesh :35-150, average_squared_distance :153-180).  Its module-level
    imports pull in the reference's own Cython mesh class (`_membrane_mesh`, not buildable here: it cimports PYME) and `shape`; neither is
    touched by the two functions the fixtures need, so `_membrane_mesh` gets an empty placeholder like the PYME modules."""
    load_shapes()
    import importlib.util
    if 'ch_shrinkwrap._membrane_mesh' not in sys.modules:
        sys.modules['ch_shrinkwrap._membrane_mesh'] = types.ModuleType('ch_shrinkwrap._membrane_mesh')
        import ch_shrinkwrap
        ch_shrinkwrap._membrane_mesh = sys.modules['ch_shrinkwrap._membrane_mesh']
    spec = importlib.util.spec_from_file_location('ch_shrinkwrap.evaluation_utils', os.path.join(REF_ROOT, 'evaluation_utils.py'))
    mod = importlib.util.module_from_spec(spec)
    sys.modules['ch_shrinkwrap.evaluation_utils'] = mod
    spec.loader.exec_module(mod)
    return mod


def new_reference_optimiser(mesh, points, **kw):
    """Construct the reference ShrinkwrapMeshConjGrad against a duck-typed mesh (ch_shrinkwrap_amd.trimesh.TriMesh)
    exactly as `_membrane_mesh.pyx:1510-1512` does, and register it as `mesh.cg` (the mesh's `point_influence`
    property re-enters the optimiser, `_membrane_mesh.pyx:1625-1634`)."""
    mcg, _, _ = load()
    cg = mcg.ShrinkwrapMeshConjGrad(mesh, points, **kw)
    mesh.cg = cg
    return cg
