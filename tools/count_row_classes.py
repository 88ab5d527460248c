import sys, numpy as np
sys.path.insert(0, ".")
from pgdrome_amd import _lib, fem
ctx = _lib.Context(0)
for n in (128, 256):
    coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
    h = ctx.mesh_upload(coords, cells)
    ak, am = ctx.atom_assemble(h, fem.STIFF), ctx.atom_assemble(h, fem.MASS)
    on = np.where(np.any((coords <= 1e-12) | (coords >= 1 - 1e-12), axis=1))[0].astype(np.int32)
    for name, bc in (("dirichlet", on), ("neumann", np.zeros(0, dtype=np.int32))):
        op = ctx.op_combine(h, [ak, am], [1.0, 5.5], bc)
        assert ctx.op_symmetrize(op)
        print(n, name, "classes of the unscaled operator:", ctx.op_classify(op))
        ctx.atom_free(op)
    del coords, cells
    ctx.atom_free(ak); ctx.atom_free(am); ctx.mesh_free(h)
