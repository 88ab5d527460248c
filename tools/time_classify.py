import sys, time, numpy as np
sys.path.insert(0, ".")
from pgdrome_amd import _lib, fem
ctx = _lib.Context(0)
n = 256
coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
h = ctx.mesh_upload(coords, cells)
ak = ctx.atom_assemble(h, fem.STIFF)
w = ctx.vec_from(1.0 + coords[:, 0] ** 2 + 0.5 * np.sin(coords.sum(axis=1)))
aw = ctx.atom_assemble(h, fem.WMASS, 0, 0, w)
op = ctx.op_combine(h, [ak, aw], [1.0, 0.5])
assert ctx.op_symmetrize(op)
for _ in range(3):
    ctx.sync(); t = time.time(); c = ctx.op_classify(op); ctx.sync(); print("variable coefficient: classes", c, "ms", 1e3 * (time.time() - t))
am = ctx.atom_assemble(h, fem.MASS)
op2 = ctx.op_combine(h, [ak, am], [1.0, 0.5])
assert ctx.op_symmetrize(op2)
for _ in range(3):
    ctx.sync(); t = time.time(); c = ctx.op_classify(op2); ctx.sync(); print("constant coefficient: classes", c, "ms", 1e3 * (time.time() - t))
