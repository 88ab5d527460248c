import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from pgdrome_amd import _lib, fem
n = 128
ctx = _lib.Context(0)
coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
mesh = ctx.mesh_upload(coords, cells)
nv = ctx.mesh_info(mesh)["nv"]
bnd = np.where(np.any((coords <= 1e-12) | (coords >= 1 - 1e-12), axis=1))[0].astype(np.int32)
ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0], bnd)
b = ctx.vec_from(np.random.default_rng(99).uniform(-1, 1, nv))
y = ctx.vec_alloc(nv)
print(ctx.kernel_counts())
print(ctx.pcg_solve(op, b, y, 0.0, 0.0, 12))
print(ctx.kernel_counts())
for unit in (0, 1):
    ctx.tune(17, unit)
    op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0], bnd)
    k0 = ctx.kernel_counts()
    ctx.pcg_solve(op, b, y, 0.0, 0.0, 12)
    k1 = ctx.kernel_counts()
    print("unit_diag", unit, {k: k1[k] - k0[k] for k in k1})
