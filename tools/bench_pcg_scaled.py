import sys, time, numpy as np
sys.path.insert(0, ".")
from pgdrome_amd import _lib, fem
ctx = _lib.Context(0)
for n in (128, 256):
    coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
    mesh = ctx.mesh_upload(coords, cells); nv = coords.shape[0]
    hull = np.where(np.any((coords < 1e-12) | (coords > 1 - 1e-12), axis=1))[0].astype(np.int32)
    del coords, cells
    ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
    op = ctx.op_combine(mesh, [ak, am], [1.0, 3.0], hull)
    b = ctx.vec_from(np.random.default_rng(3).uniform(0, 1, nv))
    for rnd in range(2):
        for sc in (0, 1):
            ctx.tune(10, sc)
            x = ctx.vec_alloc(nv)
            ctx.sync(); t0 = time.time()
            it, rel = ctx.pcg_solve(op, b, x, rtol=1e-10, maxit=20000)
            ctx.sync(); dt = time.time() - t0
            print(f"n={n} scaled={sc}: {it} iterations, relres {rel:.3e}, {dt*1e3:.1f} ms, {dt/it*1e6:.1f} us/iteration", flush=True)
            ctx.vec_free(x)
    for a in (ak, am, op): ctx.atom_free(a)
    ctx.vec_free(b); ctx.mesh_free(mesh)
