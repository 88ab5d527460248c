"""The dictionary march (k_spmv_diac_march2) on the n^3 BoxMesh with NATURAL boundaries (27 row classes, no stencil form): HIP-event time
per launch of the PCG instance (fused dot, y stored) for the two fetch depths.    python tools/bench_diac.py [n=256]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = _lib.Context(0)
coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
mesh = ctx.mesh_upload(coords, cells)
nv = ctx.mesh_info(mesh)["nv"]
del coords, cells
ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0], None)
assert ctx.op_symmetrize(op) and ctx.op_classify(op) > 0
x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, nv))
y = ctx.vec_alloc(nv)
ctx.flags_reset()
ref = None
for rnd in range(2):
    for depth in (6, 3):
        ctx.tune(24, depth)
        k0 = ctx.kernel_counts()
        for _ in range(3):
            ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
        assert ctx.kernel_counts()["diac_march"] == k0["diac_march"] + 3, ctx.kernel_counts()
        reps = 60
        ctx.timer_start()
        for _ in range(reps):
            ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
        t = ctx.timer_stop() / reps
        yy = ctx.vec_download(y)
        if ref is None:
            ref = yy
        assert np.array_equal(yy, ref)
        print("round %d  n=%d  fetch depth %d: %7.1f us per product+reduce = %5.0f GB/s on 17 B/row = %.3f of 8 TB/s"
              % (rnd, n, depth, t * 1e6, 17 * nv / t / 1e9, 17 * nv / t / 8e12), flush=True)
