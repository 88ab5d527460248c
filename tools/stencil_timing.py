"""Where a step of k_spmv_stencil_march spends its time: needs a library built with -DPGD_STENCIL_TIMING
(HIPCC_EXTRA=-DPGD_STENCIL_TIMING python -m pgdrome_amd.build --force).  Prints, for four workgroups, the s_memtime deltas
(shader clocks) between the phases of every step: [reads + FMAs + stores | stage | fetch issue | barrier]."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ctx = _lib.Context(0)
coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
mesh = ctx.mesh_upload(coords, cells)
nv = ctx.mesh_info(mesh)["nv"]
bnd = np.where(np.any((coords <= 1e-12) | (coords >= 1 - 1e-12), axis=1))[0].astype(np.int32)
del coords, cells
ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0], bnd)
assert ctx.op_symmetrize(op) and ctx.op_classify(op) > 0
x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, nv))
y = ctx.vec_alloc(nv)
ctx.flags_reset()
ctx.tune(36, L)
for _ in range(3):
    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
ctx.sync()
ctx.lib.pgd_debug_read_partials.restype = C.c_int
ctx.lib.pgd_debug_read_partials.argtypes = [C.c_int64, C.POINTER(C.c_double), C.c_int, C.c_int]
out = np.zeros(1024)
# one more launch, then read the stamps before anything overwrites the scratch
ctx.lib.pgd_spmv_dot_slot(ctx.h, op, x, y, x, 0, nv, 30)
assert ctx.lib.pgd_debug_read_partials(ctx.h, out.ctypes.data_as(C.POINTER(C.c_double)), 4096, 1024) == 0
for slot in range(4 if out.any() else 0):
    d = out[slot * 256: slot * 256 + 192].reshape(-1, 4)
    d = d[1:]          # the first row's first delta is an absolute time
    d = d[np.any(d != 0, axis=1)]
    if not len(d):
        continue
    ns = d             # s_memtime: shader clocks
    print("workgroup slot %d: %d steps; mean CLOCKS per phase [reads+FMAs+stores, stage, fetch-issue, barrier] = %s  step total %.0f clk"
          % (slot, len(ns), np.round(ns.mean(axis=0)), ns.sum(axis=1).mean()))
    print("   first 8 steps:", [list(np.round(r)) for r in ns[:8]])
ctx.timer_start()
for _ in range(40):
    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
print("what-if %s: %.1f us per product+reduce" % (os.environ.get("PGD_STENCIL_WHATIF", "0"), ctx.timer_stop() / 40 * 1e6))
print(ctx.kernel_counts())
