"""A few launches of the PCG product on the n^3 P1 BoxMesh for rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_x -- python3 tools/pmc_spmv_sym.py 256 MODE
    MODE: csr (k_spmv_csr_dict16) | rows (k_spmv_dia_rows, row order) | grid (k_spmv_dia_march, z-march, x in LDS) [zchunk] [variant]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "grid"
zchunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0      # 0: the adaptive march length the solves use
variant = int(sys.argv[4]) if len(sys.argv) > 4 else 0
ctx = _lib.Context(0)
for item in filter(None, os.environ.get("PGD_TUNE", "").split(",")):       # e.g. PGD_TUNE=19=0,47=0: the plain march, r02 form
    knob, value = (int(t) for t in item.split("="))
    ctx.tune(knob, value)
coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
mesh = ctx.mesh_upload(coords, cells)
# the bench's operator: homogeneous Dirichlet conditions on the whole hull (PMC_NATURAL=1: natural boundaries, 27 row classes, no stencil form)
bnd = None if os.environ.get("PMC_NATURAL") else np.where(np.any((coords <= 1e-12) | (coords >= 1 - 1e-12), axis=1))[0].astype(np.int32)
del coords, cells
ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0], bnd)
nv = ctx.mesh_info(mesh)["nv"]
x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, nv))
y = ctx.vec_alloc(nv)
ctx.tune(3, 0 if mode == "csr" else 1)
ctx.tune(6, 8 if mode == "grid" else 0)        # the library default (0 turns the march off)
ctx.tune(7, zchunk if mode == "grid" else 0)      # exactly this many planes per march
ctx.tune(13, variant)
if mode != "csr":
    assert ctx.op_symmetrize(op)
ctx.flags_reset()
if mode == "coded":
    # the march on the row-class dictionary as a plain product with the fused dot (k_spmv_diac_march2), 12 launches
    ctx.tune(6, 8)
    ctx.tune(7, zchunk)
    assert ctx.op_classify(op) > 0
    for _ in range(12):
        ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
elif mode == "grid":
    # the PCG instance as the solves launch it: the product of the SCALED operator (unit diagonal not loaded) with the fused
    # dot, inside pgd_pcg_solve - 12 iterations that cannot converge (rtol = atol = 0), issued eagerly (< one graph chunk)
    b = ctx.vec_from(np.random.default_rng(99).uniform(-1, 1, nv))
    ctx.pcg_solve(op, b, y, 0.0, 0.0, 12)
else:
    for _ in range(6):
        ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
ctx.sync()
print("done", n, mode, zchunk, variant, ctx.kernel_counts())
