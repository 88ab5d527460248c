"""A few k_spmv_csr launches on the n^3 P1 BoxMesh for rocprofv3 --pmc passes.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/pmc_spmv.py 256
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem
from pgdrome_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
use_dict = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = _lib.Context(0)
coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
mesh = ctx.mesh_upload(coords, cells)
del coords, cells
ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0])
nv = ctx.mesh_info(mesh)["nv"]
x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, nv))
y = ctx.vec_alloc(nv)
ctx.tune(2, use_dict)     # 1: k_spmv_csr_dict (default), 0: k_spmv_csr
for _ in range(6):
    ctx.spmv(op, x, y)
ctx.sync()
print("done", n, "dict", use_dict, "patterns", ctx.mesh_dict_count(mesh))
