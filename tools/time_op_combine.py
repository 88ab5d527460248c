"""Host cost of pgd_op_combine / pgd_vec_set with the Dirichlet hull of the 256^3 grid (390 152 nodes): the call itself and the
GPU work behind it (the kept index lists: 0.05 ms and 0.1 ms per call; 0.68 ms of k_combine_dia).

    python tools/time_op_combine.py
"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from pgdrome_amd import fem
from pgdrome_amd.hip_backend import HipBackend
be = fem.set_backend(HipBackend(0)); ctx = be.ctx
n = 256
mesh = fem.BoxMesh(fem.Point(0,0,0), fem.Point(1,1,1), n-1, n-1, n-1)
coords = mesh.coordinates()
h = ctx.mesh_upload(coords, mesh.cells())
ak, am = ctx.atom_assemble(h, 1), ctx.atom_assemble(h, 0)
bc = np.where(np.any((coords <= 1e-12) | (coords >= 1 - 1e-12), axis=1))[0].astype(np.int32)
b = ctx.vec_from(np.zeros(coords.shape[0]))
vals = np.zeros(bc.size)
for rep in range(3):
    op = ctx.op_combine(h, [ak, am], [1.0, 3.0 + rep], bc); ctx.atom_free(op)
ctx.sync()
for rep in range(5):
    t0 = time.perf_counter()
    op = ctx.op_combine(h, [ak, am], [1.0, 4.0 + rep], bc)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    ctx.vec_set(b, bc, vals)
    t3 = time.perf_counter()
    ctx.sync()
    t4 = time.perf_counter()
    ctx.atom_free(op)
    t5 = time.perf_counter()
    print("op_combine call %.3f ms (+sync %.3f) | vec_set call %.3f (+sync %.3f) | atom_free %.3f" % (1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), 1e3*(t5-t4)))
