"""A/B of the PCG product (y = A p with the fused p.q partial sums) from the CSR form (k_spmv_csr_dict16) and from
the symmetric half storage (k_spmv_sym in row order, k_spmv_sym_grid3 marching along z with x in LDS) on an n^3 P1
BoxMesh, interleaved rounds in one process.

    python tools/bench_spmv_sym.py [n ...]      (default: 256)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem
from pgdrome_amd import sizes as psizes


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [256]
    ctx = _lib.Context(0)
    for n in sizes:
        coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
        mesh = ctx.mesh_upload(coords, cells)
        info = ctx.mesh_info(mesh)
        nv, nnz = info["nv"], info["nnz"]
        del coords, cells
        ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
        op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0])
        x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, nv))
        y = ctx.vec_alloc(nv)
        alg = psizes.spmv_bytes(nv, nnz)
        used = ctx.op_symmetrize(op)
        print(f"n={n}^3 nv={nv} nnz={nnz}: symmetric storage usable: {used}", flush=True)
        ctx.flags_reset()
        for rnd in range(2):
            for sym, zchunk in ((0, 0), (1, -1), (1, 0), (1, 4), (1, 8), (1, 16), (1, 32), (1, 64)):
                # zchunk = planes per workgroup march of k_spmv_sym_grid3, forced (-1: k_spmv_sym in row order, 0: adaptive)
                ctx.tune(3, sym)
                ctx.tune(6, 0 if zchunk < 0 else 16)
                ctx.tune(7, max(zchunk, 0))
                for _ in range(3):
                    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
                ctx.sync()
                reps = 40
                t0 = time.time()
                for _ in range(reps):
                    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
                ctx.sync()
                wall = (time.time() - t0) / reps
                print(f"  round {rnd} sym {sym} zchunk {zchunk}: {wall*1e6:.1f} us per product+reduce (wall) -> {alg/wall/1e9:.0f} GB/s of the CSR "
                      f"formula = {alg/wall/8e12*100:.1f}% of 8 TB/s; p.q = {ctx.slots_download(30, 1)[0]:.12e}", flush=True)
        ctx.tune(3, 1)
        ctx.tune(6, 16)
        ctx.tune(7, 0)
        for v in (x, y):
            ctx.vec_free(v)
        for a in (ak, am, op):
            ctx.atom_free(a)
        ctx.mesh_free(mesh)
    ctx.close()


if __name__ == "__main__":
    main()
