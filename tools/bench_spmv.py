"""Micro-benchmark of k_spmv_csr and the PCG vector kernels on an n^3 P1 BoxMesh.

    python tools/bench_spmv.py [n ...]      (default: 128 256)

Prints achieved algorithmic GB/s (12 nnz + 20 n bytes per launch).
"""
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pgdrome_amd import fem
from pgdrome_amd import sizes as psizes
from pgdrome_amd import _lib


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [128, 256]
    ctx = _lib.Context(0)
    for n in sizes:
        t0 = time.time()
        coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
        t1 = time.time()
        mesh = ctx.mesh_upload(coords, cells)
        ctx.sync()
        t2 = time.time()
        info = ctx.mesh_info(mesh)
        nv, nnz = info["nv"], info["nnz"]
        del coords, cells
        ak = ctx.atom_assemble(mesh, fem.STIFF)
        am = ctx.atom_assemble(mesh, fem.MASS)
        ctx.sync()
        t3 = time.time()
        op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0])
        ctx.sync()
        t4 = time.time()
        print(f"n={n}^3 nv={nv} nnz={nnz} (formula {psizes.nnz_p1_box(n)}) host mesh {t1-t0:.2f}s upload+topology {t2-t1:.2f}s "
              f"2 atoms {t3-t2:.3f}s combine {t4-t3:.3f}s", flush=True)
        x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, nv))
        y = ctx.vec_alloc(nv)
        alg = psizes.spmv_bytes(nv, nnz)
        variants = [(64, 0), (64, 2), (64, 1)]
        for rnd in range(3):                      # interleaved rounds in one process
            for var, grid in variants:
                ctx.tune(1, var)
                ctx.tune(2, grid)
                for _ in range(3):
                    ctx.spmv(op, x, y)
                ctx.sync()
                reps = 40
                t0 = time.time()
                for _ in range(reps):
                    ctx.spmv(op, x, y)
                ctx.sync()
                wall = (time.time() - t0) / reps
                print(f"  round {rnd} rows/workgroup {var} dict {grid}: {wall*1e6:.1f} us (wall) -> {alg/wall/1e9:.0f} GB/s = {alg/wall/8e12*100:.1f}% of 8 TB/s", flush=True)
        ctx.tune(1, 64)
        ctx.tune(2, 1)
        print('  dictionary patterns:', ctx.mesh_dict_count(mesh), flush=True)
        # bilinear (spmv + dot), and a full PCG solve
        t0 = time.time()
        for _ in range(20):
            ctx.bilinear(op, x, x)
        print(f"  bilinear (host round trip each): {(time.time()-t0)/20*1e6:.1f} us", flush=True)
        b = ctx.vec_alloc(nv)
        ctx.spmv(op, x, b)
        ctx.vec_fill(y, 0.0)
        t0 = time.time()
        it, rel = ctx.pcg_solve(op, b, y, rtol=1e-8, maxit=2000)
        dt = time.time() - t0
        print(f"  pcg: {it} iterations, relres {rel:.2e}, {dt*1e3:.1f} ms -> {dt/max(it,1)*1e6:.1f} us/iteration "
              f"({(alg + 88*nv)/ (dt/max(it,1))/1e9:.0f} GB/s of 12nnz+108n)", flush=True)
        for v in (x, y, b):
            ctx.vec_free(v)
        for a in (ak, am, op):
            ctx.atom_free(a)
        ctx.mesh_free(mesh)
    ctx.close()


if __name__ == "__main__":
    main()
