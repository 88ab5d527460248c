"""Jacobi-PCG against multigrid-PCG (PGD_TUNE_PCG_PRECOND) on the bench's spatial system: -Laplace + mu on an N^3 lattice,
eliminated hull.  usage: python tools/bench_mg.py [N ...]   (default 128 256)"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pgdrome_amd import fem  # noqa: E402


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [128, 256]
    ctx = fem.get_backend()
    for npts in sizes:
        mesh = fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, 1), npts - 1, npts - 1, npts - 1)
        coords = mesh.coordinates()
        h = ctx.mesh_upload(coords, mesh.cells())
        n = coords.shape[0]
        ak, am = ctx.atom_assemble(h, fem.STIFF), ctx.atom_assemble(h, fem.MASS)
        bc = np.where(np.any((coords <= 1e-12) | (coords >= 1 - 1e-12), axis=1))[0].astype(np.int32)
        b = np.ones(n)
        b[bc] = 0.0
        bv = ctx.vec_from(b)
        for prec in (0, 1):
            ctx.tune(40, prec)
            for rep in range(3):
                op = ctx.op_combine(h, [ak, am], [1.0, 5.5], bc)
                xv = ctx.vec_alloc(n)
                ctx.vec_download(xv)                      # (drains the stream)
                t = time.perf_counter()
                it, rel = ctx.pcg_solve(op, bv, xv, 1e-10, 0.0, 10000)
                dt = time.perf_counter() - t
                ctx.vec_free(xv)
                ctx.atom_free(op)
            print("N %d  %s: %d iterations, relres %.2e, %.2f ms per solve = %.1f us per iteration"
                  % (npts, "multigrid" if prec else "Jacobi   ", it, rel, 1e3 * dt, 1e6 * dt / max(it, 1)), flush=True)
        ctx.tune(40, 0)
        print("multigrid solves / fallbacks", ctx.mg_stats(), flush=True)
        for v in (bv,):
            ctx.vec_free(v)
        for a in (ak, am):
            ctx.atom_free(a)
        ctx.mesh_free(h)


if __name__ == "__main__":
    main()
