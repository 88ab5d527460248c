"""March length of the coded product (k_spmv_diac_march2) on an n^3 grid: HIP-event time per launch for forced lengths,
interleaved rounds in one process.    python tools/bench_coded_march.py 256 16 24 32 64
A grid nx x ny x nz (the slab of a rank of the sharded solve): python tools/bench_coded_march.py 256x256x32 0 6 12 18"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem


def main():
    dims = [int(a) for a in sys.argv[1].split("x")]
    nx, ny, nz = dims if len(dims) == 3 else (dims[0],) * 3
    n = sys.argv[1]
    lengths = [int(a) for a in sys.argv[2:]]
    ctx = _lib.Context(0)
    coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), nx - 1, ny - 1, nz - 1)
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
    ref = None
    for rnd in range(3):
        for L in lengths:
            ctx.tune(7, L)
            k0 = ctx.kernel_counts()
            for _ in range(3):
                ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
            assert ctx.kernel_counts()["diac_march"] == k0["diac_march"] + 3
            reps = 60
            ctx.timer_start()
            for _ in range(reps):
                ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
            t = ctx.timer_stop() / reps
            yy = ctx.vec_download(y)
            if ref is None:
                ref = yy
            assert np.array_equal(yy, ref)
            print("round %d  n=%s  march of %4d planes (0 = adaptive): %7.1f us per product+reduce = %5.0f GB/s on 17 B/row"
                  % (rnd, n, L, t * 1e6, 17 * nv / t / 1e9), flush=True)
    ctx.tune(7, 0)


main()
