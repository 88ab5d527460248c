"""The product on a z-slab of the 256 x 256 x nz vertex grid as a row-sharded rank launches it (owned planes between two
ghost planes): row-order kernel vs the march with 2 .. 16 planes per workgroup - picks the rule of the adaptive march length
for short slabs (8 GPUs: 32 owned planes per rank, interior 30).

    python tools/bench_spmv_slab.py [owned_planes ...]      (default: 32 64 128)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [32, 64, 128]
    ctx = _lib.Context(0)
    n = 256
    for owned in sizes:
        nzp = owned + 2
        coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1, 0, nzp - 1)
        mesh = ctx.mesh_upload(coords, cells)
        nv = ctx.mesh_info(mesh)["nv"]
        del coords, cells
        ak, am = ctx.atom_assemble(mesh, fem.STIFF), ctx.atom_assemble(mesh, fem.MASS)
        op = ctx.op_combine(mesh, [ak, am], [1.0, 1.0])
        x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, nv))
        y = ctx.vec_alloc(nv)
        assert ctx.op_symmetrize(op)
        plane = n * n
        ctx.flags_reset()
        for lo, hi, what in ((plane, (nzp - 1) * plane, "owned"), (2 * plane, (nzp - 2) * plane, "interior"), (plane, 2 * plane, "one plane")):
            for name, zk, zf in (("rows", 0, 0), ("adaptive", 16, 0), ("march 1", 16, 1), ("march 2", 16, 2), ("march 3", 16, 3), ("march 4", 16, 4),
                                 ("march 8", 16, 8), ("march 16", 16, 16)):
                ctx.tune(6, zk); ctx.tune(7, zf)
                for _ in range(3):
                    ctx.spmv_dot_slot(op, x, y, x, lo, hi, 30)
                reps = 100
                ctx.timer_start()
                for _ in range(reps):
                    ctx.spmv_dot_slot(op, x, y, x, lo, hi, 30)
                t = ctx.timer_stop() / reps
                rows = hi - lo
                print(f"owned {owned:4d} {what:9s} {name:9s}: {t*1e6:7.1f} us per product+reduce = {80*rows/t/1e9:6.0f} GB/s of its own bytes", flush=True)
        ctx.tune(6, 8); ctx.tune(7, 0)
        for v in (x, y):
            ctx.vec_free(v)
        for a in (ak, am, op):
            ctx.atom_free(a)
        ctx.mesh_free(mesh)
    ctx.close()


if __name__ == "__main__":
    main()
