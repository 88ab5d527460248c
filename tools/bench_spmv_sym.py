"""A/B of the PCG product (y = A p with the fused p.q partial sums) from the CSR forms (k_spmv_csr, k_spmv_csr_dict16)
and from the symmetric half storage in diagonal form (k_spmv_dia_rows in row order, k_spmv_dia_march marching along
z with x and the plane-below couplings in LDS) on an n^3 P1 BoxMesh; interleaved rounds in one process, every figure
a HIP-event time over `reps` back-to-back launches on the library's stream.

    python tools/bench_spmv_sym.py [n ...]      (default: 256)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem
from pgdrome_amd import sizes as psizes

MODES = [  # name, sym, dict, zchunk knob, zchunk force, variant, k_spmv_dia_march3 (knob 47)
    ("csr", 0, 0, 8, 0, 0, 0), ("csr_dict16", 0, 1, 8, 0, 0, 0), ("dia_rows", 1, 1, 0, 0, 0, 0),
    ("march2 adaptive", 1, 1, 8, 0, 0, 0), ("march3 adaptive", 1, 1, 8, 0, 0, 1),
    ("march<8> adaptive", 1, 1, 8, 0, 1, 0), ("march<4> adaptive", 1, 1, 8, 0, 2, 0),
    ("march2 z4", 1, 1, 8, 4, 0, 0), ("march2 z16", 1, 1, 16, 16, 0, 0), ("march3 z16", 1, 1, 16, 16, 0, 1), ("march3 z32", 1, 1, 32, 32, 0, 1),
    ("march3 z64", 1, 1, 64, 64, 0, 1),
]


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
        own = {"csr": alg, "csr_dict16": 8 * nnz + 22 * nv}
        used = ctx.op_symmetrize(op)
        print(f"n={n}^3 nv={nv} nnz={nnz}: symmetric storage usable: {used}", flush=True)
        ctx.flags_reset()
        for rnd in range(2):
            for name, sym, dct, zk, zf, var, m3 in MODES:
                ctx.tune(3, sym); ctx.tune(2, dct); ctx.tune(6, zk); ctx.tune(7, zf); ctx.tune(13, var); ctx.tune(47, m3)
                for _ in range(3):
                    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
                reps = 40
                ctx.timer_start()
                for _ in range(reps):
                    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
                t = ctx.timer_stop() / reps
                mine = own.get(name, 80 * nv)
                print(f"  round {rnd} {name:20s}: {t*1e6:7.1f} us per product+reduce; CSR formula {alg/t/1e9:6.0f} GB/s; own minimum "
                      f"{mine/1e9:.3f} GB -> {mine/t/1e9:5.0f} GB/s = {mine/t/8e12*100:4.1f}% of 8 TB/s; p.q = {ctx.slots_download(30, 1)[0]:.12e}",
                      flush=True)
        ctx.tune(3, 1); ctx.tune(2, 1); ctx.tune(6, 8); ctx.tune(7, 0); ctx.tune(13, 0); ctx.tune(47, 0)
        for v in (x, y):
            ctx.vec_free(v)
        for a in (ak, am, op):
            ctx.atom_free(a)
        ctx.mesh_free(mesh)
    ctx.close()


if __name__ == "__main__":
    main()
