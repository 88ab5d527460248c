"""k_spmv_stencil_march against k_spmv_diac_march2 (and the CSR kernel as the reference of y) on an n^3 grid, Dirichlet hull:
HIP-event time per launch, interleaved rounds in one process, y compared bit for bit.
    python tools/bench_stencil_march.py 256 0 24 32 36 48 64        (0 = the launcher's own march length)
    python tools/bench_stencil_march.py 256x256x34 0 6 12"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem


def main():
    dims = [int(a) for a in sys.argv[1].split("x")]
    nx, ny, nz = dims if len(dims) == 3 else (dims[0],) * 3
    # march lengths, optionally as LENGTH:DEPTH:WG_PER_CU (fetch depth 3/4/6/8/10, assumed resident workgroups per CU)
    lengths = [a for a in sys.argv[2:]] or ["0"]
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
    ctx.tune(3, 0)
    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
    ref, ref_dot = ctx.vec_download(y), ctx.slots_download(30, 1)[0]
    ctx.tune(3, 1)
    assert ctx.op_symmetrize(op) and ctx.op_classify(op) > 0
    reps = 60
    for rnd in range(3):
        for form, knob35 in (("diac", 0), ("stencil", 1)):
            ctx.tune(35, knob35)
            for spec in (lengths if knob35 else ["0"]):
                parts = [int(t) for t in spec.split(":")] + [0, 0]
                L = parts[0]
                ctx.tune(36, L)
                ctx.tune(38, parts[1])
                ctx.tune(37, parts[2] or 2)
                k0 = ctx.kernel_counts()
                ctx.vec_fill(y, -7.0)
                for _ in range(3):
                    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
                k1 = ctx.kernel_counts()
                ran = [k for k in k1 if k1[k] != k0[k]]
                ctx.timer_start()
                for _ in range(reps):
                    ctx.spmv_dot_slot(op, x, y, x, 0, nv, 30)
                t = ctx.timer_stop() / reps
                yy, dd = ctx.vec_download(y), ctx.slots_download(30, 1)[0]
                same = np.array_equal(yy, ref)
                print("round %d  %dx%dx%d  %-16s march %4d (0 = own rule)  ran %s: %7.1f us per product+reduce  y bit-identical to CSR: %s  "
                      "dot rel diff %.1e" % (rnd, nx, ny, nz, form + " " + spec, L, ran, t * 1e6, same, abs(dd - ref_dot) / abs(ref_dot)), flush=True)
                if not same:
                    bad = np.where(yy != ref)[0]
                    print("   first mismatches at rows", bad[:10], "of", bad.size, " got", yy[bad[:4]], "want", ref[bad[:4]], flush=True)
    ctx.tune(35, 1)
    ctx.tune(36, 0)
    ctx.tune(38, 0)
    ctx.tune(37, 2)


main()
