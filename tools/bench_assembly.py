"""k_assemble_p1<3> (the north star's "per-element P1 local-matrix assembly ... into CSR") on the n^3 BoxMesh: time per atom and
the bytes it must move at least once (cell records + coordinates + vertex->cell adjacency + CSR pattern read, values written),
as a fraction of the 8 TB/s peak.      python tools/bench_assembly.py 128 256
Under rocprofv3 --pmc (tools/prof_assembly.sh) the same script gives the kernel's HBM-side traffic and SQ counters."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import _lib, fem

ctx = _lib.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [128, 256]:
    coords, cells = fem.box_mesh_arrays((0, 0, 0), (1, 1, 1), n - 1, n - 1, n - 1)
    mesh = ctx.mesh_upload(coords, cells)
    info = ctx.mesh_info(mesh)
    nv, nc, nnz = info["nv"], info["nc"], info["nnz"]
    del coords, cells
    res = {}
    for name, kind in (("stiffness", fem.STIFF), ("mass", fem.MASS)):
        a = ctx.atom_assemble(mesh, kind)          # warm: allocations
        ctx.atom_free(a)
        ts = []
        for _ in range(3):
            ctx.sync()
            ctx.timer_start()
            a = ctx.atom_assemble(mesh, kind)
            ts.append(ctx.timer_stop())
            ctx.atom_free(a)
        res[name] = min(ts)
    # read once: int4 cell records, 3 coordinate arrays, v2c_ptr + v2c (4 entries per cell), row_ptr + cols; written: values
    unique = 16 * nc + 24 * nv + 4 * (nv + 1) + 4 * 4 * nc + 4 * (nv + 1) + 4 * nnz + 8 * nnz
    gathered = nv and (24 * nc * 4 / nv) * (16 + 4 * 24)       # bytes a row's lane pulls through the L1: 24 cells x (record + 4 x 3 coordinates)
    # k_assemble_p1_regular (r04: regularly numbered unit-cell lattices, unweighted kinds) reads the row pointers and writes the values - nothing else
    own = 4 * (nv + 1) + 8 * nnz
    regular = bool(ctx.mesh_lattice(mesh)[0])
    out = {"n": n, "rows": nv, "cells": nc, "nnz": nnz, "unique_bytes": unique,
           "gather_free_kernel_bytes": own, "gather_free_kernel_frac_of_8TBps": {k: own / t / 8e12 for k, t in res.items()} if regular else None,
           "seconds": res, "unique_GBps": {k: unique / t / 1e9 for k, t in res.items()},
           "frac_of_8TBps": {k: unique / t / 8e12 for k, t in res.items()},
           "cell_visits": 4 * nc, "ns_per_cell_visit_per_CU": {k: 1e9 * t / (4 * nc / 256) for k, t in res.items()},
           "bytes_gathered_through_L1_per_row": gathered}
    print(json.dumps(out), flush=True)
    ctx.mesh_free(mesh)
