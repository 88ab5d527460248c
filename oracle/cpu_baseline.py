"""Oracle (TEST INFRASTRUCTURE): the timed CPU baseline leg of bench.py.

The reference's hot loop spends its time in the spatial linear solve of every
fixed-point pass; FEniCS cannot run here, so the baseline is the
C/OpenMP restatement (oracle/c/pgd_oracle.c) of the same Jacobi-PCG on the same
system, on all host cores.  Bounded sample: the first spatial system of the run
(operator and right-hand side downloaded from the device, where they were
parity-checked against the numpy oracle at small sizes), a fixed number of PCG
iterations; one fixed-point pass costs (measured PCG iterations per pass) x
(seconds per CPU iteration) - the small 1-D solves are neglected in the CPU's favour.
"""
from __future__ import annotations

import time

from . import c_oracle


def first_spatial_system(prob, spec):
    """CSR operator and rhs of the first spatial solve, as host arrays."""
    from pgdrome_amd import fem
    V = prob.V[0]
    bcs = prob.bc
    Fs = prob.get_Fsinit(prob.V, bcs, None)
    u, v = fem.TrialFunction(V), fem.TestFunction(V)
    a = prob.lhs_fct(u, v, Fs, prob.meshes, prob.dom, prob.param, prob.prob[0], 0)
    l = prob.rhs_fct(u, v, Fs, prob.meshes, prob.dom, prob.param, prob.load, [[] for _ in prob.V], prob.prob[0], 0, 0)
    A, b = fem.assemble(a), fem.assemble(l)
    fem._apply_bcs_system(A, b, bcs[0] if bcs[0] != 0 else None)
    be = fem.get_backend()
    op = A.op()
    rp, cols = be.mesh_pattern(V.mesh().handle())
    vals = be.atom_values(op, cols.size)
    be.atom_free(op)
    return rp, cols, vals, b.host().copy()


def run(prob, spec, be, pcg_its_per_step, budget_seconds=15.0):
    rp, cols, vals, b = first_spatial_system(prob, spec)
    cores = c_oracle.num_threads()
    t0 = time.perf_counter()
    c_oracle.pcg_jacobi(rp, cols, vals, b, rtol=0.0, maxit=2)
    t_probe = (time.perf_counter() - t0) / 2.0
    sample = int(max(3, min(200, budget_seconds / max(t_probe, 1e-6))))
    t0 = time.perf_counter()
    _, it, _ = c_oracle.pcg_jacobi(rp, cols, vals, b, rtol=0.0, maxit=sample)
    t_iter = (time.perf_counter() - t0) / max(it, 1)
    # the CSR product alone, measured (not derived from the iteration time): a few launches of the C restatement's SpMV
    import numpy as np
    x = np.random.default_rng(1234).uniform(-1, 1, b.size)
    c_oracle.spmv(rp, cols, vals, x)
    reps = int(max(3, min(50, 3.0 / max(t_probe, 1e-6))))
    t0 = time.perf_counter()
    for _ in range(reps):
        c_oracle.spmv(rp, cols, vals, x)
    t_spmv = (time.perf_counter() - t0) / reps
    sec_per_step = t_iter * pcg_its_per_step
    return {
        "value": 1.0 / sec_per_step if sec_per_step > 0 else None, "unit": "fixed-point iterations/s",
        "cores": cores, "kind": "port",
        "sample": "%d Jacobi-PCG iterations of the first spatial system (n=%d, nnz=%d) on %d OpenMP threads: "
                  "%.4f s/iteration x %.1f iterations per fixed-point pass (as measured on the GPU run); "
                  "reference-algorithm CPU restatement, not FEniCS" % (it, b.size, cols.size, cores, t_iter, pcg_its_per_step),
        "seconds_per_pcg_iteration": t_iter,
        "spmv_GBps": (12.0 * cols.size + 20.0 * b.size) / t_spmv / 1e9, "spmv_seconds": t_spmv, "spmv_launches_timed": reps,
    }
