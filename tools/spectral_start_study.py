"""What do the lowest eigenvectors the right-hand sides excite buy the Jacobi-PCG of the bench workload?  

    python tools/spectral_start_study.py [n=256] [passes=25] [m=40]

Harvest (study only): Lanczos on A^-1 from the first right-hand side, every A^-1 v a solve with the multigrid-preconditioned PCG,
full re-orthogonalisation; Rayleigh-Ritz of A on the basis.  Then the cfg4 run with k of the Ritz vectors as a second-level
Galerkin correction of every spatial start (x0 = x1 + Y (Y'AY)^-1 Y'(b - A x1)) for k = 0, 4, 8, 16, 32: PCG iterations and
milliseconds per pass."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem, problems
from pgdrome_amd.hip_backend import HipBackend
from pgdrome_amd.solver import PGDProblem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 25
m = int(sys.argv[3]) if len(sys.argv) > 3 else 40
if os.environ.get("PGD_STUDY_ORACLE") == "1":        # (logic check of this script on a machine without a GPU)
    from oracle.backend_numpy import NumpyBackend
    be = fem.set_backend(NumpyBackend())
else:
    be = fem.set_backend(HipBackend(0))
P = fem.Point
mesh = fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
SET = {"linear_solver": "cg", "preconditioner": "jacobi", "relative_tolerance": 1e-10}


class Done(Exception):
    pass


def first_system(prob, spec):
    V = prob.V[0]
    bcs = prob.bc
    Fs = prob.get_Fsinit(prob.V, bcs, None)
    u, v = fem.TrialFunction(V), fem.TestFunction(V)
    a = spec["lhs_fct"](u, v, Fs, prob.meshes, prob.dom, spec["param"], spec["probs"][0], 0)
    l = spec["rhs_fct"](u, v, Fs, prob.meshes, prob.dom, spec["param"], spec["load"], [[] for _ in Fs], spec["probs"][0], 0, 0)
    A, b = fem.assemble(a), fem.assemble(l)
    fem._apply_bcs_system(A, b, bcs[0])
    return A, b


def harvest(A, b, m):
    V = b.V
    prm = fem._Params(linear_solver="cg", preconditioner="amg", relative_tolerance=1e-11)
    Q = []
    v = b.copy()
    v.scale(1.0 / v.norm("l2"))
    t0 = time.time()
    its = 0
    for j in range(m):
        Q.append(v)
        w = fem.Vector(V)
        info = fem._solve_linear(A, v, w, prm)
        its += info["iterations"]
        for _ in range(2):
            for q in Q:
                w.axpy(-w.inner(q), q)
        nw = w.norm("l2")
        if nw < 1e-12:
            break
        w.scale(1.0 / nw)
        v = w
    be.sync()
    t_basis = time.time() - t0
    # Rayleigh-Ritz of A on span Q
    op = A.op()
    AQ = []
    for q in Q:
        y = fem.Vector(V)
        be.spmv(op, q.dev(), y.dev_for_write())
        y.touched_dev()
        AQ.append(y)
    G = np.array([[qi.inner(aq) for aq in AQ] for qi in Q])
    G = 0.5 * (G + G.T)
    th, S = np.linalg.eigh(G)
    Y, res = [], []
    for i in range(len(Q)):
        y = fem.Vector(V)
        be.vec_lincomb(y.dev_for_write(), [q.dev() for q in Q], [float(c) for c in S[:, i]])
        y.touched_dev()
        r = fem.Vector(V)
        be.vec_lincomb(r.dev_for_write(), [aq.dev() for aq in AQ], [float(c) for c in S[:, i]])
        r.touched_dev()
        r.axpy(-th[i], y)
        Y.append(y)
        res.append(r.norm("l2") / th[i])
    be.atom_free(op)
    be.sync()
    return th, Y, res, t_basis, time.time() - t0, its


def run(Y, k, label):
    fem.clear_caches()
    spec = problems.reaction_diffusion(mesh, 128, PGD_nmax=50, PGD_tol=1e-12)
    prob = PGDProblem(**spec)
    Vx = spec["Vs"][0]
    Yk = Y[:k]
    real = fem._rescale_start
    extra = [0.0]

    def start(lay, op, b, x):
        real(lay, op, b, x)
        if k and lay.n == Vx.dim():
            t = time.perf_counter()
            r = fem.Vector(Vx)
            be.spmv(op, x.dev(), r.dev_for_write())
            r.touched_dev()
            r.scale(-1.0)
            r.axpy(1.0, b)
            AY = []
            for y in Yk:
                w = fem.Vector(Vx)
                be.spmv(op, y.dev(), w.dev_for_write())
                w.touched_dev()
                AY.append(w)
            G = np.array([[yi.inner(w) for w in AY] for yi in Yk])
            g = np.array([y.inner(r) for y in Yk])
            c = np.linalg.solve(0.5 * (G + G.T), g)
            for ci, y in zip(c, Yk):
                x.axpy(float(ci), y)
            be.sync()
            extra[0] += time.perf_counter() - t          # (study code: the production form costs two passes over Y)
    fem._rescale_start = start
    its, tt = [], []
    st = {"i": fem.STATS["pcg_iterations"], "t": None, "x": 0.0}

    def hook(p):
        be.sync()
        now = time.perf_counter()
        its.append(fem.STATS["pcg_iterations"] - st["i"])
        st["i"] = fem.STATS["pcg_iterations"]
        if st["t"] is not None:
            tt.append(now - st["t"] - (extra[0] - st["x"]))
        st["t"], st["x"] = now, extra[0]
        if p == passes:
            raise Done()
    prob.pass_hook = hook
    try:
        for _ in range(100):
            prob.solve_PGD(_problem="linear", settings=SET)
    except Done:
        pass
    fem._rescale_start = real
    w = 5
    out = {"label": label, "k": k, "pcg_iterations_per_pass_after_warmup": float(np.mean(its[w:])),
           "ms_per_pass_after_warmup_without_the_study_overhead": 1e3 * float(np.mean(tt[w - 1:])),
           "iterations": its, "modes": len(prob.num_fp_it), "num_fp_it": [int(v) for v in prob.num_fp_it],
           "amplitude": [float(a) for a in prob.amplitude[:4]] if getattr(prob, "amplitude", None) is not None else None}
    print(json.dumps(out), flush=True)
    return out


spec0 = problems.reaction_diffusion(mesh, 128, PGD_nmax=50, PGD_tol=1e-12)
prob0 = PGDProblem(**spec0)
A, b = first_system(prob0, spec0)
th, Y, res, t_basis, t_all, mg_its = harvest(A, b, m)
print(json.dumps({"harvest": "inverse Lanczos, %d steps, multigrid PCG solves" % m, "n": n, "seconds_basis": t_basis, "seconds_total": t_all,
                  "mg_pcg_iterations": mg_its, "ritz_values_over_lowest": [float(t / th[0]) for t in th[:32]],
                  "relative_residuals": [float(r) for r in res[:32]]}), flush=True)
del A, b, prob0
base = run(Y, 0, "no spectral vectors")
for k in (4, 8, 16, 32):
    if k <= len(Y):
        run(Y, k, "k lowest Ritz vectors")
