"""CPU prototype (oracle backend, scipy): what do accurate lowest eigenvectors in the Galerkin start buy the Jacobi-PCG of a
cfg4-like run?   python tools/deflation_study_cpu.py [n=40] [passes=24]"""
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.backend_numpy import NumpyBackend
from oracle import fem_numpy as F
from pgdrome_amd import fem, problems
from pgdrome_amd.solver import PGDProblem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 24
be = fem.set_backend(NumpyBackend())
P = fem.Point
fem.START_SPACE_MAX = 64


class Done(Exception):
    pass


def eigvecs(mesh, k, cbar, which="A"):
    c, e = mesh.coordinates(), mesh.cells()
    K = F.assemble_atom(c, e, F.STIFF).tocsr()
    M = F.assemble_atom(c, e, F.MASS).tocsr()
    free = np.where(~np.any((c <= 1e-12) | (c >= 1 - 1e-12), axis=1))[0]
    Kf, Mf = K[free][:, free], M[free][:, free]
    if which == "A":
        w, V = spla.eigsh((Kf + cbar * Mf).tocsc(), k=k, sigma=0, which="LM")
    else:
        w, V = spla.eigsh(Kf.tocsc(), k=k, M=Mf.tocsc(), sigma=0, which="LM")
    out = np.zeros((c.shape[0], k))
    out[free] = V
    return w, out


def run(k, which="A", cbar=5.0, log=False):
    fem.clear_caches()
    mesh = fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
    spec = problems.reaction_diffusion(mesh, 33, PGD_nmax=50, PGD_tol=1e-12)
    prob = PGDProblem(**spec)
    V = spec["Vs"][0]
    defl = []
    if k:
        w, E = eigvecs(mesh, k, cbar, which)
        for j in range(k):
            f = fem.Function(V)
            f.vector()[:] = E[:, j]
            defl.append(f.vector())
    real = fem._rescale_start
    per = []

    def start(lay, op, b, x):
        if lay.n == V.dim():
            x._start_space = list(getattr(x, "_start_space", ())) + defl
        real(lay, op, b, x)
        if lay.n == V.dim() and log:
            A = be._obj[op][1]
            r = be._obj[b.dev()] - A @ be._obj[x.dev()]
            per.append(float(np.linalg.norm(r) / np.linalg.norm(be._obj[b.dev()])))
    fem._rescale_start = start
    its = []
    i0 = [0]

    def hook(p):
        its.append(fem.STATS["pcg_iterations"] - i0[0])
        i0[0] = fem.STATS["pcg_iterations"]
        if p == passes:
            raise Done()
    prob.pass_hook = hook
    i0[0] = fem.STATS["pcg_iterations"]
    t = time.time()
    try:
        for _ in range(10):
            prob.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "relative_tolerance": 1e-10})
    except Done:
        pass
    fem._rescale_start = real
    return its, per, time.time() - t


for which in ("A", "KM"):
    for k in (0, 1, 2, 4, 8, 16):
        if k == 0 and which != "A":
            continue
        its, per, dt = run(k, which, log=True)
        print("%s k=%2d  its/pass after 5 warm: %.1f   all: %s" % (which, k, np.mean(its[5:]), its), flush=True)
        if per:
            print("      start residuals:", " ".join("%.1e" % v for v in per), flush=True)
