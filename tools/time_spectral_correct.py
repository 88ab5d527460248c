"""Where the per-solve cost of the spectral start goes (256^3, k = 16): python tools/time_spectral_correct.py [n] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem, problems, spectral
from pgdrome_amd.hip_backend import HipBackend
from pgdrome_amd.solver import PGDProblem
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
k = int(sys.argv[2]) if len(sys.argv) > 2 else 16
be = fem.set_backend(HipBackend(0))
P = fem.Point
mesh = fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
spec = problems.reaction_diffusion(mesh, 128, PGD_nmax=50, PGD_tol=1e-12)
prob = PGDProblem(**spec)
A, b = bench._first_spatial_system(prob)
sp = spectral.get(fem, A, b, k, fem._Params())
print("harvest", sp.info["seconds"], "k", sp.k)
x = b.copy()
x.scale(0.5)
op = A.op()


def t(label, fn, reps=5):
    be.sync()
    fn()
    be.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    be.sync()
    print("%-40s %.3f ms" % (label, 1e3 * (time.perf_counter() - t0) / reps))


lay = A.lay
lo, hi = lay.owned_range()
r = fem.Vector(b.V)
t("correct() whole", lambda: sp.correct(fem, A, op, b, x))
t("A.merged + gram (cached)", lambda: sp.gram(fem, *A.merged()))
t("Vector alloc + free", lambda: fem.Vector(b.V).dev())
t("spmv", lambda: be.spmv(op, x.dev(), r.dev_for_write(), lo, hi))
t("scale + axpy", lambda: (r.scale(-1.0), r.axpy(1.0, b)))
t("multidot 16 (sync)", lambda: spectral._multidot(fem, lay, r, sp.Y))
out = be.vec_zeros(lay.n)
t("vec_zeros + free", lambda: be.vec_free(be.vec_zeros(lay.n)))
cs = [1.0] + [0.1] * sp.k
t("lincomb 17", lambda: be.vec_lincomb(out, [x.dev()] + [y.dev() for y in sp.Y], cs))
t("copy", lambda: be.vec_copy(x.dev_for_write(), out))
t("rescale_start (k=1)", lambda: fem._rescale_start(lay, op, b, x))
