"""Per PCG solve of a BASELINE config: iterations, relative residual of the start vector, size of the Galerkin start space.

    python tools/solve_trace.py cfg4 --modes 4
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem, problems
from pgdrome_amd.hip_backend import HipBackend
from pgdrome_amd.solver import PGDProblem

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--modes", type=int, default=4)
args = ap.parse_args()
be = fem.set_backend(HipBackend(0))
spec = problems.CONFIGS[args.config][0]()
spec["PGD_nmax"] = args.modes
p = PGDProblem(**spec)
_pcg = be.pcg
_gram = be.start_gram
state = {"k": 0}


def gram(op, vecs, b, lo, hi):
    state["k"] = len(vecs)
    return _gram(op, vecs, b, lo, hi)


def pcg(op, b, x, rtol, atol, maxit):
    # relative residual of the start: one CSR-free product through the operator's own form
    it, rel = _pcg(op, b, x, rtol, atol, maxit)
    print("mode %2d pass %2d: %4d iterations, start space %d vectors" % (len(p.num_fp_it) + 1, state.get("pass", 0), it, state["k"]), flush=True)
    return it, rel


be.pcg, be.start_gram = pcg, gram
p.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "preconditioner": "jacobi", "relative_tolerance": 1e-10})
print("passes", p.fp_passes, "iterations", fem.STATS["pcg_iterations"], "num_fp_it", p.num_fp_it)
