"""cProfile of a few bench-like passes of one BASELINE config on the GPU: where the HOST spends its time between the solves.

    python tools/host_profile.py cfg4 --modes 3
"""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem, problems
from pgdrome_amd.hip_backend import HipBackend
from pgdrome_amd.solver import PGDProblem

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--modes", type=int, default=3)
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--preconditioner", default="jacobi")
args = ap.parse_args()
be = fem.set_backend(HipBackend(0))
spec = problems.CONFIGS[args.config][0]()
spec["PGD_nmax"] = 1
p = PGDProblem(**spec)
p.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "preconditioner": args.preconditioner, "relative_tolerance": 1e-10})   # warm-up: atoms, caches
fem.clear_caches() if False else None
spec = problems.CONFIGS[args.config][0]()
spec["PGD_nmax"] = args.modes
p = PGDProblem(**spec)
be.sync()
pr = cProfile.Profile()
pr.enable()
p.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "preconditioner": args.preconditioner, "relative_tolerance": 1e-10})
be.sync()
pr.disable()
print("passes", p.fp_passes, "pcg seconds", fem.STATS["pcg_seconds"])
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(args.top)
