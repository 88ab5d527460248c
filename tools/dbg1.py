import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pgdrome_amd import fem, solver
from tests import pgd_cases
from pgdrome_amd.hip_backend import HipBackend
from oracle.backend_numpy import NumpyBackend
runs = pgd_cases.load_runs()
run = [r for r in runs if r["case"]=="cfg4_small" and r["problem"]=="nonlinear"][0]
orig = solver.PGDProblem._solve_dim
log = {}
def patched(self, dim, Fs, n_enr, _problem, solve_modes, settings):
    f = orig(self, dim, Fs, n_enr, _problem, solve_modes, settings)
    log.setdefault(fem.get_backend().name, []).append((dim, n_enr, fem.norm(f), f.compute_vertex_values().copy()))
    return f
solver.PGDProblem._solve_dim = patched
orig_newton = fem.NonlinearVariationalSolver.solve
def newton(self):
    out = orig_newton(self)
    print(fem.get_backend().name, "newton", self.info)
    return out
fem.NonlinearVariationalSolver.solve = newton
for be in (HipBackend(0), NumpyBackend()):
    fem.set_backend(be); fem.clear_caches()
    run2 = dict(run); run2["knobs"] = {"PGD_nmax": 1}
    p = pgd_cases.run_case(run2)
for a, b in zip(log["hip"], log["oracle-numpy"]):
    print(a[0], a[1], a[2], b[2], np.abs(a[3]-b[3]).max())
