"""Run one BASELINE config end to end on the GPU and print what happened.

    python tools/run_config.py cfg3 [--modes 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem, problems
from pgdrome_amd.hip_backend import HipBackend
from pgdrome_amd.solver import PGDProblem

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--modes", type=int, default=0)
ap.add_argument("--problem", default="linear")
ap.add_argument("--rtol", type=float, default=1e-10)
ap.add_argument("--preconditioner", default="jacobi", help='settings["preconditioner"]: "jacobi" (default) or a multigrid name ("amg")')
ap.add_argument("--spectral-start", default="0", help='settings["spectral_start"]: Ritz vectors in the second level of the Galerkin start (0: off; "auto")')
ap.add_argument("--trace", action="store_true", help="one line per enrichment step on stderr: seconds, passes, PCG iterations")
args = ap.parse_args()

be = fem.set_backend(HipBackend(0))
t0 = time.time()
builder, desc = problems.CONFIGS[args.config]
spec = builder()
if args.modes:
    spec["PGD_nmax"] = args.modes
p = PGDProblem(**spec)
be.sync()
t1 = time.time()
if args.trace:
    _fp = p.FP_solve

    def _timed(*a, **k):
        t, it0, s0, n0 = time.time(), fem.STATS["pcg_iterations"], fem.STATS["pcg_seconds"], len(p.num_fp_it)
        out = _fp(*a, **k)
        be.sync()
        dt, its = time.time() - t, fem.STATS["pcg_iterations"] - it0
        passes = p.num_fp_it[-1] if len(p.num_fp_it) > n0 else 0
        print("mode %d: %.2f s, %d passes, %d PCG iterations, %.1f us per iteration inside the solves, %.1f ms per pass outside them"
              % (len(p.num_fp_it), dt, passes, its, 1e6 * (fem.STATS["pcg_seconds"] - s0) / max(its, 1),
                 1e3 * (dt - (fem.STATS["pcg_seconds"] - s0)) / max(passes, 1)), file=sys.stderr, flush=True)
        return out
    p.FP_solve = _timed
settings = {"linear_solver": "cg", "preconditioner": args.preconditioner, "relative_tolerance": args.rtol}
if args.spectral_start not in ("0", ""):
    settings["spectral_start"] = args.spectral_start if args.spectral_start == "auto" else int(args.spectral_start)
p.solve_PGD(_problem=args.problem, settings=settings)
be.sync()
t2 = time.time()
print(json.dumps({
    "config": args.config, "description": desc, "dims": [V.dim() for V in spec["Vs"]],
    "setup_s": t1 - t0, "solve_s": t2 - t1, "modes": p.PGD_modes, "num_fp_it": p.num_fp_it,
    "fp_passes": p.fp_passes, "fp_it_per_s": p.fp_passes / (t2 - t1),
    "amplitude": p.amplitude, "err_fp_it": [float(e) for e in p.err_fp_it],
    "linear_solves": fem.STATS["linear_solves"], "pcg_iterations": fem.STATS["pcg_iterations"],
    "not_converged": p.simulation_info.count("NOT converged"), "pcg_seconds": fem.STATS["pcg_seconds"],
    "product_launches_by_kernel": be.ctx.kernel_counts(), "preconditioner": args.preconditioner,
    "spectral_start": args.spectral_start, "spectral_harvest_seconds_inside_solve_s": __import__("pgdrome_amd.spectral", fromlist=["STATS"]).STATS["harvest_seconds"],
    "solves_preconditioned_by_the_v_cycle": be.ctx.mg_stats()["solves"], "multigrid_fallbacks_to_jacobi": be.ctx.mg_stats()["fallbacks"]}))
