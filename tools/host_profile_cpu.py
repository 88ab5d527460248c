"""cProfile of the frontend's HOST work at many modes, on the CPU (oracle backend injected, tiny meshes): the Python between
the solves is the same on the GPU, the device work is not measured here.

    python tools/host_profile_cpu.py [--modes 30] [--top 40]
"""
import argparse
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem, problems
from pgdrome_amd.solver import PGDProblem
from oracle.backend_numpy import NumpyBackend

ap = argparse.ArgumentParser()
ap.add_argument("--modes", type=int, default=30)
ap.add_argument("--top", type=int, default=40)
ap.add_argument("--sort", default="tottime")
ap.add_argument("--no-profile", action="store_true", help="wall time only")
args = ap.parse_args()
fem.set_backend(NumpyBackend())


def build(nmax):
    spec = problems.transient_heat(fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, 1), 4, 4, 4), 9, 5, PGD_nmax=nmax, PGD_tol=1e-30)
    return PGDProblem(**spec)


build(1).solve_PGD(_problem="linear")
p = build(args.modes)
pr = cProfile.Profile()
t0 = time.perf_counter()
if not args.no_profile:
    pr.enable()
p.solve_PGD(_problem="linear")
pr.disable()
dt = time.perf_counter() - t0
print("modes %d passes %d: %.2f s = %.2f ms per pass (profiler %s)" % (p.PGD_modes, p.fp_passes, dt, 1e3 * dt / max(p.fp_passes, 1),
                                                                        "off" if args.no_profile else "on"))
print("prefetch", fem.STATS_PREFETCH, "fast plans", getattr(fem, "STATS_FAST", None))
if not args.no_profile:
    pstats.Stats(pr).sort_stats(args.sort).print_stats(args.top)
