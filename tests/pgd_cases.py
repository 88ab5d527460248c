"""Shared by the CPU and GPU parity tests: rebuild a golden case and compare a run with it."""
import json
import os

import numpy as np

from pgdrome_amd import fem, problems
from pgdrome_amd.solver import PGDProblem

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_runs():
    with open(os.path.join(GOLDEN, "reference_runs.json")) as f:
        return json.load(f)["runs"]


def build_case(case):
    P = fem.Point
    return {
        "cfg1": lambda: problems.poisson_1d1d(32),
        "cfg2_small": lambda: problems.reaction_diffusion(fem.RectangleMesh(P(0, 0), P(1, 1), 8, 8), 9, PGD_nmax=4),
        "cfg4_small": lambda: problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 4, 4, 4), 9, PGD_nmax=4),
        "cfg3_small": lambda: problems.transient_heat(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 4, 4, 4), 9, PGD_nmax=5),
        "cfg5_small": lambda: problems.transient_heat(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 4, 4, 4), 9, 5, PGD_nmax=5),
    }[case]()


def run_case(run, settings=None):
    p = PGDProblem(**build_case(run["case"]))
    p.norm_modes, p.stop_fp = run["norm_modes"], run["stop_fp"]
    for k, v in run["knobs"].items():
        setattr(p, k, v)
    p.solve_PGD(_problem=run["problem"], **({"settings": settings} if settings else {}))
    return p


def rank_one_sum_error(p, run):
    """Relative l2 distance between the separated sums (sign / scaling of single factors cancels)."""
    # compare term by term through the full tensor on these small cases (<= 125*9*5*5 entries)
    def tensor(modes):
        tot = None
        for m in range(len(modes[0])):
            t = np.array(modes[0][m])
            for d in range(1, len(modes)):
                t = np.multiply.outer(t, np.array(modes[d][m]))
            tot = t if tot is None else tot + t
        return tot
    mine = [[f.compute_vertex_values() for f in p.PGD_func[d]] for d in range(p.num_pgd_var)]
    a, b = tensor(mine), tensor(run["modes_vertex_values"])
    return np.linalg.norm(a - b) / np.linalg.norm(b)


def check_against_golden(p, run, mode_tol=1e-6, scalar_rtol=1e-7):
    """The bar of SURVEY 8(c): iteration counts exact, amplitude/alpha tight, modes within 1e-6 rel. L2."""
    assert p.PGD_modes == run["PGD_modes"]
    assert [int(v) for v in p.num_fp_it] == run["num_fp_it"]
    np.testing.assert_allclose(p.amplitude, run["amplitude"], rtol=scalar_rtol)
    np.testing.assert_allclose(p.alpha, run["alpha"], rtol=scalar_rtol)
    if run["stop_fp"] == "norm":
        # err_fp_it is a difference of nearly equal products: compare absolutely at the FP tolerance scale
        np.testing.assert_allclose(p.err_fp_it, run["err_fp_it"], rtol=1e-3, atol=1e-9)
    assert p.simulation_info.count("NOT converged") == run["not_converged_logged"]
    # every stored mode, factor by factor (signs included: same start vector, same iteration)
    for d in range(p.num_pgd_var):
        for m in range(p.PGD_modes):
            ref = np.array(run["modes_vertex_values"][d][m])
            got = p.PGD_func[d][m].compute_vertex_values()
            assert np.linalg.norm(got - ref) <= mode_tol * np.linalg.norm(ref), (d, m)
    assert rank_one_sum_error(p, run) <= mode_tol
