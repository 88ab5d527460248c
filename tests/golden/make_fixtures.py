"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own
``PGDProblem.solve_PGD`` (imported from /root/reference, build container only).

    python tests/golden/make_fixtures.py

What is reference and what is not, precisely:
  * the enrichment loop, the fixed-point loop, both stop tests, the mode
    normalisation, the residual pre-check and all bookkeeping are executed by
    the unmodified /root/reference/pgdrome/solver.py;
  * ``FD_matrices`` values come straight from the reference function;
  * the FEM arithmetic underneath (what the reference delegates to FEniCS
    2019.1.0, which is not installable here) is supplied by the repository's
    form frontend running on the numpy ORACLE backend, registered under the
    module name ``dolfin`` for the duration of this script.
So these fixtures pin the control flow and bookkeeping against the reference
itself; the FEM arithmetic is pinned separately by analytic known answers
(tests/test_oracle.py).  Only numbers are written - no reference source.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

from oracle.backend_numpy import NumpyBackend   # noqa: E402
from pgdrome_amd import fem, problems           # noqa: E402

fem.set_backend(NumpyBackend())
sys.modules["dolfin"] = fem
sys.modules["h5py"] = types.ModuleType("h5py")

from pgdrome.solver import FD_matrices as ref_FD_matrices   # noqa: E402
from pgdrome.solver import PGDProblem as RefPGDProblem      # noqa: E402


def case_specs():
    P = fem.Point
    return {
        "cfg1": lambda: problems.poisson_1d1d(32),
        "cfg2_small": lambda: problems.reaction_diffusion(fem.RectangleMesh(P(0, 0), P(1, 1), 8, 8), 9, PGD_nmax=4),
        "cfg4_small": lambda: problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 4, 4, 4), 9, PGD_nmax=4),
        "cfg3_small": lambda: problems.transient_heat(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 4, 4, 4), 9, PGD_nmax=5),
        "cfg5_small": lambda: problems.transient_heat(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 4, 4, 4), 9, 5, PGD_nmax=5),
        "direct_param": lambda: problems.reaction_direct_param(33, 17),
    }


RUNS = [
    # (case, _problem, norm_modes, stop_fp, extra knobs)
    ("cfg1", "linear", "stiff", "norm", {}),
    ("cfg1", "linear", "l2", "norm", {}),
    ("cfg1", "nonlinear", "stiff", "norm", {}),
    ("cfg1", "linear", "no", "delta", {"tol_fp_it": 1e-4}),
    ("cfg1", "linear", "stiff", "norm", {"max_fp_it": 2}),        # exercises "NOT converged, continue"
    ("cfg2_small", "linear", "stiff", "norm", {}),
    ("cfg4_small", "linear", "stiff", "norm", {}),
    ("cfg4_small", "nonlinear", "l2", "norm", {"PGD_nmax": 3}),
    ("cfg3_small", "linear", "stiff", "norm", {}),
    ("cfg5_small", "linear", "stiff", "norm", {}),
    # an algebraic parameter dimension: solve_modes = ["FEM", "direct"] (solver.py:637-638, 909-925), both the scalar
    # norm_aux branch of the "stiff" normalisation (:439-441) and the "l2" one
    ("direct_param", "linear", "stiff", "norm", {}, ["FEM", "direct"]),
    ("direct_param", "nonlinear", "l2", "norm", {}, ["FEM", "direct"]),
]


def run_reference(case, prob_kind, norm_modes, stop_fp, knobs, solve_modes=None):
    import re
    spec = case_specs()[case]()
    p = RefPGDProblem(**spec)
    p.norm_modes, p.stop_fp = norm_modes, stop_fp
    for k, v in knobs.items():
        setattr(p, k, v)
    # per-solve trace of what the reference hands to the linear solver: the coefficients c_t of the
    # operator A = sum_t c_t A_t as its callbacks produced them, |b|_2 and |x|_2 of every FEM solve, in order
    trace = []
    inner = fem._solve_linear

    def traced(A, b, x, prm):
        info = inner(A, b, x, prm)
        trace.append({"n": int(A.lay.n), "coefs": [float(c) for c in A.merged()[1]],
                      "b_norm": float(b.norm("l2")), "x_norm": float(x.norm("l2"))})
        return info
    fem._solve_linear = traced
    try:
        p.solve_PGD(_problem=prob_kind, **({"solve_modes": solve_modes} if solve_modes else {}))
    finally:
        fem._solve_linear = inner
    modes = [[f.compute_vertex_values().tolist() for f in p.PGD_func[d]] for d in range(p.num_pgd_var)]
    err = [np.asarray(e, dtype=float).tolist() for e in p.err_fp_it]
    res_error = [float(v) for v in re.findall(r"-- residuum norm: (\S+) --", p.simulation_info)]
    return {
        "case": case, "problem": prob_kind, "norm_modes": norm_modes, "stop_fp": stop_fp, "knobs": knobs,
        "solve_modes": solve_modes, "res_error": res_error, "solve_trace": trace,
        "stopped_on_residual": p.simulation_info.count("residuum norm smaller 1e-10"),
        "dims": [V.dim() for V in spec["Vs"]],
        "PGD_modes": int(p.PGD_modes), "num_fp_it": [int(v) for v in p.num_fp_it], "err_fp_it": err,
        "amplitude": [float(v) for v in p.amplitude], "alpha": [float(v) for v in p.alpha],
        "modes_vertex_values": modes,
        "not_converged_logged": p.simulation_info.count("NOT converged"),
    }


def fd_fixture():
    out = []
    rng = np.random.default_rng(42)
    for x in (np.linspace(0, 1, 5), np.linspace(-1, 3, 2), np.sort(rng.uniform(0, 2, 9))):
        M, D2, D1 = ref_FD_matrices(x)
        out.append({"x": x.tolist(), "M": M.toarray().tolist(), "D2": D2.toarray().tolist(),
                    "D1_up": D1.toarray().tolist()})
    return out


def heat1d_fixture():
    """The reference's own integration test module (tests/integration/test_heat1D.py), loaded as
    is: its create_meshes / create_bc / problem_assemble_* / create_PGD run the reference PGDProblem
    (heating case of test_heating; the parameter values below are that test's setUp data)."""
    import contextlib
    import importlib.util
    import io
    import logging
    spec = importlib.util.spec_from_file_location("ref_test_heat1D", "/root/reference/tests/integration/test_heat1D.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = []
    for typ in ("FEM", "FDtime"):
        param = {"rho": 1, "cp": 1, "k": 0.5, "Tamb": 25, "Q": 1, "af": 0.2, "ar": 0.2, "xc": 0.5, "lx": 1, "lt": 1}
        ranges = [[0.0, 1.0], [0.0, 1.0], [0.5, 1.0]]
        ff = 6 * np.sqrt(3) / ((param["af"] + param["ar"]) * param["af"] * param["af"] * np.pi ** 1.5)
        q = fem.Expression("ff* exp(-3*(pow(x[0]-xc,2)/pow(af,2)))", degree=4, ff=ff, af=param["af"],
                           ar=param["ar"], xc=param["xc"])
        param["IC_t"] = fem.Expression("Tamb", degree=1, Tamb=param["Tamb"])
        param["IC_x"] = fem.Expression("1.0", degree=1)
        param["IC_q"] = fem.Expression("1.0", degree=1)
        meshes, vs = mod.create_meshes([15, 10, 10], [1, 1, 1], ranges)
        logging.disable(logging.CRITICAL)
        with contextlib.redirect_stdout(io.StringIO()):
            sol, param = mod.create_PGD(param=param, vs=vs, q=q, _type=typ)
        logging.disable(logging.NOTSET)
        p = sol.problem
        out.append({"variant": typ, "PGD_modes": int(p.PGD_modes), "num_fp_it": [int(v) for v in p.num_fp_it],
                    "err_fp_it": [float(e) for e in p.err_fp_it], "amplitude": [float(a) for a in p.amplitude],
                    "alpha": [float(a) for a in p.alpha],
                    "not_converged_logged": p.simulation_info.count("NOT converged"),
                    "modes_vertex_values": [[f.compute_vertex_values().tolist() for f in p.PGD_func[d]] for d in range(3)]})
        if typ == "FEM":
            # online evaluation and error computation of the reference's own model.PGD on this solution
            from pgdrome.model import PGDErrorComputation as RefErr
            ev = sol.evaluate(0, [1, 2], [0.9, 1.0], 0)
            xs = sol.mesh[0].dataX if hasattr(sol.mesh[0], "dataX") else vs[0].mesh().coordinates()[:, 0]
            xs = np.asarray(xs, dtype=float).reshape(-1)

            def fom(smp):
                return np.cos(3.0 * xs) * smp[0] + smp[1]
            err = RefErr(fixed_dim=[0], n_samples=5, FOM_model=fom, PGD_model=sol)
            samples = err.sampling_LHS()
            errs, mean_e, max_e = err.evaluate_error()
            for d in (1, 2):
                sol.mesh[d].attributes[0].interpolationInfo = {"name": 0, "kind": "linear"}
                sol.mesh[d].attributes[0].interpolationfct = []
            ev_interp = sol.evaluate(0, [1, 2], [0.37, 0.81], 0)
            out[-1]["model"] = {"evaluate_0.9_1.0": ev.compute_vertex_values().tolist(),
                                "evaluate_interp1d_0.37_0.81": np.asarray(ev_interp).reshape(-1).tolist(),
                                "lhs_samples": samples, "errors": errs.tolist(), "mean": float(mean_e), "max": float(max_e)}
        print("heat1D", typ, "->", p.PGD_modes, "modes, fp", p.num_fp_it)
    return out


def elastic_fixture():
    """The reference's tests/integration/test_elastic.py, loaded as is: P2 spaces, default Newton
    solver, its analytic full-order model and its two PGDErrorComputation checks."""
    import contextlib
    import importlib.util
    import io
    import logging
    from pgdrome.model import PGDErrorComputation as RefErr
    spec = importlib.util.spec_from_file_location("ref_test_elastic", "/root/reference/tests/integration/test_elastic.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    logging.disable(logging.CRITICAL)
    with contextlib.redirect_stdout(io.StringIO()):
        meshes, vs = mod.create_meshes([113, 2, 100], [2, 2, 2], [[0, 1], [-1.0, 3.0], [0.2, 2.0]])
        sol = mod.main(vs, writeFlag=False, name="fixture")
    logging.disable(logging.NOTSET)
    p = sol.problem
    err = RefErr(fixed_dim=[0], n_samples=10, FOM_model=mod.FOM_solution(meshes=meshes, x=meshes[0].coordinates()),
                 PGD_model=sol)
    e, mean_e, max_e = err.evaluate_error()
    err3 = RefErr(fixed_dim=[0], FOM_model=mod.FOM_solution(meshes=meshes, x=np.array([0.5])), PGD_model=sol,
                  data_test=[[2.0, 1.5], [1.0, 1.0]], fixed_var=[0.5])
    e3, mean3, max3 = err3.evaluate_error()
    print("elastic ->", p.PGD_modes, "modes, fp", p.num_fp_it, "mean err", mean_e, "point err", mean3)
    return {"PGD_modes": int(p.PGD_modes), "num_fp_it": [int(v) for v in p.num_fp_it],
            "amplitude": [float(a) for a in p.amplitude], "alpha": [float(a) for a in p.alpha],
            "dims": [V.dim() for V in vs], "errors": e.tolist(), "mean_error": float(mean_e), "max_error": float(max_e),
            "point_errors": e3.tolist(), "mean_point_error": float(mean3),
            "modes_vertex_values": [[f.compute_vertex_values().tolist() for f in p.PGD_func[d]] for d in range(3)]}


def laplace_fixture():
    """The reference's tests/integration/test_laplace.py, loaded as is: 4-way (x, y, q, u0) problem, all-FEM
    and all-FD variants through its own create_PGD (setUp data: k 0.5, lx = ly = 3, elements 60/40/200/80)."""
    import contextlib
    import importlib.util
    import io
    import logging
    import warnings
    spec = importlib.util.spec_from_file_location("ref_test_laplace", "/root/reference/tests/integration/test_laplace.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = []
    logging.disable(logging.CRITICAL)
    warnings.filterwarnings("ignore")
    with contextlib.redirect_stdout(io.StringIO()):
        meshes, vs = mod.create_meshes([60, 40, 200, 80], [1, 1, 1, 1], [[0.0, 3.0], [0.0, 3.0], [0.0, 50.0], [10.0, 50.0]])
    for typ in ("FEM", "FD"):
        with contextlib.redirect_stdout(io.StringIO()):
            sol, prm = mod.create_PGD(param={"k": 0.5, "lx": 3, "ly": 3}, vs=vs, _type=typ)
        p = sol.problem
        u = sol.evaluate(0, [1, 2, 3], [1.5, 50, 10], 0)
        out.append({"variant": typ, "numModes": int(sol.numModes), "num_fp_it": [int(v) for v in p.num_fp_it],
                    "alpha": [float(a) for a in p.alpha], "amplitude": [float(a) for a in p.amplitude],
                    "evaluate_y1.5_q50_u10": u.compute_vertex_values().tolist(),
                    "modes_vertex_values": [[f.compute_vertex_values().tolist() for f in p.PGD_func[d]] for d in range(4)]})
        print("laplace", typ, "->", sol.numModes, "modes, fp", p.num_fp_it, "alpha", p.alpha)
    logging.disable(logging.NOTSET)
    return out


def solver_problem_fixture():
    """The reference's tests/integration/test_solver_problem.py, loaded as is, on a COARSER discretisation
    (its own functions take the element counts: 40 x 4 "crossed" cells with vector P2 in space instead of
    200 x 20; 2 / 10 / 10 elements for load factor, Young's-modulus factor and Poisson ratio): 2-D plane-strain
    cantilever, PGD variables (x, p, E, nu), "linear" and "nonlinear" solve_PGD, its FEM_reference."""
    import contextlib
    import importlib.util
    import io
    import logging
    import warnings
    spec = importlib.util.spec_from_file_location("ref_test_solver_problem",
                                                  "/root/reference/tests/integration/test_solver_problem.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    logging.disable(logging.CRITICAL)
    warnings.filterwarnings("ignore")
    ranges = [[0.0, 2.0], [0.5, 1.5], [0.1, 0.4]]
    sample, point = [1.5, 0.75, 0.2], (500.0, 50.0)
    out = {"elements_x": [40, 4], "elements_extra": [2, 10, 10], "sample": sample, "point": list(point), "runs": []}
    with contextlib.redirect_stdout(io.StringIO()):
        _, v_x = mod.create_meshX([40, 4], 2)
        _, v_e = mod.create_meshesExtra([2, 10, 10], [1, 1, 1], ranges)
        for problem, settings in (("linear", {"linear_solver": "mumps"}),
                                  ("nonlinear", {"relative_tolerance": 1e-8, "linear_solver": "mumps"})):
            params = {"E_0": 30000, "g1": fem.Constant((0.0, -0.5)), "g2": fem.Constant((0.0, -1.5))}
            p, sol = mod.main_PGD([v_x] + v_e, params, problem=problem, settings=settings)
            u = sol.evaluate(0, [1, 2, 3], sample, 0)
            out["runs"].append({"problem": problem, "numModes": int(sol.numModes),
                                "num_fp_it": [int(v) for v in p.num_fp_it],
                                "amplitude": [float(a) for a in p.amplitude], "alpha": [float(a) for a in p.alpha],
                                "evaluate_vertex_values": u.compute_vertex_values().tolist(),
                                "evaluate_point": [float(v) for v in u(point)]})
            print("solver_problem", problem, "->", sol.numModes, "modes, fp", p.num_fp_it, "amp", p.amplitude, file=sys.stderr)
        ref = mod.FEM_reference(v_x, params)(sample)
    out["fem_vertex_values"] = ref.compute_vertex_values().tolist()
    out["fem_point"] = [float(v) for v in ref(point)]
    logging.disable(logging.NOTSET)
    return out


def main():
    only = set(sys.argv[1:])
    if not only or "solver_problem" in only:
        with open(os.path.join(HERE, "reference_solver_problem.json"), "w") as f:
            json.dump({"generator": "tests/golden/make_fixtures.py",
                       "source": "reference tests/integration/test_solver_problem.py functions run unchanged on a "
                                 "coarser discretisation (main_PGD linear + nonlinear, FEM_reference)",
                       "arithmetic": "oracle numpy backend (FEniCS absent)", "run": solver_problem_fixture()}, f)
    if only and only != {"all"}:
        return
    with open(os.path.join(HERE, "reference_laplace.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fixtures.py",
                   "source": "reference tests/integration/test_laplace.py run unchanged (create_PGD, FEM and FD)",
                   "arithmetic": "oracle numpy backend (FEniCS absent)", "runs": laplace_fixture()}, f)
    with open(os.path.join(HERE, "reference_elastic.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fixtures.py",
                   "source": "reference tests/integration/test_elastic.py run unchanged (main + its error checks)",
                   "arithmetic": "oracle numpy backend (FEniCS absent)", "run": elastic_fixture()}, f)
    with open(os.path.join(HERE, "reference_heat1d.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fixtures.py",
                   "source": "reference tests/integration/test_heat1D.py run unchanged (create_PGD, heating case)",
                   "arithmetic": "oracle numpy backend (FEniCS absent)", "runs": heat1d_fixture()}, f)
    runs = [run_reference(*r) for r in RUNS]
    with open(os.path.join(HERE, "reference_runs.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fixtures.py",
                   "reference": "BAMresearch/PGDrome @ 2024_10_08, pgdrome/solver.py solve_PGD/FP_solve",
                   "arithmetic": "oracle numpy backend (FEniCS absent)", "runs": runs}, f)
    with open(os.path.join(HERE, "fd_matrices.json"), "w") as f:
        json.dump({"generator": "tests/golden/make_fixtures.py", "source": "pgdrome/solver.py:947-988 FD_matrices",
                   "cases": fd_fixture()}, f)
    for r in runs:
        print(r["case"], r["problem"], r["norm_modes"], r["stop_fp"], r["knobs"], "->", r["PGD_modes"], "modes, fp",
              r["num_fp_it"], "amp", ["%.4e" % a for a in r["amplitude"]])


if __name__ == "__main__":
    main()
