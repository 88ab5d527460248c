"""Checks shared by the CPU and GPU runs of the restated reference integration problems."""
import json
import os

import numpy as np

from tests import pgd_cases


def load(name):
    with open(os.path.join(pgd_cases.GOLDEN, name)) as f:
        return json.load(f)


def check_laplace(fem, PGDProblem, FD_matrices, variant, exact_counts=True):
    """test_laplace.py: exactly ONE mode (the reference's assertion :970-971) in both variants, alpha and
    the evaluated field as in the reference's run.  The "norm" stop test of this problem subtracts products
    of size alpha^2 ~ 7e6 to decide an error below 1e-5, i.e. it sits at rounding level: the number of
    passes is reproducible only with identical arithmetic (all-FEM on the oracle backend), not across
    summation orders (FD matrices, GPU kernels)."""
    from tests import laplace_problem
    ref = [r for r in load("reference_laplace.json")["runs"] if r["variant"] == variant][0]
    p = laplace_problem.run(fem, PGDProblem, FD_matrices, fd=(variant == "FD"))
    assert p.PGD_modes == ref["numModes"] == 1
    np.testing.assert_allclose(p.alpha, ref["alpha"], rtol=1e-8)
    if variant == "FEM" and exact_counts:
        assert [int(v) for v in p.num_fp_it] == ref["num_fp_it"]
    assert len(p.num_fp_it) == 1 and p.num_fp_it[0] < p.max_fp_it
    # the separated mode: factors are fixed only up to scalings that cancel in the product -> compare the field
    sol = p.return_PGD()
    u = sol.evaluate(0, [1, 2, 3], [1.5, 50, 10], 0).compute_vertex_values()
    r = np.array(ref["evaluate_y1.5_q50_u10"])
    assert np.linalg.norm(u - r) <= 1e-6 * np.linalg.norm(r)
    # against the 2-D full-order model on QUADRATIC triangles, with the reference's own bars
    # (test_laplace.py:1091-1092: mean relative error < 1e-6 all-FEM, < 2e-4 all-FD)
    errs = []
    for y, q, u0 in ((1.5, 50.0, 10.0), (0.37, 12.5, 41.0), (2.9, 33.0, 17.5)):
        fom = laplace_problem.full_order_profile(fem, y, q, u0)
        errs.append(np.linalg.norm(laplace_problem.pgd_profile(p, y, q, u0) - fom) / np.linalg.norm(fom))
    assert np.mean(errs) < (1e-6 if variant == "FEM" else 2e-4), errs
    return p


def check_solver_problem(fem, PGDProblem, problem="linear", exact_counts=True, mode_tol=1e-6):
    """test_solver_problem.py (2-D plane-strain cantilever, vector P2 in space, 4 PGD variables) on the coarser
    discretisation of the fixture: pass counts and amplitudes of the reference's own solve_PGD, the evaluated
    displacement field, the 2-D full-order model and the PGD-vs-full-order errors."""
    from tests import elastic2d_problem as ep
    ref = load("reference_solver_problem.json")["run"]
    run = [r for r in ref["runs"] if r["problem"] == problem][0]
    Vs = ep.spaces(fem, *ref["elements_x"], elems=tuple(ref["elements_extra"]))
    spec, knobs = ep.build(fem, Vs)
    p = PGDProblem(**spec)
    for k, v in knobs.items():
        setattr(p, k, v)
    settings = {"linear_solver": "mumps"} if problem == "linear" else {"relative_tolerance": 1e-8, "linear_solver": "mumps"}
    p.solve_PGD(_problem=problem, settings=settings)
    assert p.PGD_modes == run["numModes"]
    # the "norm" stop test of this problem decides at rounding level (the iterates settle to 1e-9 within five
    # passes, the test lets 17 go by): the pass counts are reproducible only with identical arithmetic
    if exact_counts:
        assert [int(v) for v in p.num_fp_it] == run["num_fp_it"]
    assert len(p.num_fp_it) == run["numModes"] and max(p.num_fp_it) < p.max_fp_it
    np.testing.assert_allclose(p.amplitude, run["amplitude"], rtol=1e-5, atol=1e-9)
    sol = p.return_PGD()
    u = sol.evaluate(0, [1, 2, 3], ref["sample"], 0)
    r = np.array(run["evaluate_vertex_values"])
    assert np.linalg.norm(u.compute_vertex_values() - r) <= mode_tol * np.linalg.norm(r)
    np.testing.assert_allclose(u(tuple(ref["point"])), run["evaluate_point"], rtol=1e-5)
    fom = ep.full_order(fem, Vs[0], *ref["sample"])
    fv = np.array(ref["fem_vertex_values"])
    assert np.linalg.norm(fom.compute_vertex_values() - fv) <= 1e-7 * np.linalg.norm(fv)
    # PGD against the full-order model: the same errors as the reference's own run has on this discretisation
    # (its bar "below the second-to-last amplitude", :786-787, holds on the fine 200 x 20 / 50 / 50 meshes of the
    # reference test, which tools/run_reference_tests.py runs; with 10 elements per parameter the parameter
    # interpolation dominates: 8.4e-4)
    pt = tuple(ref["point"])
    e_nodes = np.linalg.norm(u.compute_vertex_values() - fom.compute_vertex_values()) / np.linalg.norm(fom.compute_vertex_values())
    e_point = np.linalg.norm(u(pt) - fom(pt)) / np.linalg.norm(fom(pt))
    r_nodes = np.linalg.norm(r - fv) / np.linalg.norm(fv)
    r_point = np.linalg.norm(np.array(run["evaluate_point"]) - np.array(ref["fem_point"])) / np.linalg.norm(ref["fem_point"])
    assert abs(e_nodes - r_nodes) <= 1e-3 * r_nodes and abs(e_point - r_point) <= 1e-3 * r_point, (e_nodes, r_nodes, e_point, r_point)
    assert e_nodes < 2e-3 and e_point < 2e-3
    return p
