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
