"""PGD.evaluate, interpolation functions and PGDErrorComputation against
values produced by the reference's own pgdrome.model on the reference's heat1D solution
(tests/golden/reference_heat1d.json, "model" block)."""
import json
import os

import numpy as np
import pytest

from oracle.backend_numpy import NumpyBackend
from pgdrome_amd import fem
from pgdrome_amd.model import PGD, PGDErrorComputation
from pgdrome_amd.solver import FD_matrices, PGDProblem
from tests import heat1d_problem, pgd_cases


@pytest.fixture(scope="module")
def solution():
    old = fem._backend
    fem.set_backend(NumpyBackend())
    fem.clear_caches()
    p = heat1d_problem.run(fem, PGDProblem, FD_matrices, fd_time=False)
    sol = p.return_PGD()
    with open(os.path.join(pgd_cases.GOLDEN, "reference_heat1d.json")) as f:
        ref = [r for r in json.load(f)["runs"] if r["variant"] == "FEM"][0]["model"]
    yield sol, ref
    fem.set_backend(old)
    fem.clear_caches()


def test_container_shape(solution):
    sol, _ = solution
    assert isinstance(sol, PGD) and sol.numModes == 20 and sol.num_pgd_var == 3
    assert [m.numNodes for m in sol.mesh] == [16, 11, 11] and sol.mesh[0].attributes[0].data[0].shape == (16, 1)
    assert sol.problem.PGD_modes == 20


def test_evaluate_matches_reference_model(solution):
    sol, ref = solution
    u = sol.evaluate(0, [1, 2], [0.9, 1.0], 0)
    r = np.array(ref["evaluate_0.9_1.0"])
    assert np.linalg.norm(u.compute_vertex_values() - r) <= 1e-6 * np.linalg.norm(r)
    assert abs(sol.evaluate_max(0, [1, 2], [0.9, 1.0], 0) - r.max()) <= 1e-6 * abs(r.max())
    assert abs(sol.evaluate_min(0, [1, 2], [0.9, 1.0], 0) - r.min()) <= 1e-6 * abs(r.min())


def test_evaluate_argument_checks(solution):
    sol, _ = solution
    with pytest.raises(ValueError):
        sol.evaluate(0, [1], [0.9, 1.0], 0)
    with pytest.raises(ValueError):
        sol.evaluate(0, [1, 2], [0.9], 0)
    with pytest.raises(ValueError):
        sol.evaluate(0, [1, 2], [0.9, 1.0], 3)
    with pytest.raises(RuntimeError):
        sol.evaluate(0, [1, 2], [5.0, 1.0], 0)          # outside the time mesh


def test_error_computation_matches_reference(solution):
    sol, ref = solution
    xs = sol.mesh[0].dataX

    def fom(smp):
        return np.cos(3.0 * xs) * smp[0] + smp[1]
    err = PGDErrorComputation(fixed_dim=[0], n_samples=5, FOM_model=fom, PGD_model=sol)
    assert err.free_dim == [1, 2]
    samples = err.sampling_LHS()
    assert np.array_equal(np.array(samples), np.array(ref["lhs_samples"]))       # same sampler, same seed: bit-exact
    errs, mean_e, max_e = err.evaluate_error()
    np.testing.assert_allclose(errs, ref["errors"], rtol=1e-6)
    assert abs(mean_e - ref["mean"]) <= 1e-6 * ref["mean"] and abs(max_e - ref["max"]) <= 1e-6 * ref["max"]
    with pytest.raises(ValueError):
        PGDErrorComputation(fixed_dim=[0], n_samples=2, FOM_model=[], PGD_model=sol).evaluate_error()


def test_interp1d_path_matches_reference(solution):
    sol, ref = solution
    for d in (1, 2):
        sol.mesh[d].attributes[0].interpolationInfo = {"name": 0, "kind": "linear"}
        sol.mesh[d].attributes[0].interpolationfct = []
    u = sol.evaluate(0, [1, 2], [0.37, 0.81], 0)
    r = np.array(ref["evaluate_interp1d_0.37_0.81"])
    assert isinstance(u, np.ndarray) and np.linalg.norm(u.reshape(-1) - r) <= 1e-6 * np.linalg.norm(r)
    with pytest.raises(ValueError):
        sol.evaluate(0, [1, 2], [3.0, 0.81], 0)        # interp1d refuses to extrapolate
