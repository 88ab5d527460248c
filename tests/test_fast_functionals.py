"""Functionals of plain Functions (fem._FastProd / _FastForm / _fast_scalar): the shortcut through the form algebra must give
what the general polynomial gives, bit for bit, and a plan must never outlive what it was resolved from.  The integrands are
those of the reference's callbacks (tests/integration/test_heat1D.py:55-104, test_laplace.py:73-137)."""
import gc

import numpy as np
import pytest

from oracle.backend_numpy import NumpyBackend
from pgdrome_amd import fem


@pytest.fixture(autouse=True)
def oracle_backend():
    old = fem._backend
    fem.set_backend(NumpyBackend())
    fem.clear_caches()
    yield
    fem.set_backend(old)
    fem.clear_caches()


def _fun(V, seed):
    f = fem.Function(V)
    f.vector()[:] = np.random.default_rng(seed).uniform(-1.0, 1.0, V.dim())
    return f


@pytest.mark.parametrize("mesh", ["interval", "square", "box"])
def test_fast_forms_equal_the_general_path(mesh):
    m = {"interval": lambda: fem.IntervalMesh(17, 0.0, 2.0),
         "square": lambda: fem.RectangleMesh(fem.Point(0, 0), fem.Point(1, 2), 5, 4),
         "box": lambda: fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, 1), 3, 4, 2)}[mesh]()
    V = fem.FunctionSpace(m, "CG", 1)
    F, G, W = _fun(V, 1), _fun(V, 2), _fun(V, 3)
    F.vector()[:] = F.vector()[:] * 1.0          # F and G newer than the weight W
    G.vector()[:] = G.vector()[:] * 1.0
    dx = fem.dx(m)
    one = fem.Constant(1.0)
    pairs = [(G * F * dx, (one * G) * F * dx),
             (F * F * dx, (one * F) * F * dx),
             (F * dx, (one * F) * dx),
             (G.dx(0) * F * dx, (one * G).dx(0) * F * dx),
             (fem.inner(fem.grad(G), fem.grad(F)) * dx, fem.inner(fem.grad(one * G), fem.grad(F)) * dx)]
    if mesh == "interval":
        pairs.append((W * G * F * dx, (one * W) * G * F * dx))
        pairs.append((G.dx(0) * F.dx(0) * dx, (one * G).dx(0) * F.dx(0) * dx))
    for fast, general in pairs:
        assert type(fast) is fem._FastForm and type(general) is not fem._FastForm
        for _ in range(2):                         # resolved, then planned
            fem._SCALAR_MEMO.clear()
            a = fem.assemble(fast)
            fem._SCALAR_MEMO.clear()
            b = fem.assemble(general)
            assert a == b and np.isfinite(a)
    assert fem.STATS_FAST["planned"] > 0
    # the fast form is still a Form: sums and scalings go through the general representation
    s = fem.assemble(2.0 * (G * F * dx) + F * F * dx)
    assert s == pytest.approx(2.0 * fem.assemble(G * F * dx) + fem.assemble(F * F * dx), rel=1e-15)


def test_plans_follow_the_iterates_of_a_scope_and_nothing_stale_survives():
    m = fem.IntervalMesh(11, 0.0, 1.0)
    V = fem.FunctionSpace(m, "CG", 1)
    mode, w = _fun(V, 5), _fun(V, 6)
    mode.vector()[:] = mode.vector()[:] * 1.0
    dx = fem.dx(m)
    M = None
    vals = []
    for k in range(4):                              # a new iterate object per "solve", as in solver.py:746
        it = _fun(V, 10 + k)
        with fem.functional_scope(("test", 0), [it.vector()]):
            v = fem.assemble(mode * it * dx)
            vw = fem.assemble(w * mode * it * dx)
        if M is None:
            rows = []
            for j in range(V.dim()):
                e = fem.Function(V)
                e.vector()[:] = np.eye(V.dim())[j]
                rows.append(fem.assemble(e * mode * dx))
            M = np.array(rows)                      # M[j] = e_j^T M mode
        assert v == pytest.approx(float(M @ it.vector()[:]), rel=1e-13)
        vals.append((v, vw))
    assert fem.STATS_FAST["planned"] >= 4           # the later iterates took the plan of the first
    assert len({v for v, _ in vals}) == 4
    # the weight changes: the weighted atom of the plan is stale, the value must follow
    it = _fun(V, 99)
    it.vector()[:] = it.vector()[:] * 1.0
    with fem.functional_scope(("test", 0), [it.vector()]):
        before = fem.assemble(w * mode * it * dx)
    w.vector()[:] = 2.0 * w.vector()[:]
    mode.vector()[:] = mode.vector()[:] * 1.0       # (keep the weight the oldest of the three)
    it.vector()[:] = it.vector()[:] * 1.0
    with fem.functional_scope(("test", 0), [it.vector()]):
        after = fem.assemble(w * mode * it * dx)
    assert after == pytest.approx(2.0 * before, rel=1e-13)
    # a stored mode dies and another vector may take its address: the plan's weak reference must notice
    for k in range(20):
        tmp = _fun(V, 200 + k)
        a = fem.assemble(tmp * it * dx)
        expect = float(tmp.vector()[:] @ np.array([fem.assemble(_unit(V, j) * it * dx) for j in range(V.dim())])) if k == 0 else None
        if expect is not None:
            assert a == pytest.approx(expect, rel=1e-13)
        fem._SCALAR_MEMO.clear()
        b = fem.assemble((fem.Constant(1.0) * tmp) * it * dx)
        assert a == b
        del tmp
        gc.collect()


def _unit(V, j):
    e = fem.Function(V)
    e.vector()[:] = np.eye(V.dim())[j]
    return e


def test_vector_valued_functions_keep_the_general_path():
    m = fem.RectangleMesh(fem.Point(0, 0), fem.Point(1, 1), 3, 3)
    V = fem.VectorFunctionSpace(m, "CG", 1)
    u = fem.Function(V)
    u.vector()[:] = np.arange(V.dim(), dtype=float)
    with pytest.raises(TypeError):
        u * u
    form = fem.inner(u, u) * fem.dx(m)
    assert type(form) is not fem._FastForm
    assert fem.assemble(form) > 0.0
