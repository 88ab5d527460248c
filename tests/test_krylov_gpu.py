"""Non-symmetric systems (VERDICT r03 "missing 3"): BiCGStab with Jacobi scaling on the CSR product (csrc/pgd_krylov.hip) where
the reference's LinearVariationalSolver + MUMPS solves whatever the callbacks produce (/root/reference/pgdrome/solver.py:627-636).
Oracle: the sparse DIRECT solve (SuperLU) of the oracle's assembly, as for the banded LU of the small time systems."""
import numpy as np
import pytest
import scipy.sparse.linalg as spla

from oracle import fem_numpy as F
from pgdrome_amd import fem, problems
from pgdrome_amd.solver import PGDProblem

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip_backend():
    from pgdrome_amd.hip_backend import HipBackend
    old = fem._backend
    be = fem.set_backend(HipBackend(0))
    fem.clear_caches()
    yield be
    fem.set_backend(old)
    fem.clear_caches()


def _boundary(coords):
    lo, hi = coords.min(axis=0), coords.max(axis=0)
    return np.where(np.any((coords <= lo + 1e-12) | (coords >= hi - 1e-12), axis=1))[0].astype(np.int32)


@pytest.mark.parametrize("name", ["rect96", "box28"])
def test_bicgstab_against_superlu(ctx, name):
    """2-D and 3-D convection-diffusion-reaction (cell Peclet numbers up to ~0.5): pgd_bicgstab_solve against SuperLU on the oracle's matrix to 1e-8, from a zero and from a non-zero start; an exhausted
    iteration budget reports its residual instead of failing; a zero right-hand side gives zero."""
    coords, cells = {"rect96": lambda: F.rectangle_mesh((0, 0), (1, 1), 95, 95), "box28": lambda: F.box_mesh((0, 0, 0), (1, 1, 1), 27, 27, 27)}[name]()
    gdim = coords.shape[1]
    n = coords.shape[0]
    beta = (30.0, -12.0, 7.0)[:gdim]
    h = ctx.mesh_upload(coords, cells)
    atoms, coefs, A = [ctx.atom_assemble(h, F.STIFF), ctx.atom_assemble(h, F.MASS)], [1.0, 3.0], None
    A = F.assemble_atom(coords, cells, F.STIFF) + 3.0 * F.assemble_atom(coords, cells, F.MASS)
    for a, b_a in enumerate(beta):
        atoms.append(ctx.atom_assemble(h, F.CONV, a, 0))
        coefs.append(b_a)
        A = A + b_a * F.assemble_atom(coords, cells, F.CONV, a, 0)
    bc = _boundary(coords)
    op = ctx.op_combine(h, atoms, coefs, bc)
    rng = np.random.default_rng(5)
    b = rng.uniform(-1, 1, n)
    b[bc] = 0.0
    Abc, bb = F.apply_dirichlet(A.tocsr(), b.copy(), bc)
    assert abs(Abc - Abc.T).max() > 1e-3 * abs(Abc).max()               # the system really is not symmetric
    ref = spla.spsolve(Abc.tocsc(), bb)
    bv = ctx.vec_from(b)
    for start in (np.zeros(n), 0.3 * ref + 0.01 * rng.uniform(-1, 1, n) * (np.abs(ref).max())):
        start = start.copy()
        start[bc] = 0.0
        xv = ctx.vec_from(start)
        it, rel = ctx.bicgstab(op, bv, xv, 1e-11, 0.0, 20000)
        x = ctx.vec_download(xv)
        assert rel <= 1e-11 and 1 <= it < 20000
        assert np.linalg.norm(x - ref) <= 1e-8 * np.linalg.norm(ref), (name, it, rel)
        assert np.linalg.norm(bb - Abc @ x) <= 1.01e-11 * np.linalg.norm(bb) + 1e-300      # the TRUE residual meets the bar
        ctx.vec_free(xv)
    xv = ctx.vec_alloc(n)
    it, rel = ctx.bicgstab(op, bv, xv, 1e-11, 0.0, 3)
    assert it == 3 and rel > 1e-11                                       # budget exhausted: reported, not raised
    zv = ctx.vec_alloc(n)
    ctx.vec_fill(xv, 0.0)
    it, rel = ctx.bicgstab(op, zv, xv, 1e-10, 0.0, 100)
    assert it == 0 and rel == 0.0 and not np.any(ctx.vec_download(xv))
    for v in (bv, xv, zv):
        ctx.vec_free(v)
    for a in [op] + atoms:
        ctx.atom_free(a)
    ctx.mesh_free(h)


@pytest.mark.parametrize("dim", [2, 3])
def test_convection_diffusion_through_the_frontend(hip_backend, dim):
    """solve(a == L) with u.dx(a) * v * dx terms on a 2-D / 3-D space: HIP engine (BiCGStab) against the oracle backend
    (SuperLU) to 1e-8; maximum_iterations / error_on_nonconvergence behave as for the PCG."""
    from oracle.backend_numpy import NumpyBackend
    P = fem.Point

    def run(backend, **prm):
        fem.set_backend(backend)
        fem.clear_caches()
        mesh = fem.RectangleMesh(P(0, 0), P(1, 1), 80, 64) if dim == 2 else fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 24, 20, 22)
        V = fem.FunctionSpace(mesh, "CG", 1)
        u, v = fem.TrialFunction(V), fem.TestFunction(V)
        f = fem.interpolate(fem.Expression("1.0 + x[0] - 0.5 * x[1]", degree=1), V)
        a = fem.inner(fem.grad(u), fem.grad(v)) * fem.dx(mesh) + fem.Constant(2.0) * u * v * fem.dx(mesh)
        for k, b_k in enumerate((25.0, -9.0, 6.0)[:dim]):
            a = a + fem.Constant(b_k) * u.dx(k) * v * fem.dx(mesh)
        w = fem.Function(V)
        info = fem.solve(a == f * v * fem.dx(mesh), w, bcs=fem.DirichletBC(V, 0, lambda x, on: on),
                         solver_parameters=dict({"linear_solver": "bicgstab", "relative_tolerance": 1e-11}, **prm))
        return np.asarray(w.compute_vertex_values()).copy(), info
    try:
        xg, info = run(hip_backend)
        assert info["method"] == "jacobi_bicgstab" and info["relres"] <= 1e-11 and info["iterations"] > 5
        with pytest.raises(RuntimeError, match="BiCGStab did not reach"):
            run(hip_backend, maximum_iterations=2)
        run(hip_backend, maximum_iterations=2, error_on_nonconvergence=False)
        xo, _ = run(NumpyBackend())
    finally:
        fem.set_backend(hip_backend)
        fem.clear_caches()
    assert np.linalg.norm(xg - xo) <= 1e-8 * np.linalg.norm(xo)


@pytest.mark.parametrize("dim", [2, 3])
def test_three_way_pgd_with_a_convective_spatial_term(hip_backend, dim):
    """problems.convection_diffusion: space x diffusivity x velocity scale; every spatial solve of the fixed-point loop is
    non-symmetric.  HIP engine against the oracle backend (direct solves): pass counts, amplitudes, every mode to 1e-6."""
    from oracle.backend_numpy import NumpyBackend
    P = fem.Point

    def run(backend):
        fem.set_backend(backend)
        fem.clear_caches()
        mesh = fem.RectangleMesh(P(0, 0), P(1, 1), 40, 36) if dim == 2 else fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 14, 12, 13)
        p = PGDProblem(**problems.convection_diffusion(mesh, PGD_nmax=4))
        p.solve_PGD(_problem="linear", settings={"linear_solver": "bicgstab", "relative_tolerance": 1e-11})
        return p, [[np.asarray(f.compute_vertex_values()).copy() for f in p.PGD_func[d]] for d in range(3)]
    try:
        i0 = fem.STATS.get("bicgstab_iterations", 0)
        pg, mg = run(hip_backend)
        assert fem.STATS.get("bicgstab_iterations", 0) > i0 + 10 * sum(int(v) for v in pg.num_fp_it)
        po, mo = run(NumpyBackend())
    finally:
        fem.set_backend(hip_backend)
        fem.clear_caches()
    assert [int(v) for v in pg.num_fp_it] == [int(v) for v in po.num_fp_it] and pg.PGD_modes == po.PGD_modes == 4
    assert pg.simulation_info.count("NOT converged") == po.simulation_info.count("NOT converged") == 0
    np.testing.assert_allclose(pg.amplitude, po.amplitude, rtol=1e-6)
    np.testing.assert_allclose(pg.alpha, po.alpha, rtol=1e-6)
    for d in range(3):
        for m in range(4):
            assert np.linalg.norm(mg[d][m] - mo[d][m]) <= 1e-6 * np.linalg.norm(mo[d][m]), (d, m)
