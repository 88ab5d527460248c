"""The CPU restatement of the multigrid-preconditioned CG (oracle/mg_numpy.py) against the assembled system: the design of
pgdrome_amd/csrc/pgd_mg.hip checked without a GPU; tests/test_kernels_gpu.py compares the HIP path with it."""
import numpy as np
import pytest
import scipy.sparse as sps
import scipy.sparse.linalg as spla

from oracle import fem_numpy as F
from oracle import mg_numpy as MG


def _system(npts, mu):
    co, ce = F.box_mesh((0, 0, 0), (1, 1, 1), npts - 1, npts - 1, npts - 1)
    A = (F.assemble_atom(co, ce, F.STIFF) + mu * F.assemble_atom(co, ce, F.MASS)).tocsr()
    bc = np.where(np.any((co <= 1e-12) | (co >= 1 - 1e-12), axis=1))[0]
    free = np.ones(co.shape[0], dtype=bool)
    free[bc] = False
    # identity rows and columns (no lifting: the test's b is a right-hand side of the eliminated system)
    D = sps.diags(free.astype(float))
    A_el = (D @ A @ D + sps.diags((~free).astype(float))).tocsr()
    # the stencil: the row of a node in the middle of the lattice
    n1 = npts
    mid = (n1 // 2) * n1 * n1 + (n1 // 2) * n1 + n1 // 2
    c = np.array([A[mid, mid + dx + n1 * dy + n1 * n1 * dz] for dx, dy, dz in MG.OFFS])
    return A_el, c, bc


def test_galerkin_product_of_the_p1_stencil_stays_on_its_pattern():
    _, c, _ = _system(9, 3.0)
    S = MG.full27(c)
    for _ in range(5):
        G = MG.galerkin(S)
        assert MG.off_pattern_leak(G) <= 1e-13
        assert G[1, 1, 1] > 0 and np.allclose(G, G[::-1, ::-1, ::-1], rtol=0, atol=1e-15 * abs(G[1, 1, 1]))
        S = G
    # stiffness doubles, mass grows eightfold from level to level: the Galerkin operator IS the P1 operator of the coarse mesh
    co, ce = F.box_mesh((0, 0, 0), (1, 1, 1), 4, 4, 4)          # spacing 1/4 = twice the spacing of the 9-node lattice
    A2 = (F.assemble_atom(co, ce, F.STIFF) + 3.0 * F.assemble_atom(co, ce, F.MASS)).tocsr()
    mid = 2 * 25 + 2 * 5 + 2
    c2 = np.array([A2[mid, mid + dx + 5 * dy + 25 * dz] for dx, dy, dz in MG.OFFS])
    np.testing.assert_allclose(MG.slots(MG.galerkin(MG.full27(c))), c2, rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("npts", [24, 25])
def test_multigrid_pcg_restatement_solves_the_assembled_system(npts):
    """Even and odd node counts (far faces without / with a coarse counterpart): the restatement converges in a lattice-independent
    number of iterations to the solution of the assembled system with identity rows."""
    A, c, bc = _system(npts, 3.0)
    rng = np.random.default_rng(5)
    n = npts ** 3
    b = rng.uniform(-1, 1, n)
    shape = (npts, npts, npts)
    x, it, rel = MG.pcg(shape, c, b.reshape(shape))
    assert it <= 22 and rel <= 1e-10
    xd = spla.spsolve(A.tocsc(), b)
    assert np.linalg.norm(x.ravel() - xd) <= 1e-8 * np.linalg.norm(xd)
    assert np.array_equal(x.ravel()[bc], b[bc])
    xj, itj, relj = MG.pcg(shape, c, b.reshape(shape), multigrid=False, maxit=2000)
    assert itj > 3 * it and np.linalg.norm(xj.ravel() - xd) <= 1e-8 * np.linalg.norm(xd)


def test_v_cycle_is_symmetric_positive_definite():
    _, c, _ = _system(17, 3.0)
    shape = (16, 16, 16)
    levels = MG.build_levels(shape, c)
    rng = np.random.default_rng(6)
    u, v = (rng.uniform(-1, 1, shape) * levels[0].mask for _ in range(2))
    Mu, Mv = MG.vcycle(levels, 0, u), MG.vcycle(levels, 0, v)
    assert abs((v * Mu).sum() - (u * Mv).sum()) <= 1e-12 * abs((v * Mu).sum())
    assert (u * Mu).sum() > 0 and (v * Mv).sum() > 0
