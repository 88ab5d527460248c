"""Oracle (TEST INFRASTRUCTURE, not product code): P1 finite elements in numpy.

Restates, on the CPU, the arithmetic that the reference hands to FEniCS
2019.1.0 for every per-dimension solve of its fixed-point loop
(/root/reference/pgdrome/solver.py:598-636, 677-716: assemble ``a`` and ``l``,
apply the Dirichlet conditions, solve) and for its norms and scalar
functionals (solver.py:342, 365-372, 443, 754, 836-842).  FEniCS is a
third-party dependency pinned at ``fenics=2019.1.0``
(/root/reference/environment.yml:8) and is not vendored under /root/reference,
so this file follows its *published* algorithm: Lagrange P1 elements on
simplices, exact integration of the polynomial integrands, global scatter of
the element tensors into a CSR matrix over the "dofs sharing a cell" pattern,
and Dirichlet rows replaced by identity.

For P1 basis functions with P1 data every integrand used on the hot path is a
polynomial of degree <= 3 on a simplex and has the closed form

    int_K  l_0^a0 ... l_D^aD  =  |K| D! a0! ... aD! / (a0 + ... + aD + D)!

(l_k barycentric coordinates), so no quadrature choice enters.

Mesh conventions ([3P-memory] for dolfin's builtin
meshes): IntervalMesh vertices ``a + i (b-a)/n``; RectangleMesh "right":
vertex ``iy (nx+1) + ix``, triangles ``(v0,v1,v3),(v0,v2,v3)``; BoxMesh: vertex
``iz (ny+1)(nx+1) + iy (nx+1) + ix`` and six tetrahedra per cube that all
contain the main diagonal v0-v7.

Parity status: the formulas are pinned by the known-answer tests in
tests/test_oracle.py (analytic element matrices, the values held by the
reference's tests).  Bit-level parity with FEniCS is UNPINNED.
"""
from __future__ import annotations

import math

import numpy as np
import scipy.sparse as sps
import scipy.sparse.linalg as spla

# atom kinds - numeric values are shared with include/pgd_amd.h
MASS, STIFF, DUDV, CONV, CONVT, WMASS, WSTIFF = range(7)
KIND_NAMES = ("mass", "stiff", "dudv", "conv", "convt", "wmass", "wstiff")


# --------------------------------------------------------------------------- meshes
def interval_mesh(n, a=0.0, b=1.0):
    """dolfin.IntervalMesh(n, a, b): n cells, n+1 vertices, vertex i at a + i h."""
    coords = (a + (b - a) * np.arange(n + 1, dtype=np.float64) / n).reshape(-1, 1)
    cells = np.stack([np.arange(n), np.arange(1, n + 1)], axis=1).astype(np.int32)
    return coords, cells


def rectangle_mesh(p0, p1, nx, ny, diagonal="right"):
    """dolfin.RectangleMesh(Point(p0), Point(p1), nx, ny, diagonal)."""
    xs = p0[0] + (p1[0] - p0[0]) * np.arange(nx + 1, dtype=np.float64) / nx
    ys = p0[1] + (p1[1] - p0[1]) * np.arange(ny + 1, dtype=np.float64) / ny
    X, Y = np.meshgrid(xs, ys, indexing="xy")  # shape (ny+1, nx+1), x fastest
    coords = np.stack([X.ravel(), Y.ravel()], axis=1)
    ix, iy = np.meshgrid(np.arange(nx), np.arange(ny), indexing="xy")
    v0 = (iy * (nx + 1) + ix).ravel()
    v1, v2, v3 = v0 + 1, v0 + (nx + 1), v0 + (nx + 1) + 1
    if diagonal == "right":
        tris = [(v0, v1, v3), (v0, v2, v3)]
    elif diagonal == "left":
        tris = [(v0, v1, v2), (v1, v2, v3)]
    else:
        raise ValueError("diagonal must be 'right' or 'left' (crossed: out of scope)")
    cells = np.empty((2 * nx * ny, 3), dtype=np.int32)
    for k, t in enumerate(tris):
        cells[k::2] = np.stack(t, axis=1)
    return coords, cells


def box_mesh(p0, p1, nx, ny, nz):
    """dolfin.BoxMesh(Point(p0), Point(p1), nx, ny, nz): 6 tets per cube."""
    xs = p0[0] + (p1[0] - p0[0]) * np.arange(nx + 1, dtype=np.float64) / nx
    ys = p0[1] + (p1[1] - p0[1]) * np.arange(ny + 1, dtype=np.float64) / ny
    zs = p0[2] + (p1[2] - p0[2]) * np.arange(nz + 1, dtype=np.float64) / nz
    Z, Y, X = np.meshgrid(zs, ys, xs, indexing="ij")  # x fastest
    coords = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    iz, iy, ix = np.meshgrid(np.arange(nz), np.arange(ny), np.arange(nx), indexing="ij")
    sx, sy = nx + 1, (nx + 1) * (ny + 1)
    v0 = (iz * sy + iy * sx + ix).ravel()
    v1, v2, v3 = v0 + 1, v0 + sx, v0 + sx + 1
    v4, v5, v6, v7 = v0 + sy, v1 + sy, v2 + sy, v3 + sy
    tets = [(v0, v1, v3, v7), (v0, v1, v7, v5), (v0, v5, v7, v4),
            (v0, v3, v2, v7), (v0, v6, v4, v7), (v0, v2, v6, v7)]
    cells = np.empty((6 * nx * ny * nz, 4), dtype=np.int32)
    for k, t in enumerate(tets):
        cells[k::6] = np.stack(t, axis=1)
    return coords, cells


# ----------------------------------------------------------------- element geometry
def _geometry(coords, cells):
    """Cell volumes |K| (nc,) and barycentric gradients g (nc, D+1, D)."""
    X = coords[cells]                      # (nc, D+1, D)
    D = X.shape[2]
    if X.shape[1] != D + 1:
        raise ValueError("only simplices with tdim == gdim are supported")
    J = X[:, 1:, :] - X[:, :1, :]          # rows: x_k - x_0, (nc, D, D)
    det = np.linalg.det(J)
    vol = np.abs(det) / math.factorial(D)
    Jinv = np.linalg.inv(J)                # columns k: grad of lambda_{k+1}
    g = np.empty((X.shape[0], D + 1, D))
    g[:, 1:, :] = np.transpose(Jinv, (0, 2, 1))
    g[:, 0, :] = -g[:, 1:, :].sum(axis=1)
    return vol, g


# 4-point Gauss-Legendre on [0, 1]: exact to degree 7 (P2 x P2 x P2 weights need 6)
_GQ = np.polynomial.legendre.leggauss(4)
GAUSS_X, GAUSS_W = 0.5 * (_GQ[0] + 1.0), 0.5 * _GQ[1]


def p2_interval_nodes(coords, cells):
    """Node coordinates and cell -> node table of the P2 space on an interval mesh: nodes ordered along
    the interval (vertex i -> 2 i, midpoint of cell i -> 2 i + 1), cell record (v0, v1, mid)."""
    x = coords[:, 0]
    nodes = np.empty(2 * x.size - 1)
    nodes[0::2] = x
    nodes[1::2] = 0.5 * (x[:-1] + x[1:])
    tab = np.stack([2 * cells[:, 0], 2 * cells[:, 1], 2 * cells[:, 0] + 1], axis=1).astype(np.int32)
    return nodes.reshape(-1, 1), tab


def element_matrices_p2_interval(coords, cells, kind, w=None):
    """3x3 local matrices of quadratic Lagrange elements on intervals by Gauss quadrature (exact for
    the polynomial integrands): N0 = (1-s)(1-2s), N1 = s(2s-1), N2 = 4s(1-s) on s in [0,1] from v0 to v1."""
    hs = coords[cells[:, 1], 0] - coords[cells[:, 0], 0]          # signed length
    s_ = GAUSS_X
    N = np.stack([(1 - s_) * (1 - 2 * s_), s_ * (2 * s_ - 1), 4 * s_ * (1 - s_)])      # (3, nq)
    dN = np.stack([4 * s_ - 3, 4 * s_ - 1, 4 - 8 * s_])                               # d/ds
    nc = cells.shape[0]
    wq = np.ones((nc, s_.size))
    if kind in (WMASS, WSTIFF):
        if w is None:
            raise ValueError("weighted atom needs nodal weights")
        wq = np.asarray(w, dtype=np.float64)[cells] @ N                               # w at the quadrature points
    jac = np.abs(hs)[:, None] * GAUSS_W[None, :] * wq                                 # (nc, nq)
    if kind in (MASS, WMASS):
        return np.einsum("cq,iq,jq->cij", jac, N, N)
    if kind in (STIFF, DUDV, WSTIFF):
        return np.einsum("cq,iq,jq->cij", jac / (hs ** 2)[:, None], dN, dN)
    if kind == CONV:
        return np.einsum("cq,iq,jq->cij", jac / hs[:, None], N, dN)
    if kind == CONVT:
        return np.einsum("cq,iq,jq->cij", jac / hs[:, None], dN, N)
    raise ValueError(f"unknown atom kind {kind}")


# ---- quadratic Lagrange elements on triangles / tetrahedra: EXACT integration by polynomial algebra in
# barycentric coordinates (no quadrature): int_K prod l_k^{a_k} = |K| D! prod a_k! / (sum a_k + D)!
P2_EDGES = {2: ((1, 2), (0, 2), (0, 1)), 3: ((2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1))}   # UFC local edges


def _poly_mul(p, q):
    out = {}
    for ea, ca in p.items():
        for eb, cb in q.items():
            e = tuple(x + y for x, y in zip(ea, eb))
            out[e] = out.get(e, 0.0) + ca * cb
    return out


def _poly_int(p, D):
    """Integral over the simplex divided by its volume."""
    tot = 0.0
    for e, c in p.items():
        num = math.factorial(D)
        for a in e:
            num *= math.factorial(a)
        tot += c * num / math.factorial(sum(e) + D)
    return tot


def _p2_basis(D):
    """P2 shape functions and their lambda-derivatives as polynomials {exponents: coef} in (l_0..l_D):
    vertex i: l_i (2 l_i - 1); edge (a, b): 4 l_a l_b; node order = vertices, then P2_EDGES[D]."""
    def mono(*idx):
        e = [0] * (D + 1)
        for i in idx:
            e[i] += 1
        return tuple(e)
    N, dN = [], []
    for i in range(D + 1):
        N.append({mono(i, i): 2.0, mono(i): -1.0})
        dN.append([({mono(i): 4.0, mono(): -1.0} if k == i else {}) for k in range(D + 1)])
    for a, b in P2_EDGES[D]:
        N.append({mono(a, b): 4.0})
        dN.append([({mono(b): 4.0} if k == a else {mono(a): 4.0} if k == b else {}) for k in range(D + 1)])
    return N, dN


_P2_REF = {}


def _p2_reference_tensors(D):
    """Per-unit-volume integrals: mass[a,b], dd[a,b,k,l] = I(d_k N_a d_l N_b), nd[a,b,l] = I(N_a d_l N_b),
    wmass[a,b,c] = I(N_a N_b N_c), wdd[a,b,c,k,l] = I(N_c d_k N_a d_l N_b)."""
    if D not in _P2_REF:
        N, dN = _p2_basis(D)
        nn = len(N)
        mass = np.zeros((nn, nn)); dd = np.zeros((nn, nn, D + 1, D + 1)); nd = np.zeros((nn, nn, D + 1))
        wmass = np.zeros((nn, nn, nn)); wdd = np.zeros((nn, nn, nn, D + 1, D + 1))
        for a in range(nn):
            for b in range(nn):
                ab = _poly_mul(N[a], N[b])
                mass[a, b] = _poly_int(ab, D)
                for c in range(nn):
                    wmass[a, b, c] = _poly_int(_poly_mul(ab, N[c]), D)
                for l in range(D + 1):
                    if dN[b][l]:
                        nd[a, b, l] = _poly_int(_poly_mul(N[a], dN[b][l]), D)
                    for k in range(D + 1):
                        if dN[a][k] and dN[b][l]:
                            pk = _poly_mul(dN[a][k], dN[b][l])
                            dd[a, b, k, l] = _poly_int(pk, D)
                            for c in range(nn):
                                wdd[a, b, c, k, l] = _poly_int(_poly_mul(pk, N[c]), D)
        _P2_REF[D] = (mass, dd, nd, wmass, wdd)
    return _P2_REF[D]


def p2_simplex_nodes(coords, cells):
    """Nodes of the P2 space on a triangle / tetrahedron mesh: the vertices, then one node per edge
    (edges numbered by sorted vertex pair); cell record = (vertices..., edge nodes in UFC local order)."""
    D = coords.shape[1]
    nv = coords.shape[0]
    pairs = np.concatenate([np.sort(cells[:, list(e)], axis=1) for e in P2_EDGES[D]], axis=0)
    uniq, inv = np.unique(pairs, axis=0, return_inverse=True)
    inv = inv.reshape(len(P2_EDGES[D]), cells.shape[0]).T
    nodes = np.concatenate([coords, 0.5 * (coords[uniq[:, 0]] + coords[uniq[:, 1]])], axis=0)
    tab = np.concatenate([cells, nv + inv], axis=1).astype(np.int32)
    return nodes, tab


def element_matrices_p2_simplex(coords, cells, kind, a=0, b=0, w=None):
    """Local 6x6 / 10x10 matrices of P2 triangles / tetrahedra (row = test i, column = trial j)."""
    D = coords.shape[1]
    vol, g = _geometry(coords, cells[:, :D + 1])          # g[c, k, :] = grad lambda_k
    mass, dd, nd, wmass, wdd = _p2_reference_tensors(D)
    gg = np.einsum("ckd,cld->ckl", g, g)
    if kind == MASS:
        return vol[:, None, None] * mass[None]
    if kind == STIFF:
        return vol[:, None, None] * np.einsum("ijkl,ckl->cij", dd, gg)
    if kind == DUDV:      # trial derivative along a, test derivative along b
        return vol[:, None, None] * np.einsum("ijkl,ck,cl->cij", dd, g[:, :, b], g[:, :, a])
    if kind == CONV:      # int u_{,a} v : N_i d_l N_j (grad l_l)_a
        return vol[:, None, None] * np.einsum("ijl,cl->cij", nd, g[:, :, a])
    if kind == CONVT:
        return vol[:, None, None] * np.einsum("jil,cl->cij", nd, g[:, :, b])
    wl = np.asarray(w, dtype=np.float64)[cells] if w is not None else None
    if kind == WMASS:
        return vol[:, None, None] * np.einsum("ijm,cm->cij", wmass, wl)
    if kind == WSTIFF:
        return vol[:, None, None] * np.einsum("ijmkl,cm,ckl->cij", wdd, wl, gg)
    raise ValueError(f"unknown atom kind {kind}")


def element_matrices(coords, cells, kind, a=0, b=0, w=None):
    """Local matrices, row = test index i, column = trial index j: P1 simplices; P2 intervals (three nodes
    per cell on a 1-D mesh); P2 triangles / tetrahedra (6 / 10 nodes per cell)."""
    if coords.shape[1] == 1 and cells.shape[1] == 3:
        return element_matrices_p2_interval(coords, cells, kind, w)
    if (coords.shape[1], cells.shape[1]) in ((2, 6), (3, 10)):
        return element_matrices_p2_simplex(coords, cells, kind, a, b, w)
    vol, g = _geometry(coords, cells)
    nc, nv, D = g.shape
    if kind == MASS:
        m = (np.ones((nv, nv)) + np.eye(nv)) / ((D + 1) * (D + 2))
        return vol[:, None, None] * m[None]
    if kind == STIFF:
        return vol[:, None, None] * np.einsum("cid,cjd->cij", g, g)
    if kind == DUDV:      # trial derivative along a, test derivative along b
        return vol[:, None, None] * np.einsum("ci,cj->cij", g[:, :, b], g[:, :, a])
    if kind == CONV:      # int u_{,a} v
        return (vol / (D + 1))[:, None, None] * np.broadcast_to(g[:, None, :, a], (nc, nv, nv))
    if kind == CONVT:     # int u v_{,b}
        return (vol / (D + 1))[:, None, None] * np.broadcast_to(g[:, :, None, b], (nc, nv, nv))
    if kind in (WMASS, WSTIFF):
        if w is None:
            raise ValueError("weighted atom needs vertex weights")
        wl = np.asarray(w, dtype=np.float64)[cells]            # (nc, D+1)
        if kind == WSTIFF:
            wbar = wl.mean(axis=1)
            return (vol * wbar)[:, None, None] * np.einsum("cid,cjd->cij", g, g)
        # int l_i l_j l_k = |K| D!/(D+3)! * c(i,j,k); c = 6 (i=j=k), 2 (two equal), 1
        c = np.ones((nv, nv, nv))
        for i in range(nv):
            for j in range(nv):
                for k in range(nv):
                    s = len({i, j, k})
                    c[i, j, k] = {1: 6.0, 2: 2.0, 3: 1.0}[s]
        fac = math.factorial(D) / math.factorial(D + 3)
        return (vol * fac)[:, None, None] * np.einsum("ijk,ck->cij", c, wl)
    raise ValueError(f"unknown atom kind {kind}")


# --------------------------------------------------------------------- CSR assembly
def csr_pattern(nv_total, cells):
    """(row_ptr int32, cols int32): all vertex pairs that share a cell, sorted."""
    nv = cells.shape[1]
    rows = np.repeat(cells, nv, axis=1).ravel()
    cols = np.tile(cells, (1, nv)).ravel()
    P = sps.coo_matrix((np.ones(rows.size, dtype=np.int8), (rows, cols)),
                       shape=(nv_total, nv_total)).tocsr()
    P.sort_indices()
    return P.indptr.astype(np.int32), P.indices.astype(np.int32)


def assemble_atom(coords, cells, kind, a=0, b=0, w=None):
    """Global CSR matrix (float64, sorted columns, structural zeros kept)."""
    n = coords.shape[0]
    nv = cells.shape[1]
    Ke = element_matrices(coords, cells, kind, a, b, w)
    rows = np.repeat(cells, nv, axis=1).ravel()
    cols = np.tile(cells, (1, nv)).ravel()
    A = sps.coo_matrix((Ke.ravel(), (rows, cols)), shape=(n, n)).tocsr()
    A.sort_indices()
    return A


def apply_dirichlet(A, b, dofs, values=None):
    """Symmetric elimination: rows/cols of ``dofs`` -> identity, b adjusted.

    The reference's LinearVariationalSolver replaces the rows only; for the
    (all homogeneous) conditions on the hot path both give the same solution
    (SURVEY.md section 2.1), the symmetric form keeps the operator SPD for PCG.
    """
    A = A.tocsr(copy=True)
    b = np.array(b, dtype=np.float64, copy=True)
    dofs = np.asarray(dofs, dtype=np.int64)
    n = A.shape[0]
    g = np.zeros(n)
    if values is not None:
        g[dofs] = values
        b -= A @ g
    mask = np.zeros(n, dtype=bool)
    mask[dofs] = True
    keep = sps.diags((~mask).astype(np.float64))
    A = keep @ A @ keep + sps.diags(mask.astype(np.float64))
    b[dofs] = g[dofs]
    A = A.tocsr()
    A.sort_indices()
    return A, b


# -------------------------------------------------------------------------- solvers
def pcg_jacobi(A, b, x0=None, rtol=1e-10, atol=0.0, maxit=10000):
    """Jacobi-preconditioned CG; stops when ||r||_2 <= max(rtol ||b||_2, atol).

    Same recurrence and stopping rule as the HIP ``pgd_pcg_solve`` so that
    iteration counts are comparable (they may differ by +-1 through summation
    order).  Returns (x, iterations, final ||r||/||b||).
    """
    A = A.tocsr()
    n = A.shape[0]
    x = np.zeros(n) if x0 is None else np.array(x0, dtype=np.float64, copy=True)
    dinv = 1.0 / A.diagonal()
    r = b - A @ x
    bnorm = float(np.sqrt(b @ b))
    tol = max(rtol * bnorm, atol)
    rr = float(r @ r)
    if np.sqrt(rr) <= tol:
        return x, 0, (np.sqrt(rr) / bnorm if bnorm > 0 else 0.0)
    z = dinv * r
    p = z.copy()
    rz = float(r @ z)
    it = 0
    while it < maxit:
        q = A @ p
        alpha = rz / float(p @ q)
        x += alpha * p
        r -= alpha * q
        z = dinv * r
        rz_new = float(r @ z)
        rr = float(r @ r)
        it += 1
        if np.sqrt(rr) <= tol:
            break
        p = z + (rz_new / rz) * p
        rz = rz_new
    return x, it, (np.sqrt(rr) / bnorm if bnorm > 0 else 0.0)


def direct_solve(A, b):
    """Sparse LU (SuperLU) - the stand-in for the reference's MUMPS solve
    (solver.py:633) and literally its FD-mode solve (solver.py:939)."""
    return spla.spsolve(A.tocsc(), b)


def bilinear(A, x, y):
    """x^T A y - the scalar functional ``assemble(x * A * y * dx)``."""
    return float(x @ (A @ y))


def l2_norm(M, x):
    """dolfin.norm(f) = sqrt(f^T M f) with the consistent mass matrix."""
    return math.sqrt(abs(bilinear(M, x, x)))


# -------------------------------------------------------------------------- sizes
def nnz_p1_box(n):
    """nnz of the 15-point P1 pattern on an n^3-vertex BoxMesh."""
    return n ** 3 + 6 * n * n * (n - 1) + 6 * n * (n - 1) ** 2 + 2 * (n - 1) ** 3


def nnz_p1_rect(n):
    return n * n + 4 * n * (n - 1) + 2 * (n - 1) ** 2


def spmv_bytes(n, nnz):
    """Algorithmic bytes of one CSR SpMV."""
    return nnz * (8 + 4) + n * (4 + 8 + 8)
