"""Oracle (TEST INFRASTRUCTURE): the numpy FEM restatement as a backend of the
host-side form frontend (pgdrome_amd.fem).  It lets the ``-m "not gpu"`` tests
run the host logic (form capture, PGDProblem control flow, sharding) on CPU and
lets tests/golden/make_fixtures.py drive the reference's own solve_PGD.

It is never installed by the product: tests inject it with fem.set_backend().
Same interface as pgdrome_amd.hip_backend.HipBackend.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sps

from . import fem_numpy as F

NSLOTS = 64


class _Mesh:
    def __init__(self, coords, cells):
        self.coords, self.cells = coords, cells
        self.rp, self.cols = F.csr_pattern(coords.shape[0], cells)
        self.n = coords.shape[0]
        rows = np.repeat(np.arange(self.n), np.diff(self.rp))
        self.kl = int((rows - self.cols).max())
        self.ku = int((self.cols - rows).max())


class _BlockedMesh:
    """Vector-valued layout: dof (node i, component c) = ncomp i + c (pgd_mesh_blocked)."""

    def __init__(self, base, base_handle, ncomp):
        self.base, self.base_handle, self.ncomp = base, base_handle, ncomp
        self.coords, self.cells = base.coords, base.cells
        self.n = base.n * ncomp
        P = sps.kron(sps.csr_matrix((np.ones(base.cols.size), base.cols, base.rp), shape=(base.n, base.n)),
                     np.ones((ncomp, ncomp))).tocsr()
        P.sort_indices()
        self.rp, self.cols = P.indptr.astype(np.int32), P.indices.astype(np.int32)
        self.kl, self.ku = base.kl * ncomp + ncomp - 1, base.ku * ncomp + ncomp - 1


class NumpyBackend:
    name = "oracle-numpy"

    def __init__(self, direct_above=None):
        # systems larger than `direct_above` rows are solved by a sparse direct factorisation instead of
        # the Jacobi-PCG loop in numpy (ill-conditioned elasticity systems of the reference's
        # test_solver_problem need > 10^4 iterations); None: always the PCG restatement
        self.direct_above = direct_above
        self._obj = {}
        self._next = 1
        self.slots = np.zeros(NSLOTS)
        self._flags = [0, 0, 0]
        self.spmv_calls = 0

    def _put(self, o):
        h = self._next
        self._next += 1
        self._obj[h] = o
        return h

    # ---- meshes
    def mesh(self, coords, cells):
        return self._put(_Mesh(np.asarray(coords, dtype=np.float64), np.asarray(cells, dtype=np.int32)))

    def mesh_blocked(self, mh, ncomp):
        return self._put(_BlockedMesh(self._obj[mh], mh, int(ncomp)))

    def atom_embed(self, bmh, src, cv, cu, coef=1.0, dst=0):
        b = self._obj[bmh]
        smh, A = self._obj[src]
        assert smh == b.base_handle, "atom_embed: atom of another layout"
        E = np.zeros((b.ncomp, b.ncomp))
        E[cv, cu] = coef
        blk = sps.kron(A, E).tocsr()
        if dst:
            self._obj[dst] = (bmh, (self._obj[dst][1] + blk).tocsr())
            return dst
        return self._put((bmh, blk))

    def mesh_info(self, mh):
        m = self._obj[mh]
        return dict(nv=m.n, nc=m.cells.shape[0], nnz=int(m.rp[-1]), kl=m.kl, ku=m.ku,
                    max_row=int(np.diff(m.rp).max()))

    def mesh_pattern(self, mh):
        m = self._obj[mh]
        return m.rp, m.cols

    def mesh_free(self, mh):
        self._obj.pop(mh, None)

    # ---- vectors
    def vec_zeros(self, n):
        return self._put(np.zeros(int(n)))

    def vec_from(self, a):
        return self._put(np.array(a, dtype=np.float64))

    def vec_to_host(self, v):
        return self._obj[v].copy()

    def vec_upload(self, v, a):
        self._obj[v][:] = a

    def vec_free(self, v):
        self._obj.pop(v, None)

    def vec_copy(self, dst, src):
        self._obj[dst][:] = self._obj[src]

    def vec_scale(self, v, a):
        self._obj[v] *= a

    def vec_mul(self, y, a, x):
        self._obj[y][:] = self._obj[a] * self._obj[x]

    def vec_axpy(self, y, a, x):
        self._obj[y] += a * self._obj[x]

    def vec_fill(self, v, a):
        self._obj[v][:] = a

    def vec_lincomb(self, y, xs, coefs):
        acc = np.zeros_like(self._obj[y])
        for x, c in zip(xs, coefs):
            acc += c * self._obj[x]
        self._obj[y][:] = acc

    def vec_set(self, v, idx, vals):
        self._obj[v][np.asarray(idx)] = vals

    def vec_dot(self, x, y, lo=0, hi=-1):
        a, b = self._obj[x], self._obj[y]
        hi = a.size if hi < 0 else hi
        return float(a[lo:hi] @ b[lo:hi])

    def vec_multidot(self, x, ys, lo=0, hi=-1):
        return np.array([self.vec_dot(x, y, lo, hi) for y in ys])

    def vec_multidot_pair(self, x0, x1, ys, lo=0, hi=-1):
        return self.vec_multidot(x0, ys, lo, hi), self.vec_multidot(x1, ys, lo, hi)

    # ---- atoms and operators
    def atom_product_form(self, atom):
        return 0

    def atom(self, mh, kind, da, db, w):
        m = self._obj[mh]
        A = F.assemble_atom(m.coords, m.cells, kind, da, db, self._obj[w] if w else None)
        return self._put((mh, A))

    def atom_values(self, a, nnz):
        mh, A = self._obj[a]
        m = self._obj[mh]
        full = sps.csr_matrix((np.ones(m.cols.size), m.cols, m.rp), shape=(m.n, m.n))
        out = np.zeros(m.cols.size)
        # align A's stored entries with the mesh pattern (A may have dropped structural zeros)
        A = A.tocsr()
        A.sort_indices()
        if A.nnz == m.cols.size:
            return A.data.copy()
        pos = {}
        for i in range(m.n):
            for k in range(m.rp[i], m.rp[i + 1]):
                pos[(i, m.cols[k])] = k
        C = A.tocoo()
        for i, j, v in zip(C.row, C.col, C.data):
            out[pos[(i, j)]] = v
        return out

    def atom_free(self, a):
        self._obj.pop(a, None)

    def combine(self, mh, atoms, coefs, bc_vertices=None, reuse=0):
        m = self._obj[mh]
        A = None
        for a, c in zip(atoms, coefs):
            t = c * self._obj[a][1]
            A = t if A is None else A + t
        if bc_vertices is not None and len(bc_vertices):
            A, _ = F.apply_dirichlet(A, np.zeros(m.n), np.asarray(bc_vertices))
        if reuse:
            self._obj[reuse] = (mh, A.tocsr())
            return reuse
        return self._put((mh, A.tocsr()))

    def spmv(self, A, x, y, r0=0, r1=-1):
        M = self._obj[A][1]
        r1 = M.shape[0] if r1 < 0 else r1
        self._obj[y][r0:r1] = (M[r0:r1] @ self._obj[x])
        self.spmv_calls += 1

    def bilinear(self, A, x, y, r0=0, r1=-1):
        M = self._obj[A][1]
        r1 = M.shape[0] if r1 < 0 else r1
        return float(self._obj[x][r0:r1] @ (M[r0:r1] @ self._obj[y]))

    def bilinear_many(self, A, x, ys, r0=0, r1=-1):
        return np.array([self.bilinear(A, x, y, r0, r1) for y in ys])

    # ---- solvers
    def start_gram(self, op, vecs, b, r0=0, r1=-1):
        A = self._obj[op][1]
        n = A.shape[0]
        r1 = n if r1 < 0 else r1
        V = [self._obj[v] for v in vecs]
        full = [A @ v for v in V]
        W = [w[r0:r1] for w in full]
        G = np.array([[float(vi[r0:r1] @ wj) for wj in W] for vi in V])
        G = 0.5 * (G + G.T)
        g = np.array([float(v[r0:r1] @ self._obj[b][r0:r1]) for v in V])
        self._gram = (op, full) if (r0 == 0 and r1 == n and len(V) <= 9) else None
        return G, g

    def start_residual(self, op, coefs, b, r):
        held = getattr(self, "_gram", None)
        if held is None or held[0] != op or len(held[1]) != len(coefs):
            raise RuntimeError("start_residual: the products of this operator are not held")
        self._obj[r][:] = self._obj[b] - sum(float(c) * w for c, w in zip(coefs, held[1]))

    def pcg(self, op, b, x, rtol, atol, maxit):
        A = self._obj[op][1]
        if self.direct_above is not None and A.shape[0] > self.direct_above:
            self._obj[x][:] = F.direct_solve(A, self._obj[b])
            r = self._obj[b] - A @ self._obj[x]
            bn = np.linalg.norm(self._obj[b])
            return 1, float(np.linalg.norm(r) / bn) if bn > 0 else 0.0
        sol, it, rel = F.pcg_jacobi(self._obj[op][1], self._obj[b], self._obj[x], rtol, atol, maxit)
        self._obj[x][:] = sol
        return it, rel

    # ---- the V-cycle on a z-slab of a row-sharded lattice (oracle/mg_numpy.py::Slab restates pgd_mg_slab_*)
    def mg_slab_setup(self, op, nz_global, z_first, own0, own1):
        from . import mg_numpy as MG
        mh, A = self._obj[op]
        m = self._obj[mh]
        xs, ys = np.unique(np.round(m.coords[:, 0], 12)), np.unique(np.round(m.coords[:, 1], 12))
        nx, ny = xs.size, ys.size
        P = nx * ny
        if m.coords.shape[1] != 3 or m.n % P or own0 % P or own1 % P or min(nx, ny, nz_global) < 8:
            return 0
        nzloc, lz0, lz1 = m.n // P, own0 // P, own1 // P
        # the stencil of an interior owned row, then every owned row checked against it (eliminated rows = the global hull)
        zi = min(max(lz0, 1 if z_first == 0 else lz0), lz1 - 1)
        i0 = zi * P + (ny // 2) * nx + nx // 2
        c = np.array([A[i0, i0 + dx + nx * dy + P * dz] if i0 + dx + nx * dy + P * dz < m.n else 0.0 for dx, dy, dz in MG.OFFS])
        if not c[0] > 0.0:
            return 0
        sl = MG.Slab((nz_global, ny, nx), c, z_first, nzloc, lz0, lz1)
        probe = np.random.default_rng(3).uniform(-1, 1, (nzloc, ny, nx)) * sl.mask
        want = (MG.apply(sl.levels[0].S, probe) * sl.mask + probe * (1.0 - sl.mask)).ravel()
        got = A @ (probe.ravel() + 0.0)
        if np.abs(want - got)[own0:own1].max() > 1e-12 * np.abs(want).max():
            return 0
        diag = A.diagonal()[own0:own1].reshape(lz1 - lz0, ny, nx)
        offd = np.asarray(abs(A[own0:own1]).sum(axis=1)).ravel().reshape(lz1 - lz0, ny, nx) - np.abs(diag)
        ident = (diag == 1.0) & (offd == 0.0)
        if not np.array_equal(ident, sl.mask[lz0:lz1] == 0.0):
            return 0
        self._mg_slab = sl
        return sl.n_coarse

    def _slab3(self, v):
        sl = self._mg_slab
        return self._obj[v].reshape(sl.nzloc, sl.mask.shape[1], sl.mask.shape[2])

    def mg_slab_fix_start(self, op, b, x, own0, own1):
        elim = (self._mg_slab.mask == 0.0).ravel()
        elim[:own0] = False
        elim[own1:] = False
        self._obj[x][elim] = self._obj[b][elim]

    def mg_slab_down(self, r, t):
        self._obj[t][:] = self._mg_slab.down(self._slab3(r)).ravel()

    def mg_slab_restrict(self, t, b1):
        self._obj[b1][:] = self._mg_slab.restrict(self._slab3(t)).ravel()

    def mg_coarse(self, b1, x1):
        sl = self._mg_slab
        self._obj[x1][:] = sl.coarse(self._obj[b1].reshape(sl.levels[1].shape)).ravel()

    def mg_slab_up(self, r, x1, t, z, slot=-1):
        sl = self._mg_slab
        zz, dot = sl.up(self._slab3(r), self._obj[x1].reshape(sl.levels[1].shape))
        self._obj[z][:] = zz.ravel()
        if slot >= 0:
            self.slots[slot] = dot
            return None
        return dot

    def bicgstab(self, op, b, x, rtol, atol, maxit):
        """The oracle of the non-symmetric solve is the sparse DIRECT solve (SuperLU), like the reference's MUMPS."""
        A = self._obj[op][1]
        self._obj[x][:] = F.direct_solve(A, self._obj[b])
        r = self._obj[b] - A @ self._obj[x]
        bn = np.linalg.norm(self._obj[b])
        return 1, float(np.linalg.norm(r) / bn) if bn > 0 else 0.0

    def band_solve(self, op, b, x):
        self._obj[x][:] = F.direct_solve(self._obj[op][1], self._obj[b])

    # ---- pieces of the row-sharded PCG, eager numpy versions of the *_slot kernels
    def slots_tensor(self):
        import torch
        return torch.from_numpy(self.slots)

    def vec_tensor(self, v):
        import torch
        return torch.from_numpy(self._obj[v])

    def slots_get(self, first=0, count=NSLOTS):
        return self.slots[first:first + count].copy()

    def flags_reset(self):
        self._flags = [0, 0, 0]

    def flags(self):
        return tuple(self._flags)

    def op_diag_inv(self, op, dinv):
        self._obj[dinv][:] = 1.0 / self._obj[op][1].diagonal()

    def spmv_dot_slot(self, A, x, y, w, r0, r1, slot):
        if self._flags[0]:
            return
        if r1 == r0:
            self.slots[slot] = 0.0
            return
        self.spmv(A, x, y, r0, r1)
        self.slots[slot] = float(self._obj[w][r0:r1] @ self._obj[y][r0:r1])

    def pcg_init_slot(self, b, q, dinv, r, z, p, lo, hi, slot):
        o = self._obj
        o[r][lo:hi] = o[b][lo:hi] - o[q][lo:hi]
        o[z][lo:hi] = o[dinv][lo:hi] * o[r][lo:hi]
        o[p][lo:hi] = o[z][lo:hi]
        self.slots[slot] = float(o[r][lo:hi] @ o[z][lo:hi])
        self.slots[slot + 1] = float(o[r][lo:hi] @ o[r][lo:hi])
        self.slots[slot + 2] = float(o[b][lo:hi] @ o[b][lo:hi])

    def pcg_tol_slot(self, rtol, atol, s_rr, s_bb, s_tol2):
        tol2 = max(rtol * rtol * self.slots[s_bb], atol * atol)
        self.slots[s_tol2] = tol2
        if self.slots[s_rr] <= tol2:
            self._flags[0] = 1

    def pcg_xr_slot(self, x, r, p, q, dinv, z, lo, hi, s_rz, s_pq, s_out):
        if self._flags[0]:
            return
        o = self._obj
        alpha = self.slots[s_rz] / self.slots[s_pq]
        o[x][lo:hi] += alpha * o[p][lo:hi]
        o[r][lo:hi] -= alpha * o[q][lo:hi]
        o[z][lo:hi] = o[dinv][lo:hi] * o[r][lo:hi]
        self.slots[s_out] = float(o[r][lo:hi] @ o[z][lo:hi])
        self.slots[s_out + 1] = float(o[r][lo:hi] @ o[r][lo:hi])

    def pcg_check_slot(self, s_rr, s_tol2):
        if self._flags[0]:
            return
        self._flags[1] += 1
        self.slots[6] = self.slots[s_rr]
        if self.slots[s_rr] <= self.slots[s_tol2]:
            self._flags[0] = 1

    def pcg_p_slot(self, p, z, lo, hi, s_num, s_den):
        if self._flags[0]:
            return
        o = self._obj
        beta = self.slots[s_num] / self.slots[s_den]
        o[p][lo:hi] = o[z][lo:hi] + beta * o[p][lo:hi]

    def cg_init_slot(self, b, q, dinv, r, u, p, s, lo, hi, base):
        o = self._obj
        o[r][lo:hi] = o[b][lo:hi] - o[q][lo:hi]
        o[u][lo:hi] = o[dinv][lo:hi] * o[r][lo:hi]
        o[p][lo:hi] = 0.0
        o[s][lo:hi] = 0.0
        self.slots[base] = float(o[r][lo:hi] @ o[u][lo:hi])
        self.slots[base + 1] = float(o[r][lo:hi] @ o[r][lo:hi])
        self.slots[base + 8] = float(o[b][lo:hi] @ o[b][lo:hi])

    def cg_update_slot(self, x, r, u, w, p, s, dinv, lo, hi, base):
        if self._flags[0]:
            return
        o = self._obj
        alpha, beta = self.slots[base + 5], self.slots[base + 6]
        o[p][lo:hi] = o[u][lo:hi] + beta * o[p][lo:hi]
        o[s][lo:hi] = o[w][lo:hi] + beta * o[s][lo:hi]
        o[x][lo:hi] += alpha * o[p][lo:hi]
        o[r][lo:hi] -= alpha * o[s][lo:hi]
        o[u][lo:hi] = o[dinv][lo:hi] * o[r][lo:hi]
        self.slots[base] = float(o[r][lo:hi] @ o[u][lo:hi])
        self.slots[base + 1] = float(o[r][lo:hi] @ o[r][lo:hi])

    def cg_scalars_slot(self, base, init, rtol, atol):
        if self._flags[0]:
            return
        S = self.slots
        g, rr, d = S[base], S[base + 1], S[base + 2] + S[base + 3] + S[base + 4]
        if init:
            S[5] = max(rtol * rtol * S[base + 8], atol * atol)
        else:
            self._flags[1] += 1
        S[6] = rr
        if rr <= S[5]:
            self._flags[0] = 1
            return
        beta = 0.0 if init else g / S[base + 7]
        alpha = g / d if init else g / (d - beta * g / S[base + 5])
        S[base + 5], S[base + 6], S[base + 7] = alpha, beta, g

    def slots_set(self, vals, first=0):
        vals = np.asarray(vals, dtype=np.float64)
        self.slots[first:first + vals.size] = vals

    def sync(self):
        pass

    def prof_enable(self, on=True):
        pass

    def prof_read(self):
        return dict(launches=0, seconds=0.0, bytes=0.0)
