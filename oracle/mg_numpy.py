"""TEST INFRASTRUCTURE - CPU restatement (numpy) of the multigrid-preconditioned CG of pgdrome_amd/csrc/pgd_mg.hip.

Only tests/ may import this.  It restates, array operation by array operation, what the HIP path does when
settings["preconditioner"] names a multigrid (the reference forwards that key to its linear solver, solver.py:593-594; which
preconditioner runs behind it is PETSc's business there - there is no reference algorithm to follow, so this file pins the
build's OWN algorithm: same hierarchy, same cycle, same stop test, hence the same iteration counts as the GPU to +-1):

  * operator = one 15-point stencil c[0..7] (slot s = dx + 2 dy + 4 dz, symmetric partner at the negative offset) on an
    nx x ny x nz lattice whose hull is eliminated (identity rows);
  * coarse node k = fine node 2k, eliminated iff that fine node is; an even node count leaves the far face without a coarse
    counterpart: the last coarse node is free, beyond it zero;
  * P = P1 interpolation of the nested 6-tets-per-cube meshes (1 at the node, 1/2 along the 14 mesh edges), R = P^T,
    coarse stencil = Galerkin product on the infinite lattice (stays on the 15-point pattern);
  * V(1,1), damped Jacobi omega = 6/7, coarsest level (first with an axis under 8 nodes) 24 sweeps from zero;
  * PCG on the system with identity rows, x = b on those rows from the start, stop: |b - A x| <= rtol |b|.
"""
import numpy as np

OFFS = [(s & 1, (s >> 1) & 1, s >> 2) for s in range(8)]          # (dx, dy, dz) of slot s
OMEGA = 6.0 / 7.0
BOTTOM_SWEEPS = 24


def full27(c):
    S = np.zeros((3, 3, 3))
    for s, (dx, dy, dz) in enumerate(OFFS):
        S[1 + dz, 1 + dy, 1 + dx] = c[s]
        S[1 - dz, 1 - dy, 1 - dx] = c[s]
    return S


def slots(S):
    return np.array([S[1 + dz, 1 + dy, 1 + dx] for dx, dy, dz in OFFS])


def apply(S, x):
    """y = S * x on the lattice, zero outside the array; x indexed [z, y, x]."""
    p = np.pad(x, 1)
    y = np.zeros_like(x)
    nz, ny, nx = x.shape
    for a in range(3):
        for b in range(3):
            for c in range(3):
                if S[a, b, c] != 0.0:
                    y += S[a, b, c] * p[a:a + nz, b:b + ny, c:c + nx]
    return y


def interpolation_stencil():
    return full27([1.0] + [0.5] * 7)


def prolong(ec, fshape):
    f = np.zeros(fshape)
    f[0::2, 0::2, 0::2] = ec[:(fshape[0] + 1) // 2, :(fshape[1] + 1) // 2, :(fshape[2] + 1) // 2]
    return apply(interpolation_stencil(), f)


def restrict(r, cshape):
    t = apply(interpolation_stencil(), r)
    return t[0:2 * cshape[0]:2, 0:2 * cshape[1]:2, 0:2 * cshape[2]:2].copy()


def galerkin(S):
    """Stencil of P^T A P: unit at a coarse node of the infinite lattice -> prolong -> apply -> restrict.  Returns the 27-point
    array; entries off the 15-point pattern are rounding noise for P1 operators of this mesh (the caller checks)."""
    ec = np.zeros((7, 7, 7))
    ec[3, 3, 3] = 1.0
    y = apply(S, prolong(ec, (13, 13, 13)))
    return restrict(y, (7, 7, 7))[2:5, 2:5, 2:5].copy()


def off_pattern_leak(G):
    on = np.zeros((3, 3, 3), dtype=bool)
    for dx, dy, dz in OFFS:
        on[1 + dz, 1 + dy, 1 + dx] = True
        on[1 - dz, 1 - dy, 1 - dx] = True
    return float(np.abs(G[~on]).max()) / float(np.abs(G).max())


class Level:
    pass


def build_levels(shape, c):
    """shape = (nz, ny, nx) of the fine lattice, c = its stencil; hull eliminated on the fine lattice."""
    levels = []
    far = (True, True, True)
    S = full27(c)
    while True:
        L = Level()
        L.shape, L.S, L.w = shape, S, OMEGA / S[1, 1, 1]
        m = np.ones(shape)
        m[0, :, :] = 0.0
        m[:, 0, :] = 0.0
        m[:, :, 0] = 0.0
        for ax in range(3):
            if far[ax]:
                sl = [slice(None)] * 3
                sl[ax] = shape[ax] - 1
                m[tuple(sl)] = 0.0
        L.mask = m
        levels.append(L)
        if min(shape) < 8:
            break
        far = tuple(far[ax] and shape[ax] % 2 == 1 for ax in range(3))
        shape = tuple((s + 1) // 2 for s in shape)
        S = galerkin(S)
    return levels


def vcycle(levels, l, b):
    L = levels[l]
    if l == len(levels) - 1:
        x = L.w * b * L.mask
        for _ in range(BOTTOM_SWEEPS - 1):
            x = (x + L.w * (b - apply(L.S, x))) * L.mask
        return x
    t = (b - L.w * apply(L.S, b)) * L.mask                  # residual behind x1 = w b
    C = levels[l + 1]
    bc = restrict(t, C.shape) * C.mask
    e = vcycle(levels, l + 1, bc)
    x = (L.w * b + prolong(e, L.shape)) * L.mask
    return (x + L.w * (b - apply(L.S, x))) * L.mask


def pcg(shape, c, b, x0=None, rtol=1e-10, maxit=200, multigrid=True):
    """b, x0: arrays [z, y, x] over ALL nodes (eliminated rows: x = b).  Returns (x, iterations, relres)."""
    levels = build_levels(shape, c)
    L = levels[0]
    free = L.mask
    x = np.zeros(shape) if x0 is None else x0.copy()
    x = x * free + b * (1.0 - free)
    r = (b - apply(L.S, x * free)) * free                   # identity rows: b - x = 0
    bb = float((b * b).sum())
    M = (lambda v: vcycle(levels, 0, v)) if multigrid else (lambda v: v / L.S[1, 1, 1])
    z = M(r)
    p = z.copy()
    rz = float((r * z).sum())
    it = 0
    rr = float((r * r).sum())
    while it < maxit and rr > rtol * rtol * bb:
        q = apply(L.S, p) * free
        a = rz / float((p * q).sum())
        x += a * p
        r -= a * q
        it += 1
        rr = float((r * r).sum())
        if rr <= rtol * rtol * bb:
            break
        z = M(r)
        rz2 = float((r * z).sum())
        p = z + (rz2 / rz) * p
        rz = rz2
    return x, it, (rr / bb) ** 0.5 if bb > 0 else 0.0


# ---- the same cycle on a z-slab of a row-sharded lattice (pgd_mg_slab_* of pgdrome_amd/csrc/pgd_mg.hip) -------------------
# Level 0 stays with the rows: a rank holds the local planes [0, nzloc) = global planes [zoff, zoff + nzloc), owns [lz0, lz1) of
# them, the others are ghost planes the caller's halo exchange fills.  Levels >= 1 are whole on every rank: each rank restricts
# into the coarse planes its slab covers (coarse plane Z belongs to the owner of fine plane 2 Z), zero elsewhere; the sum over the
# ranks is the whole coarse right-hand side.  Array operation by array operation what the kernels do, so a sharded run makes the
# same iterates as `pcg` above.
class Slab:
    def __init__(self, shape_global, c, zoff, nzloc, lz0, lz1):
        self.levels = build_levels(shape_global, c)
        self.zoff, self.nzloc, self.lz0, self.lz1 = zoff, nzloc, lz0, lz1
        L = self.levels[0]
        self.mask = L.mask[zoff:zoff + nzloc]                      # free nodes of the local planes (global hull = eliminated)
        self.own = np.zeros((nzloc, 1, 1))
        self.own[lz0:lz1] = 1.0
        self.n_coarse = int(np.prod(self.levels[1].shape))

    def down(self, r):
        """t = r - w A r on the owned planes (r: local array with current ghost planes); 0 on the other planes."""
        L = self.levels[0]
        return (r - L.w * apply(L.S, r)) * self.mask * self.own

    def restrict(self, t):
        """This rank's part of the level-1 right-hand side (t: local array with current ghost planes)."""
        C = self.levels[1]
        full = apply(interpolation_stencil(), t)
        b1 = np.zeros(C.shape)
        g0, g1 = self.zoff + self.lz0, self.zoff + self.lz1
        for Z in range((g0 + 1) // 2, (g1 + 1) // 2):
            b1[Z] = full[2 * Z - self.zoff, 0:2 * C.shape[1]:2, 0:2 * C.shape[2]:2] * C.mask[Z]
        return b1

    def coarse(self, b1):
        return vcycle(self.levels, 1, b1)

    def up(self, r, x1):
        """z = t + w (r - A t) with t = w r + P x1: t on all local planes (no exchange), z on the owned ones; (z, r . z owned)."""
        L = self.levels[0]
        t = (L.w * r + prolong(x1, L.shape)[self.zoff:self.zoff + self.nzloc]) * self.mask
        z = (t + L.w * (r - apply(L.S, t))) * self.mask * self.own
        return z, float((r * z).sum())
