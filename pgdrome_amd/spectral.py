"""Spectral start space of the spatial Jacobi-PCG solves (round 4; VERDICT r03 "next" #2).

Every spatial system of a PGD run is  A = sum_t c_t A_t  over the SAME few atoms (K + c M, coefficients from the other
dimensions' functionals) with the same Dirichlet set, so all of them share - to O(h^2) - the eigenvectors at the low end of the
spectrum, which is where Jacobi-PCG spends its iterations.  Harvested ONCE per (space, Dirichlet set) they take the components
along them out of every later start residual:

    harvest   Lanczos on A^-1 started from the first right-hand side (so the Krylov space holds the eigenvectors the run's
              right-hand sides excite, not the whole low end), m = 2.5 k steps, every A^-1 v a solve of the engine's own PCG
              under the multigrid preconditioner (pgd_mg.hip; about 20 ms at 256^3), full re-orthogonalisation;
              Rayleigh-Ritz of A on the basis; the k lowest Ritz vectors whose residual passed are kept (k x n doubles of HBM).
    per solve second level of the Galerkin start (fem._rescale_start is the first):  x0 = x1 + Y (Y'AY)^-1 Y'(b - A x1).
              Y'AY = sum_t c_t (Y'A_t Y) from Gram matrices kept per atom (Y vanishes on the eliminated nodes, so the
              operator's Dirichlet rows play no part): one product, one multi-dot over Y, one linear combination over Y -
              about 1.5 ms at 256^3 with k = 16 (2.7 ms with 32) against the 20 - 37 ms the iterations it saves would take.

Measured (tools/spectral_start_study.py, bench.py --spectral-start k, profiles/r04_spectral_start_study_*.jsonl; cfg4, 256^3): PCG
iterations per pass 551 -> 425 / 338 / 303 / 298 for k = 16 / 32 / 48 / 64 (2.5 k Lanczos steps each: the accuracy of the Ritz pairs
matters - 32 vectors out of 40 steps were WORSE than 16), 9.7 -> 12.2 / 14.6 / 15.7 / 15.6 passes/s.  Beyond that the right-hand sides
of later modes are spread over the whole spectrum and the first residual stays at 1e-3 .. 1e-5 of |b| - HISTORY.md (r04) has the tables.

Opt-in: ``settings["spectral_start"] = k`` (forwarded like every other key of the reference's ``settings``,
solver.py:593-594) or PGD_SPECTRAL_START=k - or "auto": 32 vectors, harvested once a space has seen 48 large SPD solves (a run that
short never pays the harvest back); off by default, so that a run without it is the run of the earlier rounds bit for
bit.  The harvest needs the multigrid preconditioner to be cheap; where the operator has no structure for it (or the rows are
sharded) the request is dropped with a log line unless PGD_SPECTRAL_ANY_SOLVER=1 accepts Jacobi-PCG solves (m cold solves).
The reference has no counterpart (direct solves, solver.py:627-636); the iterates equal the ones without it to the solver's
tolerance."""
from __future__ import annotations

import logging
import os
import time
import weakref

import numpy as np

LOG = logging.getLogger("pgdrome_amd.spectral")
MIN_ROWS = int(os.environ.get("PGD_SPECTRAL_MIN_ROWS", "200000"))         # smaller systems: the solves are launch-bound, nothing to win
RESIDUAL_BAR = 1e-2                                                         # relative residual |A y - theta y| / (theta |y|) of a kept Ritz pair
STATS = {"harvests": 0, "harvest_seconds": 0.0, "corrections": 0, "dropped_requests": 0}
_SPACES = {}      # (id(layout), Dirichlet signature) -> SpectralStart | None (None: asked for, not available)


GS_PASSES = int(os.environ.get("PGD_SPECTRAL_GS_PASSES", "2"))            # re-orthogonalisation passes per Lanczos step of the harvest
AUTO_K = 32                 # settings["spectral_start"] = "auto": this many vectors ...
AUTO_AFTER = 48             # ... harvested when a space has seen this many large SPD solves (the harvest pays back after ~65 passes: short runs never pay)
_AUTO_SOLVES = {}


def requested(prm):
    """k of settings["spectral_start"] (an unset key of the nested parameter dictionary is an empty dictionary), else the default;
    "auto": -AUTO_K (get() then waits for AUTO_AFTER solves on the space before it harvests)."""
    v = prm.get("spectral_start", None) if hasattr(prm, "get") else None
    if v is None or isinstance(v, dict):
        v = os.environ.get("PGD_SPECTRAL_START", "0") or "0"
    if isinstance(v, str) and v.strip().lower() == "auto":
        return -AUTO_K
    try:
        return max(int(v), 0)
    except (TypeError, ValueError):
        return 0


def clear():
    _SPACES.clear()
    _AUTO_SOLVES.clear()


class SpectralStart:
    def __init__(self, lay, Y, theta, residuals, info):
        self.lay = weakref.ref(lay)
        self.Y, self.theta, self.residuals, self.info = Y, theta, residuals, info
        self.k = len(Y)
        self._gram = {}           # atom handle -> Y' A_t Y

    def gram(self, fem, handles, coefs):
        be = fem.get_backend()
        lay = self.lay()
        lo, hi = lay.owned_range()
        G = np.zeros((self.k, self.k))
        for h, c in zip(handles, coefs):
            Gt = self._gram.get(h)
            if Gt is None:
                Gt = np.zeros((self.k, self.k))
                w = fem.Vector(self.Y[0].V)
                for j, y in enumerate(self.Y):
                    fem._halo(lay, y)
                    be.spmv(h, y.dev(), w.dev_for_write(), lo, hi)
                    w.touched_dev()
                    Gt[:, j] = _multidot(fem, lay, w, self.Y)
                Gt = 0.5 * (Gt + Gt.T)
                self._gram[h] = Gt
            G += float(c) * Gt
        return G

    def correct(self, fem, A, op, b, x, start_coefs=None):
        """x += Y (Y'AY)^-1 Y'(b - A x): the start residual loses its components along the kept Ritz vectors.
        `start_coefs`: the coefficients fem._rescale_start has just given x in its start vectors - the library still holds
        their products with A (pgd_start_gram), so b - A x is one linear combination instead of a product with an operator
        that has only its diagonal form yet (a fresh operator's plain product forms the CSR values first: 3 ms at 256^3)."""
        be = fem.get_backend()
        lay = A.lay
        lo, hi = lay.owned_range()
        handles, coefs = A.merged()
        G = self.gram(fem, handles, coefs)
        if x._zero:
            r = b
        else:
            r = fem.Vector(b.V)
            held = False
            if start_coefs is not None and lay.part is None and hasattr(be, "start_residual"):
                try:
                    be.start_residual(op, start_coefs, b.dev(), r.dev_for_write())
                    held = True
                except Exception:          # noqa: BLE001 - products not held (more than 9 start vectors): the product below
                    held = False
            if not held:
                fem._halo(lay, x)
                be.spmv(op, x.dev(), r.dev_for_write(), lo, hi)
                r.touched_dev()
                r.scale(-1.0)
                r.axpy(1.0, b)
            else:
                r.touched_dev()
        g = _multidot(fem, lay, r, self.Y)
        if not (np.all(np.isfinite(G)) and np.all(np.isfinite(g))):
            return False
        d = np.sqrt(np.abs(np.diag(G)))
        d[d == 0.0] = 1.0
        coef = np.linalg.lstsq(G / np.outer(d, d), g / d, rcond=1e-12)[0] / d
        if not np.all(np.isfinite(coef)) or not np.any(coef):
            return False
        out = be.vec_zeros(lay.n)
        vecs = [y.dev() for y in self.Y]
        cs = [float(c) for c in coef]
        if not x._zero:
            vecs, cs = [x.dev()] + vecs, [1.0] + cs
        be.vec_lincomb(out, vecs, cs)
        be.vec_copy(x.dev_for_write(), out)
        be.vec_free(out)
        x.touched_dev()
        if lay.part is not None:
            x._halo_version = -1
        STATS["corrections"] += 1
        return True


def _multidot(fem, lay, x, ys):
    be = fem.get_backend()
    lo, hi = lay.owned_range()
    out = []
    for i in range(0, len(ys), 16):
        chunk = ys[i:i + 16]
        v = be.vec_multidot(x.dev(), [y.dev() for y in chunk], lo, hi) if len(chunk) > 1 else [be.vec_dot(x.dev(), chunk[0].dev(), lo, hi)]
        out.extend(float(t) for t in v)
    out = np.array(out)
    if lay.part is not None:
        out = lay.part.comm.allreduce_array(out)
    return out


def _signature(A):
    bc = A.bc_vertices
    return (int(bc.size), hash(bc[:: max(1, bc.size // 4096)].tobytes()))


def get(fem, A, b, k, prm):
    """The spectral start space of A's layout and Dirichlet set; harvested from (A, b) the first time it is asked for."""
    key = (id(A.lay), _signature(A))
    hit = _SPACES.get(key, False)
    if hit is not False and (hit is None or hit.lay() is A.lay):
        return hit
    if k < 0:                                # "auto": not before the space has seen AUTO_AFTER solves
        n = _AUTO_SOLVES.get(key, 0) + 1
        _AUTO_SOLVES[key] = n
        if n < AUTO_AFTER:
            return None
        k = -k
    sp = None
    try:
        sp = harvest(fem, A, b, k)
    except _Unavailable as e:
        STATS["dropped_requests"] += 1
        LOG.warning("spectral start asked for but not available on this system: %s", e)
    _SPACES[key] = sp
    if hit is False:
        weakref.finalize(A.lay, _SPACES.pop, key, None)        # the k vectors (2 GB at 256^3) go with their layout
    return sp


class _Unavailable(Exception):
    pass


def harvest(fem, A, b, k, steps=None):
    be = fem.get_backend()
    lay, V = A.lay, b.V
    t0 = time.perf_counter()
    m = int(steps) if steps else max(int(2.5 * k), k + 4)
    # (PGD_SPECTRAL_HARVEST_PRECONDITIONER=jacobi: the inverse-Lanczos solves through the Jacobi-PCG even where the V-cycle applies - the
    # price of a harvest without any multigrid, measured in HISTORY.md r04)
    prec = os.environ.get("PGD_SPECTRAL_HARVEST_PRECONDITIONER", "amg")
    prm = fem._Params(linear_solver="cg", preconditioner=prec, relative_tolerance=1e-10, spectral_start=0)
    any_solver = os.environ.get("PGD_SPECTRAL_ANY_SOLVER") == "1" or prec not in fem.MULTIGRID_NAMES
    v = b.copy()
    if A.bc_vertices.size:
        be.vec_set(v.dev(), A.bc_vertices, 0.0)          # the whole Krylov space then vanishes on the eliminated nodes
        v.touched_dev()
    nv = v.norm("l2")
    if not nv > 0.0:
        raise _Unavailable("zero right-hand side")
    v.scale(1.0 / nv)
    Q, its = [], 0
    for j in range(m):
        Q.append(v)
        w = fem.Vector(V)
        info = fem._solve_linear(A, v, w, prm)
        its += int(info.get("iterations", 0))
        if j == 0 and info.get("method") != "mg_pcg" and not any_solver:
            raise _Unavailable("the multigrid preconditioner does not apply here (%s): a harvest through Jacobi-PCG solves would "
                               "cost %d cold solves; PGD_SPECTRAL_ANY_SOLVER=1 accepts that" % (info.get("method"), m))
        for _ in range(GS_PASSES):                         # classical Gram-Schmidt, twice: w -= Q (Q'w), eight vectors per pass
            h = _multidot(fem, lay, w, Q)
            out = fem.Vector(V)
            be.vec_lincomb(out.dev_for_write(), [w.dev()] + [q.dev() for q in Q], [1.0] + [-float(hj) for hj in h])
            out.touched_dev()
            w = out
        nw = w.norm("l2")
        if not nw > 1e-13:
            break
        w.scale(1.0 / nw)
        v = w
    # Rayleigh-Ritz of A on span Q
    op = A.op()
    lo, hi = lay.owned_range()
    try:
        AQ = []
        for q in Q:
            y = fem.Vector(V)
            fem._halo(lay, q)
            be.spmv(op, q.dev(), y.dev_for_write(), lo, hi)
            y.touched_dev()
            AQ.append(y)
    finally:
        be.atom_free(op)
    G = np.array([_multidot(fem, lay, aq, Q) for aq in AQ])
    G = 0.5 * (G + G.T)
    th, S = np.linalg.eigh(G)
    Y, theta, res = [], [], []
    r = fem.Vector(V)
    for i in range(len(Q)):
        if len(Y) == k:
            break
        y = fem.Vector(V)
        be.vec_lincomb(y.dev_for_write(), [q.dev() for q in Q], [float(c) for c in S[:, i]])
        y.touched_dev()
        be.vec_lincomb(r.dev_for_write(), [aq.dev() for aq in AQ], [float(c) for c in S[:, i]])
        r.touched_dev()
        r.axpy(-float(th[i]), y)
        rel = r.norm("l2") / abs(th[i]) if th[i] != 0.0 else float("inf")
        if th[i] > 0.0 and rel < RESIDUAL_BAR:
            Y.append(y)
            theta.append(float(th[i]))
            res.append(float(rel))
    del Q, AQ, r
    be.sync()
    dt = time.perf_counter() - t0
    STATS["harvests"] += 1
    STATS["harvest_seconds"] += dt
    if not Y:
        raise _Unavailable("no Ritz pair converged in %d steps" % m)
    info = {"vectors": len(Y), "asked": k, "lanczos_steps": m, "inner_pcg_iterations": its, "seconds": dt,
            "ritz_values": theta, "relative_residuals": res, "rows": lay.n}
    LOG.info("spectral start: %d Ritz vectors of %d rows in %.2f s (%d inverse-Lanczos steps, %d inner PCG iterations)",
             len(Y), lay.n, dt, m, its)
    return SpectralStart(lay, Y, theta, res, info)
