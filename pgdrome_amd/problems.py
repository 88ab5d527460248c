"""Synthetic separable problems of BASELINE.json's configs, written exactly the
way a PGDrome user writes a problem: meshes + FunctionSpaces, a ``bc_fct`` and
the two weak-form callbacks ``lhs_fct`` / ``rhs_fct`` that dispatch on ``typ``
(callback contract: /root/reference/pgdrome/solver.py:547-569; grammar modelled
on /root/reference/tests/integration/test_laplace.py:73-366 and
test_heat1D.py:55-266).  ``bench.py`` and the parity tests run them through
``PGDProblem``; tests/golden/make_fixtures.py runs the same callbacks through
the reference's own ``PGDProblem``.

All loads and coefficients are P1 (interpolated), so the closed-form P1 atoms
integrate every form exactly.  Markers are written with numpy-friendly logic so
they evaluate vectorised on multi-million-vertex meshes.
"""
from __future__ import annotations

from . import fem


def _on_boundary(x, on_boundary):
    return on_boundary


# ------------------------------------------------------------------ config 1: 1D x 1D Poisson
def poisson_1d1d(n=32):
    """-Laplace(u) = 1 on (0,1)^2, u = 0 on the boundary, u = sum X(x) Y(y)."""
    meshes = [fem.IntervalMesh(n - 1, 0.0, 1.0) for _ in range(2)]
    Vs = [fem.FunctionSpace(m, "CG", 1) for m in meshes]
    load = [[fem.interpolate(fem.Expression("1.0", degree=1), V)] for V in Vs]

    def bc_fct(Vs, dom, param):
        return [fem.DirichletBC(Vs[0], 0, _on_boundary), fem.DirichletBC(Vs[1], 0, _on_boundary)]

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d, o = (0, 1) if typ == "r" else (1, 0)
        return (fem.Constant(fem.assemble(Fs[o] * Fs[o] * fem.dx(meshes[o]))) * u.dx(0) * v.dx(0) * fem.dx(meshes[d])
                + fem.Constant(fem.assemble(Fs[o].dx(0) * Fs[o].dx(0) * fem.dx(meshes[o]))) * u * v * fem.dx(meshes[d]))

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        d, o = (0, 1) if typ == "r" else (1, 0)
        l = fem.Constant(fem.assemble(Q[o][0] * Fs[o] * fem.dx(meshes[o]))) * Q[d][0] * v * fem.dx(meshes[d])
        for old in range(nE):
            l += (-fem.Constant(fem.assemble(PGD_func[o][old] * Fs[o] * fem.dx(meshes[o])))
                  * PGD_func[d][old].dx(0) * v.dx(0) * fem.dx(meshes[d])
                  - fem.Constant(fem.assemble(PGD_func[o][old].dx(0) * Fs[o].dx(0) * fem.dx(meshes[o])))
                  * PGD_func[d][old] * v * fem.dx(meshes[d]))
        return l

    return dict(name="poisson_1d1d", name_coord=["X", "Y"], modes_info=["U", "Node", "Scalar"], Vs=Vs,
                bc_fct=bc_fct, load=load, param={}, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=["r", "s"],
                PGD_nmax=3, PGD_tol=1e-10)


# ------------------------------------------------- an ALGEBRAIC parameter dimension (solve mode "direct")
def reaction_direct_param(n_x=33, n_e=17, e_range=(1.0, 20.0), PGD_nmax=4, PGD_tol=1e-8):
    """-u'' + E u = 1 on (0, 1), u(0) = u(1) = 0, u = sum X(x) G(E): config 2's physics in 1-D with the parameter
    dimension solved ALGEBRAICALLY, ``solve_modes = ["FEM", "direct"]`` (/root/reference/pgdrome/solver.py:637-638,
    717-718 -> direct_solve :909-925): for that dimension the callbacks return plain arrays, one value per dof
    (a_i = X'.X' + E_i X.X, b_i = the load minus the stored modes' terms at E_i), and the new factor is b / a.
    Called for the "stiff" normalisation (solver.py:424-441: Functions in both slots, ``dim`` out of range) the
    left-hand-side callback returns the scalar the reference takes as ``norm_aux`` (:439-441)."""
    meshes = [fem.IntervalMesh(n_x - 1, 0.0, 1.0), fem.IntervalMesh(n_e - 1, e_range[0], e_range[1])]
    Vs = [fem.FunctionSpace(m, "CG", 1) for m in meshes]
    load = [[fem.interpolate(fem.Expression("1.0", degree=1), V)] for V in Vs]
    param = {"E": fem.interpolate(fem.Expression("x[0]", degree=1), Vs[1])}

    def bc_fct(Vs, dom, param):
        return [fem.DirichletBC(Vs[0], 0, _on_boundary), 0]

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        E = param["E"]
        if typ == "x":
            return (fem.Constant(fem.assemble(Fs[1] * Fs[1] * fem.dx(meshes[1]))) * u.dx(0) * v.dx(0) * fem.dx(meshes[0])
                    + fem.Constant(fem.assemble(E * Fs[1] * Fs[1] * fem.dx(meshes[1]))) * u * v * fem.dx(meshes[0]))
        c_k = fem.assemble(Fs[0].dx(0) * Fs[0].dx(0) * fem.dx(meshes[0]))
        c_m = fem.assemble(Fs[0] * Fs[0] * fem.dx(meshes[0]))
        if dim >= len(Fs):
            return (c_k * fem.assemble(Fs[1] * Fs[1] * fem.dx(meshes[1]))
                    + c_m * fem.assemble(E * Fs[1] * Fs[1] * fem.dx(meshes[1])))
        return c_k + c_m * E.vector()[:]

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        E = param["E"]
        if typ == "x":
            l = fem.Constant(fem.assemble(Q[1][0] * Fs[1] * fem.dx(meshes[1]))) * Q[0][0] * v * fem.dx(meshes[0])
            for old in range(nE):
                l += (-fem.Constant(fem.assemble(PGD_func[1][old] * Fs[1] * fem.dx(meshes[1])))
                      * PGD_func[0][old].dx(0) * v.dx(0) * fem.dx(meshes[0])
                      - fem.Constant(fem.assemble(E * PGD_func[1][old] * Fs[1] * fem.dx(meshes[1])))
                      * PGD_func[0][old] * v * fem.dx(meshes[0]))
            return l
        b = fem.assemble(Q[0][0] * Fs[0] * fem.dx(meshes[0])) * Q[1][0].vector()[:]
        for old in range(nE):
            b = b - ((fem.assemble(PGD_func[0][old].dx(0) * Fs[0].dx(0) * fem.dx(meshes[0]))
                      + fem.assemble(PGD_func[0][old] * Fs[0] * fem.dx(meshes[0])) * E.vector()[:])
                     * PGD_func[1][old].vector()[:])
        return b

    return dict(name="reaction_direct_param", name_coord=["X", "E"], modes_info=["U", "Node", "Scalar"], Vs=Vs,
                bc_fct=bc_fct, load=load, param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=["x", "e"],
                PGD_nmax=PGD_nmax, PGD_tol=PGD_tol)


# ------------------------------------- configs 2 and 4: -Laplace(u) + mu u = 1, u(x; mu)
def reaction_diffusion(space_mesh, n_mu=128, mu_range=(1.0, 10.0), PGD_nmax=10, PGD_tol=1e-8, degree=1):
    """Space (2-D or 3-D, P1 - or P2: `degree`) x 1-D parameter mu: atoms K_x (x) M_mu + M_x (x) Mw_mu, w = mu."""
    mu_mesh = fem.IntervalMesh(n_mu - 1, mu_range[0], mu_range[1])
    meshes = [space_mesh, mu_mesh]
    Vs = [fem.FunctionSpace(space_mesh, "CG", int(degree)), fem.FunctionSpace(mu_mesh, "CG", 1)]
    load = [[fem.interpolate(fem.Expression("1.0", degree=1), Vs[0])],
            [fem.interpolate(fem.Expression("1.0", degree=1), Vs[1])]]
    param = {"mu": fem.interpolate(fem.Expression("x[0]", degree=1), Vs[1])}

    def bc_fct(Vs, dom, param):
        return [fem.DirichletBC(Vs[0], 0, _on_boundary), 0]

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        mu = param["mu"]
        if typ == "x":
            return (fem.Constant(fem.assemble(Fs[1] * Fs[1] * fem.dx(meshes[1])))
                    * fem.inner(fem.grad(u), fem.grad(v)) * fem.dx(meshes[0])
                    + fem.Constant(fem.assemble(mu * Fs[1] * Fs[1] * fem.dx(meshes[1])))
                    * u * v * fem.dx(meshes[0]))
        return (fem.Constant(fem.assemble(fem.inner(fem.grad(Fs[0]), fem.grad(Fs[0])) * fem.dx(meshes[0])))
                * u * v * fem.dx(meshes[1])
                + fem.Constant(fem.assemble(Fs[0] * Fs[0] * fem.dx(meshes[0])))
                * mu * u * v * fem.dx(meshes[1]))

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        mu = param["mu"]
        if typ == "x":
            l = fem.Constant(fem.assemble(Q[1][0] * Fs[1] * fem.dx(meshes[1]))) * Q[0][0] * v * fem.dx(meshes[0])
            for old in range(nE):
                l += (-fem.Constant(fem.assemble(PGD_func[1][old] * Fs[1] * fem.dx(meshes[1])))
                      * fem.inner(fem.grad(PGD_func[0][old]), fem.grad(v)) * fem.dx(meshes[0])
                      - fem.Constant(fem.assemble(mu * PGD_func[1][old] * Fs[1] * fem.dx(meshes[1])))
                      * PGD_func[0][old] * v * fem.dx(meshes[0]))
            return l
        l = fem.Constant(fem.assemble(Q[0][0] * Fs[0] * fem.dx(meshes[0]))) * Q[1][0] * v * fem.dx(meshes[1])
        for old in range(nE):
            l += (-fem.Constant(fem.assemble(fem.inner(fem.grad(PGD_func[0][old]), fem.grad(Fs[0])) * fem.dx(meshes[0])))
                  * PGD_func[1][old] * v * fem.dx(meshes[1])
                  - fem.Constant(fem.assemble(PGD_func[0][old] * Fs[0] * fem.dx(meshes[0])))
                  * mu * PGD_func[1][old] * v * fem.dx(meshes[1]))
        return l

    return dict(name="reaction_diffusion", name_coord=["X", "mu"], modes_info=["U", "Node", "Scalar"], Vs=Vs,
                bc_fct=bc_fct, load=load, param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=["x", "m"],
                PGD_nmax=PGD_nmax, PGD_tol=PGD_tol)


# ----------------- configs 3 and 5: rho c dT/dt - k Laplace(T) = q(x) (x) 1(t) [(x) params]
def transient_heat(space_mesh, n_t=256, n_p=0, rho_c=1.0, k=0.1, p_range=(0.5, 2.0), PGD_nmax=20,
                   PGD_tol=1e-6):
    """Space x time [x two material parameters]:  mu1 rho c dT/dt - mu2 k Laplace(T) = q.

    T = 0 on the spatial boundary, T(t=0) = 0 as a Dirichlet condition on the
    time dimension (as test_heat1D.py:43-52 does).  Atoms:
    M_x (x) C_t [(x) Mw_1 (x) M_2]  +  K_x (x) M_t [(x) M_1 (x) Mw_2];  the time
    problem is non-symmetric (u'v) and goes to the banded direct solve."""
    gdim = space_mesh.geometry().dim()
    t_mesh = fem.IntervalMesh(n_t - 1, 0.0, 1.0)
    meshes = [space_mesh, t_mesh]
    names, probs = ["X", "t"], ["x", "t"]
    if n_p:
        meshes += [fem.IntervalMesh(n_p - 1, p_range[0], p_range[1]) for _ in range(2)]
        names += ["mu1", "mu2"]
        probs += ["p1", "p2"]
    Vs = [fem.FunctionSpace(m, "CG", 1) for m in meshes]
    D = len(meshes)
    r2 = " + ".join("pow(x[%d]-0.5, 2)" % i for i in range(gdim))
    q_x = fem.interpolate(fem.Expression("exp(-(%s)/0.02)" % r2, degree=1), Vs[0])
    load = [[q_x]] + [[fem.interpolate(fem.Expression("1.0", degree=1), V)] for V in Vs[1:]]
    param = {"rho_c": rho_c, "k": k}
    if n_p:
        param["w1"] = fem.interpolate(fem.Expression("x[0]", degree=1), Vs[2])
        param["w2"] = fem.interpolate(fem.Expression("x[0]", degree=1), Vs[3])

    def bc_fct(Vs, dom, param):
        def t0(x, on_boundary):
            return x[0] < 1e-10
        return [fem.DirichletBC(Vs[0], 0, _on_boundary), fem.DirichletBC(Vs[1], 0, t0)] + [0] * (len(Vs) - 2)

    # per term (0: capacity, 1: conduction) and dimension, the functional G^T A F of the other dims
    def other(term, d, G, F, meshes, param):
        m = meshes[d]
        if d == 0:
            return fem.assemble(G * F * fem.dx(m)) if term == 0 else \
                fem.assemble(fem.inner(fem.grad(G), fem.grad(F)) * fem.dx(m))
        if d == 1:
            # the trial side (old mode or the iterate itself) carries the time derivative
            return fem.assemble(G.dx(0) * F * fem.dx(m)) if term == 0 else fem.assemble(G * F * fem.dx(m))
        w = param["w1"] if d == 2 else param["w2"]
        weighted = (term == 0 and d == 2) or (term == 1 and d == 3)
        return fem.assemble(w * G * F * fem.dx(m)) if weighted else fem.assemble(G * F * fem.dx(m))

    def own(term, d, u, v, meshes, param):
        m = meshes[d]
        if d == 0:
            return u * v * fem.dx(m) if term == 0 else fem.inner(fem.grad(u), fem.grad(v)) * fem.dx(m)
        if d == 1:
            return u.dx(0) * v * fem.dx(m) if term == 0 else u * v * fem.dx(m)
        w = param["w1"] if d == 2 else param["w2"]
        weighted = (term == 0 and d == 2) or (term == 1 and d == 3)
        return w * u * v * fem.dx(m) if weighted else u * v * fem.dx(m)

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d = probs.index(typ)
        a = 0
        for term, phys in ((0, param["rho_c"]), (1, param["k"])):
            c = phys
            for j in range(D):
                if j != d:
                    c *= other(term, j, Fs[j], Fs[j], meshes, param)
            a += fem.Constant(c) * own(term, d, u, v, meshes, param)
        return a

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        d = probs.index(typ)
        c = 1.0
        for j in range(D):
            if j != d:
                c *= fem.assemble(Q[j][0] * Fs[j] * fem.dx(meshes[j]))
        l = fem.Constant(c) * Q[d][0] * v * fem.dx(meshes[d])
        for old in range(nE):
            for term, phys in ((0, param["rho_c"]), (1, param["k"])):
                c = phys
                for j in range(D):
                    if j != d:
                        c *= other(term, j, PGD_func[j][old], Fs[j], meshes, param)
                l += -fem.Constant(c) * own(term, d, PGD_func[d][old], v, meshes, param)
        return l

    return dict(name="transient_heat", name_coord=names, modes_info=["T", "Node", "Scalar"], Vs=Vs,
                bc_fct=bc_fct, load=load, param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=probs,
                PGD_nmax=PGD_nmax, PGD_tol=PGD_tol)


# ----------------- a CONVECTIVE spatial term: - kappa Laplace(u) + w beta . grad(u) + u = 1, u = sum X(x) K(kappa) W(w)
def convection_diffusion(space_mesh, n_k=9, n_w=9, beta=(12.0, -5.0, 3.0), k_range=(0.5, 2.0), w_range=(0.0, 1.0), PGD_nmax=6,
                         PGD_tol=1e-8):
    """Space (2-D or 3-D, P1) x diffusivity kappa x velocity scale w (three-way separated).  The spatial operator
    a_K K_x + a_C sum_a beta_a C_{x,a} + a_M M_x  carries convection atoms u.dx(a) * v * dx - the atom the reference uses on its
    time axis (/root/reference/tests/integration/test_heat1D.py:80) - on a 2-D / 3-D space: NOT symmetric, solved by the
    reference's MUMPS (solver.py:627-636) and by BiCGStab here (csrc/pgd_krylov.hip)."""
    gdim = space_mesh.geometry().dim() if hasattr(space_mesh, "geometry") else space_mesh.coordinates().shape[1]
    meshes = [space_mesh, fem.IntervalMesh(n_k - 1, k_range[0], k_range[1]), fem.IntervalMesh(n_w - 1, w_range[0], w_range[1])]
    Vs = [fem.FunctionSpace(m, "CG", 1) for m in meshes]
    load = [[fem.interpolate(fem.Expression("1.0", degree=1), V)] for V in Vs]
    param = {"kappa": fem.interpolate(fem.Expression("x[0]", degree=1), Vs[1]),
             "w": fem.interpolate(fem.Expression("x[0]", degree=1), Vs[2]), "beta": tuple(beta[:gdim])}
    probs = ["x", "k", "w"]

    def bc_fct(Vs, dom, param):
        return [fem.DirichletBC(Vs[0], 0, _on_boundary), 0, 0]

    # term t of the operator = product over the dimensions of these bilinear forms (f, g: trial / test or two Functions)
    def own(term, d, f, g, meshes, param):
        m = meshes[d]
        if d == 0:
            if term == 0:
                return fem.inner(fem.grad(f), fem.grad(g)) * fem.dx(m)
            if term == 1:
                form = 0
                for a, b_a in enumerate(param["beta"]):
                    form = form + fem.Constant(b_a) * f.dx(a) * g * fem.dx(m)
                return form
            return f * g * fem.dx(m)
        weight = param["kappa"] if (d == 1 and term == 0) else (param["w"] if (d == 2 and term == 1) else None)
        return weight * f * g * fem.dx(m) if weight is not None else f * g * fem.dx(m)

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d = probs.index(typ)
        a = 0
        for term in range(3):
            c = 1.0
            for j in range(3):
                if j != d:
                    c *= fem.assemble(own(term, j, Fs[j], Fs[j], meshes, param))
            a = a + fem.Constant(c) * own(term, d, u, v, meshes, param)
        return a

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        d = probs.index(typ)
        c = 1.0
        for j in range(3):
            if j != d:
                c *= fem.assemble(Q[j][0] * Fs[j] * fem.dx(meshes[j]))
        l = fem.Constant(c) * Q[d][0] * v * fem.dx(meshes[d])
        for old in range(nE):
            for term in range(3):
                c = 1.0
                for j in range(3):
                    if j != d:
                        # (trial slot = the stored mode, test slot = the iterate: the convection form is not symmetric)
                        c *= fem.assemble(own(term, j, PGD_func[j][old], Fs[j], meshes, param))
                l = l + fem.Constant(-c) * own(term, d, PGD_func[d][old], v, meshes, param)
        return l

    return dict(name="convection_diffusion", name_coord=["X", "kappa", "w"], modes_info=["U", "Node", "Scalar"], Vs=Vs,
                bc_fct=bc_fct, load=load, param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=probs,
                PGD_nmax=PGD_nmax, PGD_tol=PGD_tol)


def elastic_block(space_mesh, n_e=9, e_range=(0.5, 2.0), nu=0.3, k_found=2.0, PGD_nmax=3, PGD_tol=1e-8, degree=1, traction=None):
    """A 3-D block clamped at x = 0 on an elastic foundation under its own weight: VECTOR-valued P1 displacement u(X; e), Young's
    modulus factor e as the second PGD variable.   int eps(v) : (e C(nu)) eps(u) + k v . u dX = int g . v dX,  g = (0, 0, -1);
    u = sum_m U_m(X) W_m(e).
    (Voigt strain as in the reference's elastic test, /root/reference/tests/integration/test_solver_problem.py:59-75, which is 2-D
    and P2; this one exists to carry a vector-valued space through the row-sharded solve.)"""
    lam, mu = nu / ((1.0 + nu) * (1.0 - 2.0 * nu)), 1.0 / (2.0 * (1.0 + nu))
    C = fem.as_matrix([[lam + 2 * mu, lam, lam, 0, 0, 0], [lam, lam + 2 * mu, lam, 0, 0, 0], [lam, lam, lam + 2 * mu, 0, 0, 0],
                       [0, 0, 0, mu, 0, 0], [0, 0, 0, 0, mu, 0], [0, 0, 0, 0, 0, mu]])
    g = fem.Constant((0.0, 0.0, -1.0))
    meshes = [space_mesh, fem.IntervalMesh(n_e - 1, e_range[0], e_range[1])]
    Vs = [fem.VectorFunctionSpace(meshes[0], "CG", int(degree)), fem.FunctionSpace(meshes[1], "CG", 1)]
    param = {"e": fem.interpolate(fem.Expression("x[0]", degree=1), Vs[1])}

    def strain(w):
        return fem.as_vector([w[0].dx(0), w[1].dx(1), w[2].dx(2), w[1].dx(2) + w[2].dx(1), w[0].dx(2) + w[2].dx(0), w[0].dx(1) + w[1].dx(0)])

    def clamped(x, on_boundary):
        return on_boundary and fem.near(x[0], 0.0)

    def bc_fct(Vs, dom, param):
        return [fem.DirichletBC(Vs[0], fem.Constant((0.0, 0.0, 0.0)), clamped), 0]

    def op_form(t, j, a, b, meshes, param):
        """term t of the operator on dimension j: t = 0 the strain energy x e-weighted mass, t = 1 the foundation x mass"""
        if j == 0:
            return (fem.inner(C * strain(a), strain(b)) if t == 0 else fem.Constant(k_found) * fem.inner(a, b)) * fem.dx(meshes[0])
        return (param["e"] * a * b if t == 0 else a * b) * fem.dx(meshes[1])

    def load_form(j, b, meshes):
        if j == 0:
            l = fem.dot(g, b) * fem.dx(meshes[0])
            if traction is not None:          # (a surface load on the whole boundary - exterior-facet integrals, `ds`)
                l = l + fem.dot(fem.Constant(tuple(traction)), b) * fem.ds(meshes[0])
            return l
        return b * fem.dx(meshes[1])

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d = 0 if typ == "x" else 1
        a = 0
        for t in (0, 1):
            c = fem.assemble(op_form(t, 1 - d, Fs[1 - d], Fs[1 - d], meshes, param))
            a = a + fem.Constant(c) * op_form(t, d, u, v, meshes, param)
        return a

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        d = 0 if typ == "x" else 1
        l = fem.Constant(fem.assemble(load_form(1 - d, Fs[1 - d], meshes))) * load_form(d, v, meshes)
        for old in range(nE):
            for t in (0, 1):
                c = fem.assemble(op_form(t, 1 - d, PGD_func[1 - d][old], Fs[1 - d], meshes, param))
                l = l - fem.Constant(c) * op_form(t, d, PGD_func[d][old], v, meshes, param)
        return l

    return dict(name="elastic_block", name_coord=["X", "e"], modes_info=["U", "Node", "Vector"], Vs=Vs, bc_fct=bc_fct, load=[],
                param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=["x", "e"], PGD_nmax=PGD_nmax, PGD_tol=PGD_tol)


def make_problem(spec, cls):
    """PGDProblem(**spec) for either implementation of the class."""
    return cls(**spec)


CONFIGS = {
    # name -> (builder, description); the sizes of BASELINE.json's configs
    "cfg1": (lambda: poisson_1d1d(32), "1Dx1D separated Poisson, 32 P1 dofs per dimension, 3 modes"),
    "cfg2": (lambda: reaction_diffusion(fem.RectangleMesh(fem.Point(0, 0), fem.Point(1, 1), 255, 255), 128,
                                        PGD_nmax=10), "2D-space 256^2 P1 x 1D-parameter (128), 10 modes"),
    "cfg3": (lambda: transient_heat(fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, 1), 127, 127, 127), 256,
                                    PGD_nmax=20), "3D-space 128^3 P1 x 1D-time (256), 20 modes"),
    "cfg4": (lambda: reaction_diffusion(fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, 1), 255, 255, 255), 128,
                                        PGD_nmax=10), "3D-space 256^3 P1 x 1D-parameter (128)"),
    "cfg5": (lambda: transient_heat(fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, 1), 255, 255, 255), 256, 64,
                                    PGD_nmax=50), "3D-space 256^3 P1 x time (256) x 2 parameters (64), 50 modes"),
}
