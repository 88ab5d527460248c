"""The 2-D Laplace problem of the reference's integration test
(/root/reference/tests/integration/test_laplace.py: PGD variables x, y, source amplitude q, boundary
value u0; P1 intervals 60 / 40 / 200 / 80 elements; all-FEM and all-FD variants), restated in this
repository's own words.

    -k (T_xx + T_yy) = Q_x(x) Q_y(y) q,   T = u0 (1 - x/3) on x = 0 and x = lx
    T = B_x B_y B_q B_u + sum_m X_m(x) Y_m(y) W_m(q) U_m(u0)      (B: the lifted boundary data)

Operator terms: 0: k K_x (x) M_y (x) M_q (x) M_u,   1: k M_x (x) K_y (x) M_q (x) M_u.
Right-hand side: load - A(lifting) - sum_old A(old mode).  In the FD variant every dimension is
finite differences: K -> -D2, M -> lumped mass (``FD_matrices`` on the sorted dof coordinates),
functionals are quadratic forms with those matrices and each problem is handed to ``FD_solve``
as (matrix, vector).  The reference asserts that BOTH variants converge in exactly one mode.
"""
import numpy as np


def build(fem, FD_matrices, fd=False, elems=(60, 40, 200, 80)):
    param = {"k": 0.5, "lx": 3, "ly": 3}
    ranges = [(0.0, 3.0), (0.0, 3.0), (0.0, 50.0), (10.0, 50.0)]
    meshes = [fem.IntervalMesh(elems[i], ranges[i][0], ranges[i][1]) for i in range(4)]
    Vs = [fem.FunctionSpace(m, "CG", 1) for m in meshes]
    lift = [fem.interpolate(fem.Expression("1.0-1.0/3.0*x[0]", degree=1), Vs[0]),
            fem.interpolate(fem.Expression("1.0", degree=1), Vs[1]),
            fem.interpolate(fem.Expression("1.0", degree=1), Vs[2]),
            fem.interpolate(fem.Expression("x[0]", degree=1), Vs[3])]
    param["lift"] = lift
    load = [[fem.interpolate(fem.Expression("x[0]<L/2 ? 1.0 : 0", degree=1, L=param["lx"]), Vs[0])],
            [fem.interpolate(fem.Expression("1.0", degree=1), Vs[1])],
            [fem.interpolate(fem.Expression("x[0]", degree=1), Vs[2])],
            [fem.interpolate(fem.Expression("1.0", degree=1), Vs[3])]]
    probs = ["r", "s", "t", "u"]
    stiff_dim = (0, 1)            # term t carries the stiffness on dimension stiff_dim[t]

    if fd:
        M, K = [], []
        for V in Vs:
            x = np.array(V.tabulate_dof_coordinates()[:].flatten())
            order = np.argsort(x)
            m, d2, _ = FD_matrices(x[order])
            M.append(m[order, :][:, order])
            K.append(-1.0 * d2[order, :][:, order])
        param["M"], param["K"] = M, K
        x0 = np.array(Vs[0].tabulate_dof_coordinates()[:].flatten())
        param["bc_idx"] = np.array([np.where(x0 == 0)[0], np.where(x0 == param["lx"])[0]]).flatten()

    def bc_fct(Vs, dom, param):
        def leftright(x, on_boundary):
            return on_boundary and fem.near(x[0], 0.0, 1e-6) or fem.near(x[0], param["lx"], 1e-6)
        return [fem.DirichletBC(Vs[0], 0, leftright), 0, 0, 0]

    def mat(term, j, param):
        return param["K"][j] if stiff_dim[term] == j else param["M"][j]

    def functional(term, j, G, F, meshes, param):
        if fd:
            return F.vector()[:].transpose() @ mat(term, j, param) @ G.vector()[:]
        if stiff_dim[term] == j:
            return fem.assemble(G.dx(0) * F.dx(0) * fem.dx(meshes[j]))
        return fem.assemble(G * F * fem.dx(meshes[j]))

    def own_form(term, d, u, v, meshes):
        return u.dx(0) * v.dx(0) * fem.dx(meshes[d]) if stiff_dim[term] == d else u * v * fem.dx(meshes[d])

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d = probs.index(typ)
        others = [j for j in range(4) if j != d]
        coef = [param["k"] * np.prod([functional(t, j, Fs[j], Fs[j], meshes, param) for j in others]) for t in (0, 1)]
        if fd:
            a = coef[0] * mat(0, d, param) + coef[1] * mat(1, d, param)
            if d == 0:
                a = a.tolil()
                a[:, param["bc_idx"]] = 0.0
                a[param["bc_idx"], :] = 0.0
                a[param["bc_idx"], param["bc_idx"]] = 1.0
            return a
        return fem.Constant(coef[0]) * own_form(0, d, u, v, meshes) + fem.Constant(coef[1]) * own_form(1, d, u, v, meshes)

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        d = probs.index(typ)
        others = [j for j in range(4) if j != d]
        known = [param["lift"]] + [[PGD_func[j][old] for j in range(4)] for old in range(nE)]
        if fd:
            c = np.prod([Fs[j].vector()[:].transpose() @ param["M"][j] @ Q[j][0].vector()[:] for j in others])
            l = c * param["M"][d] @ Q[d][0].vector()[:]
            for G in known:
                for t in (0, 1):
                    c = param["k"] * np.prod([functional(t, j, G[j], Fs[j], meshes, param) for j in others])
                    l = l - c * mat(t, d, param) @ G[d].vector()[:]
            if d == 0:
                l[param["bc_idx"]] = 0
            return l
        c = np.prod([fem.assemble(Q[j][0] * Fs[j] * fem.dx(meshes[j])) for j in others])
        l = fem.Constant(c) * Q[d][0] * v * fem.dx(meshes[d])
        for G in known:
            for t in (0, 1):
                c = param["k"] * np.prod([functional(t, j, G[j], Fs[j], meshes, param) for j in others])
                l += -fem.Constant(c) * own_form(t, d, G[d], v, meshes)
        return l

    spec = dict(name="test_x_y_q_u00", name_coord=["X", "Y", "q", "u0"], modes_info=["T", "Node", "Scalar"], Vs=Vs,
                dom=0, bc_fct=bc_fct, load=load, param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=probs,
                seq_fp=np.arange(4), PGD_nmax=7)
    knobs = dict(stop_fp="norm", max_fp_it=50, tol_fp_it=1e-5, norm_modes="stiff")
    return spec, knobs, (["FD"] * 4 if fd else ["FEM"] * 4), (param["M"] if fd else [])


def run(fem, PGDProblem, FD_matrices, fd=False):
    spec, knobs, solve_modes, MM = build(fem, FD_matrices, fd)
    p = PGDProblem(**spec)
    if fd:
        p.MM = MM
    for k, v in knobs.items():
        setattr(p, k, v)
    p.solve_PGD(_problem="linear", solve_modes=solve_modes)
    return p


def full_order_profile(fem, y, q, u0, elems=(60, 40), degree=2):
    """The 2-D full-order model the reference's test compares the PGD solution with
    (test_laplace.py:867-929): quadratic triangles on [0, 3]^2, T = u0 (1 - x/3) on the left / right edge,
    source q on the left half; returns T(x_i, y) on the 61 vertices of the PGD x-mesh."""
    k, lx, ly = 0.5, 3.0, 3.0
    mesh = fem.RectangleMesh(fem.Point(0, 0), fem.Point(lx, ly), elems[0], elems[1])
    V = fem.FunctionSpace(mesh, "CG", degree)
    v, T = fem.TestFunction(V), fem.TrialFunction(V)
    source = fem.Expression("x[0]<L/2 ? q00 : 0", degree=1, L=lx, q00=q)
    a = k * fem.inner(fem.grad(v), fem.grad(T)) * fem.dx()
    l = v * source * fem.dx()
    bc = fem.DirichletBC(V, fem.Expression("u00*(1. - 1./3.*x[0])", degree=1, u00=u0),
                         lambda x, on_boundary: on_boundary and fem.near(x[0], 0.0, 1e-6) or fem.near(x[0], lx, 1e-6))
    sol = fem.Function(V)
    fem.solve(a == l, sol, bcs=bc)
    xs = np.linspace(0, lx, elems[0] + 1)
    return np.array([sol((x, y)) for x in xs])


def pgd_profile(p, y, q, u0):
    """PGD field on the x-mesh for (y, q, u0), boundary lifting added as the reference's test does (:1043-1080)."""
    sol = p.return_PGD()
    lift = p.param["lift"]
    return sol.evaluate(0, [1, 2, 3], [y, q, u0], 0).compute_vertex_values() + \
        lift[0].compute_vertex_values() * lift[1](y) * lift[2](q) * lift[3](u0)
