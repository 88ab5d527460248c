"""The 1-D linear-elastic truss of the reference's integration test
(/root/reference/tests/integration/test_elastic.py: PGD variables x, load factor p, stiffness
factor E; quadratic Lagrange elements on all three meshes; default Newton-type solver; "stiff"
normalisation), restated in this repository's own words.

    -(E A u')' = p A x ... separated as  u(x, p, E) = sum X(x) P(p) W(E),   u(0) = u(1) = 0
    operator:  K_x (x) M_p (x) Mw_E  (w = E),      load:  1(x) (x) p(p) (x) 1(E)

Analytic solution used by the reference as its oracle (test_elastic.py:294-303):
    u = p / (2 E) (x - x^2)
"""
import numpy as np


def build(fem, elems=(113, 2, 100), order=2):
    ranges = [(0.0, 1.0), (-1.0, 3.0), (0.2, 2.0)]
    meshes = [fem.IntervalMesh(elems[i], ranges[i][0], ranges[i][1]) for i in range(3)]
    Vs = [fem.FunctionSpace(m, "P", order) for m in meshes]
    param = {"A": 1.0, "p_0": 1.0, "E_0": 1.0, "Efunc": fem.Expression("x[0]", degree=4)}
    load = [[fem.Expression("1.0", degree=4)],
            [fem.Expression("p*A*x[0]", p=param["p_0"], A=param["A"], degree=4)],
            [fem.Expression("1.0", degree=4)]]
    probs = ["r", "s", "t"]

    def bc_fct(Vs, dom, param):
        def left(x, on_boundary):
            return x < 0.0 + 1e-5

        def right(x, on_boundary):
            return x > 1.0 - 1e-5
        return [[fem.DirichletBC(Vs[0], 0.0, left), fem.DirichletBC(Vs[0], 0.0, right)], 0, 0]

    def functional(j, G, F, meshes, param):
        """G^T A_j F of the single operator term on dimension j."""
        m = meshes[j]
        if j == 0:
            return fem.assemble(F.dx(0) * param["E_0"] * G.dx(0) * param["A"] * fem.dx(m))
        if j == 1:
            return fem.assemble(F * G * fem.dx(m))
        return fem.assemble(F * param["Efunc"] * G * fem.dx(m))

    def own_form(d, u, v, meshes, param):
        m = meshes[d]
        if d == 0:
            return v.dx(0) * param["E_0"] * u.dx(0) * param["A"] * fem.dx(m)
        if d == 1:
            return v * u * fem.dx(m)
        return v * param["Efunc"] * u * fem.dx(m)

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d = probs.index(typ)
        c = np.prod([functional(j, Fs[j], Fs[j], meshes, param) for j in range(3) if j != d])
        return fem.Constant(c) * own_form(d, u, v, meshes, param)

    def rhs_fct(u, v, Fs, meshes, dom, param, G, PGD_func, typ, nE, dim):
        d = probs.index(typ)
        area = [param["A"], 1.0, 1.0]
        c = np.prod([fem.assemble(Fs[j] * G[j][0] * area[j] * fem.dx(meshes[j])) for j in range(3) if j != d])
        l = fem.Constant(c) * v * G[d][0] * area[d] * fem.dx(meshes[d])
        for old in range(nE):
            c = np.prod([functional(j, PGD_func[j][old], Fs[j], meshes, param) for j in range(3) if j != d])
            l += -fem.Constant(c) * own_form(d, PGD_func[d][old], v, meshes, param)
        return l

    spec = dict(name="Uniaxial1D-PGD-XPE", name_coord=["X", "P", "E"], modes_info=["U_x", "Node", "Scalar"],
                Vs=Vs, dom=0, bc_fct=bc_fct, load=load, param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct,
                probs=probs, seq_fp=[0, 1, 2], PGD_nmax=10)
    knobs = dict(stop_fp="norm", max_fp_it=50, tol_fp_it=1e-5, norm_modes="stiff")
    return spec, knobs, meshes


def analytic(x, p, E):
    return p / (2.0 * E) * (x - x * x)


def run_and_check(fem, PGDProblem, PGDErrorComputation):
    """Solve and apply the reference test's own assertions (test_elastic.py:330-380)."""
    spec, knobs, meshes = build(fem)
    prob = PGDProblem(**spec)
    for k, v in knobs.items():
        setattr(prob, k, v)
    prob.solve_PGD()                       # default _problem="nonlinear", as the reference test
    sol = prob.return_PGD()
    x = meshes[0].coordinates()

    def fom(smp):
        return analytic(x, smp[0], smp[1])
    err = PGDErrorComputation(fixed_dim=[0], n_samples=10, FOM_model=fom, PGD_model=sol)
    _, mean_e, max_e = err.evaluate_error()

    def fom_pt(smp):
        return analytic(np.array([0.5]), smp[0], smp[1])
    err3 = PGDErrorComputation(fixed_dim=[0], FOM_model=fom_pt, PGD_model=sol, data_test=[[2.0, 1.5], [1.0, 1.0]],
                               fixed_var=[0.5])
    _, mean_pt, _ = err3.evaluate_error()
    return prob, sol, mean_e, max_e, mean_pt
