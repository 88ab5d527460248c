"""The 2-D plane-strain cantilever of the reference's integration test
(/root/reference/tests/integration/test_solver_problem.py: PGD variables X = (x, y) in space, load factor p,
Young's-modulus factor E, Poisson ratio nu; VECTOR-valued P2 on a "crossed" mesh in space, P1 intervals for the
parameters; facets marked through MeshFunction / SubDomain, load through ds), restated in this repository's
own words and with the element counts as arguments.

    int eps(v) : (E E0 C(nu)) eps(u) dX = p int_top-left g1 . v ds + p int_top-right g2 . v ds,   u = 0 at x = 0
    u = sum_m U_m(X) P_m(p) W_m(E) N_m(nu)

The elasticity matrix separates as  C(nu) = f_0(nu) C_0 + f_1(nu) C_1  (Voigt notation, plane strain):
    f_0 = 1 / (2 (1 + nu) (1 - 2 nu)),  C_0 = [[1,1,0],[1,1,0],[0,0,0]];   f_1 = 1 / (2 (1 + nu)),  C_1 = [[1,-1,0],[-1,1,0],[0,0,1]]
so the operator has two terms t: (strain energy with C_t) x (mass in p) x (E-weighted mass) x (f_t-weighted mass),
and the load two terms k: (g_k on its half of the top edge) x p x 1 x 1.
"""
import numpy as np

LX, LY, E0 = 1000.0, 100.0, 30000.0
RANGES = [(0.0, 2.0), (0.5, 1.5), (0.1, 0.4)]
C_T = (np.array([[1.0, 1.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 0.0]]),
       np.array([[1.0, -1.0, 0.0], [-1.0, 1.0, 0.0], [0.0, 0.0, 1.0]]))
G_K = ((0.0, -0.5), (0.0, -1.5))       # traction on the left / right half of the top edge (facet tags 2 / 3)


def strain(fem, w):
    """Voigt strain (e_xx, e_yy, gamma_xy) of a vector field."""
    return fem.as_vector([w[0].dx(0), w[1].dx(1), w[0].dx(1) + w[1].dx(0)])


def mark_facets(fem, mesh):
    facets = fem.MeshFunction("size_t", mesh, mesh.topology().dim() - 1)
    facets.set_all(0)

    class Clamped(fem.SubDomain):
        def inside(self, x, on_boundary):
            return fem.near(x[0], 0.0)

    class TopLeft(fem.SubDomain):
        def inside(self, x, on_boundary):
            return fem.near(x[1], LY) and x[0] < 0.5 * LX

    class TopRight(fem.SubDomain):
        def inside(self, x, on_boundary):
            return fem.near(x[1], LY) and x[0] > 0.5 * LX

    Clamped().mark(facets, 1)
    TopLeft().mark(facets, 2)
    TopRight().mark(facets, 3)
    return facets


def spaces(fem, nx=40, ny=4, elems=(2, 10, 10)):
    mesh_x = fem.RectangleMesh(fem.Point(0.0, 0.0), fem.Point(LX, LY), nx, ny, "crossed")
    Vs = [fem.VectorFunctionSpace(mesh_x, "P", 2)]
    for n, (a, b) in zip(elems, RANGES):
        Vs.append(fem.FunctionSpace(fem.IntervalMesh(n, a, b), "P", 1))
    return Vs


def build(fem, Vs):
    """Arguments of PGDProblem for the 4-way separated cantilever."""
    Ct = [fem.as_matrix(c) for c in C_T]
    g = [fem.Constant(v) for v in G_K]
    weight = [None, None, fem.Expression("E0 * x[0]", degree=4, E0=E0),
              [fem.Expression("1.0/(2.0 * (1.0 + x[0]) * (1.0 - 2.0 * x[0]))", degree=10),
               fem.Expression("1.0/(2.0 * (1.0 + x[0]))", degree=10)]]
    load_factor = [None, fem.Expression("x[0]", degree=4), fem.Expression("1.0", degree=4), fem.Expression("1.0", degree=4)]
    probs = ["r", "s", "t", "v"]

    def dom_fct(Vs, param):
        return [mark_facets(fem, Vs[0].mesh()), 0, 0, 0]

    def bc_fct(Vs, dom, param):
        return [[fem.DirichletBC(Vs[0], fem.Constant((0.0, 0.0)), dom[0], 1)], 0, 0, 0]

    def op_form(t, j, a, b, meshes):
        """Term t of the operator on dimension j applied to (trial-side a, test-side b)."""
        if j == 0:
            return fem.inner(Ct[t] * strain(fem, a), strain(fem, b)) * fem.dx(meshes[0])
        if j == 1:
            return a * b * fem.dx(meshes[1])
        if j == 2:
            return a * weight[2] * b * fem.dx(meshes[2])
        return a * weight[3][t] * b * fem.dx(meshes[3])

    def load_form(k, j, b, meshes, dom):
        if j == 0:
            ds = fem.Measure("ds", domain=meshes[0], subdomain_data=dom[0])
            return fem.dot(g[k], b) * ds(2 + k)
        return load_factor[j] * b * fem.dx(meshes[j])

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d = probs.index(typ)
        a = 0
        for t in (0, 1):
            c = np.prod([fem.assemble(op_form(t, j, Fs[j], Fs[j], meshes)) for j in range(4) if j != d])
            a = a + fem.Constant(c) * op_form(t, d, u, v, meshes)
        return a

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        d = probs.index(typ)
        l = 0
        for k in (0, 1):
            c = np.prod([fem.assemble(load_form(k, j, Fs[j], meshes, dom)) for j in range(4) if j != d])
            l = l + fem.Constant(c) * load_form(k, d, v, meshes, dom)
        for old in range(nE):
            for t in (0, 1):
                c = np.prod([fem.assemble(op_form(t, j, PGD_func[j][old], Fs[j], meshes)) for j in range(4) if j != d])
                l = l - fem.Constant(c) * op_form(t, d, PGD_func[d][old], v, meshes)
        return l

    spec = dict(name="PGD_xpEv", name_coord=["X", "P", "E", "nu"], modes_info=["U", "Node", "Vector"], Vs=Vs,
                bc_fct=bc_fct, dom_fct=dom_fct, load=[], param={}, rhs_fct=rhs_fct, lhs_fct=lhs_fct,
                probs=probs, seq_fp=[0, 1, 2, 3], PGD_nmax=7)
    knobs = dict(max_fp_it=50, stop_fp="norm", tol_fp_it=1e-4, norm_modes="stiff")
    return spec, knobs


def full_order(fem, V, p, e, nu):
    """The 2-D finite-element solution for fixed (p, E-factor, nu) the PGD field is compared with
    (reference: FEM_reference, test_solver_problem.py:625-690)."""
    mesh = V.mesh()
    facets = mark_facets(fem, mesh)
    ds = fem.Measure("ds", domain=mesh, subdomain_data=facets)
    E = e * E0
    C = fem.as_matrix(E / ((1.0 + nu) * (1.0 - 2.0 * nu)) *
                      np.array([[1.0 - nu, nu, 0.0], [nu, 1.0 - nu, 0.0], [0.0, 0.0, (1.0 - 2.0 * nu) / 2.0]]))
    u, v = fem.TrialFunction(V), fem.TestFunction(V)
    a = fem.inner(C * strain(fem, u), strain(fem, v)) * fem.dx
    l = p * fem.dot(fem.Constant(G_K[0]), v) * ds(2) + p * fem.dot(fem.Constant(G_K[1]), v) * ds(3)
    sol = fem.Function(V, name="Displacement")
    fem.solve(a == l, sol, [fem.DirichletBC(V, fem.Constant((0.0, 0.0)), facets, 1)])
    return sol
