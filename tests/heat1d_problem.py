"""The 1-D transient heat problem of the reference's integration test
(/root/reference/tests/integration/test_heat1D.py: PGD variables x, t, q; Goldak-type
source; initial condition lifted through IC_x (x) IC_t (x) IC_q; all-FEM and
FEM/FD-in-time variants), restated here in this repository's own words so the parity
tests can run it where the reference tree does not exist.

    rho cp dT/dt - k d2T/dx2 = Q_x(x) Q_t(t) Q_q(q),    T = IC + sum_m X_m(x) S_m(t) W_m(q)

Operator terms:  0: rho cp * M_x (x) C_t (x) M_q      1: k * K_x (x) M_t (x) M_q
Right-hand side: load - A(IC) - sum_old A(mode_old): the lifted initial condition enters
exactly like one more "old mode".  In the FD-in-time variant the time factors of all
functionals are quadratic forms with the user-built matrices D1_up_t / M_t (dof order) and
the time problem is returned as (matrix, vector) with the initial-condition row imposed.
"""
import numpy as np


def build(fem, FD_matrices, elems=(15, 10, 10), fd_time=False):
    param = {"rho": 1, "cp": 1, "k": 0.5, "Tamb": 25, "Q": 1, "af": 0.2, "ar": 0.2, "xc": 0.5, "lx": 1, "lt": 1}
    ranges = [(0.0, param["lx"]), (0.0, param["lt"]), (0.5, 1.0)]
    meshes = [fem.IntervalMesh(elems[i], ranges[i][0], ranges[i][1]) for i in range(3)]
    Vs = [fem.FunctionSpace(m, "CG", 1) for m in meshes]
    ff = 6 * np.sqrt(3) / ((param["af"] + param["ar"]) * param["af"] * param["af"] * np.pi ** 1.5)
    q_x = fem.interpolate(fem.Expression("ff* exp(-3*(pow(x[0]-xc,2)/pow(af,2)))", degree=4, ff=ff,
                                         af=param["af"], ar=param["ar"], xc=param["xc"]), Vs[0])
    load = [q_x, fem.interpolate(fem.Expression("1.0", degree=1), Vs[1]),
            fem.interpolate(fem.Expression("x[0]*Q", Q=param["Q"], degree=1), Vs[2])]
    IC = [fem.interpolate(fem.Expression("1.0", degree=1), Vs[0]),
          fem.interpolate(fem.Expression("Tamb", degree=1, Tamb=param["Tamb"]), Vs[1]),
          fem.interpolate(fem.Expression("1.0", degree=1), Vs[2])]
    param["IC"] = IC
    phys = (param["rho"] * param["cp"], param["k"])
    probs = ["r", "s", "w"]

    if fd_time:
        t = np.array(Vs[1].tabulate_dof_coordinates()[:].flatten())
        order = np.argsort(t)
        M_t, _, D1 = FD_matrices(t[order])
        param["M_t"] = M_t[order, :][:, order]
        param["D1_up_t"] = D1[order, :][:, order]
        param["bc_idx"] = np.where(t == 0)[0]

    def bc_fct(Vs, dom, param):
        def start(x, on_boundary):
            return x < 0.0 + 1e-5
        return [0, fem.DirichletBC(Vs[1], 0, start), 0]

    def tmat(term, param):
        return param["D1_up_t"] if term == 0 else param["M_t"]

    def functional(term, j, G, F, meshes, param):
        """G^T A_{j,term} F: trial side G (old mode / IC / the iterate), test side F."""
        m = meshes[j]
        if j == 0:
            return fem.assemble(G * F * fem.dx(m)) if term == 0 else fem.assemble(G.dx(0) * F.dx(0) * fem.dx(m))
        if j == 1:
            if fd_time:
                return F.vector()[:].transpose() @ tmat(term, param) @ G.vector()[:]
            return fem.assemble(G.dx(0) * F * fem.dx(m)) if term == 0 else fem.assemble(G * F * fem.dx(m))
        return fem.assemble(G * F * fem.dx(m))

    def own_form(term, d, u, v, meshes):
        m = meshes[d]
        if d == 0:
            return u * v * fem.dx(m) if term == 0 else u.dx(0) * v.dx(0) * fem.dx(m)
        if d == 1:
            return u.dx(0) * v * fem.dx(m) if term == 0 else u * v * fem.dx(m)
        return u * v * fem.dx(m)

    def load_functional(j, Q, F, meshes, param):
        if j == 1 and fd_time:
            return F.vector()[:].transpose() @ param["M_t"] @ Q[1].vector()[:]
        return fem.assemble(Q[j] * F * fem.dx(meshes[j]))

    def lhs_fct(u, v, Fs, meshes, dom, param, typ, dim):
        d = probs.index(typ)
        others = [j for j in range(3) if j != d]
        coef = [phys[term] * np.prod([functional(term, j, Fs[j], Fs[j], meshes, param) for j in others])
                for term in (0, 1)]
        if d == 1 and fd_time:
            a = (coef[0] * param["D1_up_t"] + coef[1] * param["M_t"]).tolil()
            a[:, param["bc_idx"]] = 0
            a[param["bc_idx"], :] = 0
            a[param["bc_idx"], param["bc_idx"]] = 1
            return a
        return fem.Constant(coef[0]) * own_form(0, d, u, v, meshes) + fem.Constant(coef[1]) * own_form(1, d, u, v, meshes)

    def rhs_fct(u, v, Fs, meshes, dom, param, Q, PGD_func, typ, nE, dim):
        d = probs.index(typ)
        others = [j for j in range(3) if j != d]
        c_load = np.prod([load_functional(j, Q, Fs[j], meshes, param) for j in others])
        # everything that is subtracted: the lifted initial condition, then the stored modes
        known = [param["IC"]] + [[PGD_func[j][old] for j in range(3)] for old in range(nE)]
        if d == 1 and fd_time:
            l = c_load * param["M_t"] @ Q[1].vector()[:]
            for G in known:
                for term in (0, 1):
                    c = phys[term] * np.prod([functional(term, j, G[j], Fs[j], meshes, param) for j in others])
                    l = l - c * tmat(term, param) @ G[1].vector()[:]
            l[param["bc_idx"]] = 0
            return l
        l = fem.Constant(c_load) * Q[d] * v * fem.dx(meshes[d])
        for G in known:
            for term in (0, 1):
                c = phys[term] * np.prod([functional(term, j, G[j], Fs[j], meshes, param) for j in others])
                l += -fem.Constant(c) * own_form(term, d, G[d], v, meshes)
        return l

    spec = dict(name="1DHeatEqu-PGD-XTQ", name_coord=["X", "T", "Q"], modes_info=["T", "Node", "Scalar"], Vs=Vs,
                dom=0, bc_fct=bc_fct, load=load, param=param, rhs_fct=rhs_fct, lhs_fct=lhs_fct, probs=probs,
                seq_fp=np.arange(3), PGD_nmax=20)
    knobs = dict(stop_fp="norm", max_fp_it=50, tol_fp_it=1e-5, norm_modes="stiff", PGD_tol=1e-5)
    solve_modes = ["FEM", "FD", "FEM"] if fd_time else ["FEM", "FEM", "FEM"]
    MM = [0, param["M_t"], 0] if fd_time else []
    return spec, knobs, solve_modes, MM


def run(fem, PGDProblem, FD_matrices, fd_time=False, elems=(15, 10, 10)):
    spec, knobs, solve_modes, MM = build(fem, FD_matrices, elems, fd_time)
    p = PGDProblem(**spec)
    if fd_time:
        p.MM = MM
    for k, v in knobs.items():
        setattr(p, k, v)
    p.solve_PGD(_problem="linear", solve_modes=solve_modes)
    return p
