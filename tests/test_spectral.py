"""The spectral start space (pgdrome_amd/spectral.py): Ritz vectors harvested once per space and Dirichlet set join the
Galerkin start of every large SPD solve.  CPU: the logic on the oracle backend (harvest through Jacobi-PCG solves);
GPU: the harvest through the multigrid-preconditioned PCG, fewer Jacobi-PCG iterations, the same run."""
import numpy as np
import pytest

from pgdrome_amd import fem, problems, spectral
from pgdrome_amd.solver import PGDProblem

SET = {"linear_solver": "cg", "preconditioner": "jacobi", "relative_tolerance": 1e-10}


@pytest.fixture
def hip_backend():
    from pgdrome_amd.hip_backend import HipBackend
    old = fem._backend
    be = fem.set_backend(HipBackend(0))
    fem.clear_caches()
    yield be
    fem.set_backend(old)
    fem.clear_caches()


def _run(shape, k, nmax=4, n_mu=17):
    fem.clear_caches()
    P = fem.Point
    mesh = fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape)
    p = PGDProblem(**problems.reaction_diffusion(mesh, n_mu, PGD_nmax=nmax, PGD_tol=1e-12))
    i0, s0 = fem.STATS["pcg_iterations"], dict(spectral.STATS)
    p.solve_PGD(_problem="linear", settings=dict(SET, spectral_start=k) if k else SET)
    sp = [v for v in spectral._SPACES.values() if v is not None]
    inner = sum(v.info["inner_pcg_iterations"] for v in sp)
    modes = [[np.asarray(f.compute_vertex_values()).copy() for f in p.PGD_func[d]] for d in range(2)]
    return p, modes, fem.STATS["pcg_iterations"] - i0 - inner, {k_: spectral.STATS[k_] - s0[k_] for k_ in s0}, sp


def _same_run(pa, ma, pb, mb):
    assert [int(v) for v in pa.num_fp_it] == [int(v) for v in pb.num_fp_it] and pa.PGD_modes == pb.PGD_modes
    np.testing.assert_allclose(pa.amplitude, pb.amplitude, rtol=1e-7)
    np.testing.assert_allclose(pa.alpha, pb.alpha, rtol=1e-7)
    for d in range(2):
        for m in range(pa.PGD_modes):
            # (a relative error e in the earlier modes is an error e / amplitude[m] in mode m: solves to 1e-10)
            bar = max(1e-6, 1e-9 / max(pa.amplitude[m], 1e-300))
            sgn = 1.0 if float(ma[d][m] @ mb[d][m]) >= 0 else -1.0
            assert np.linalg.norm(ma[d][m] - sgn * mb[d][m]) <= bar * np.linalg.norm(mb[d][m]), (d, m)


def test_spectral_start_on_the_oracle_backend(monkeypatch):
    """Same run, fewer Jacobi-PCG iterations; the kept vectors are Ritz pairs of the first spatial operator with small residuals
    that vanish on the eliminated nodes; without a multigrid preconditioner the request is dropped unless told otherwise."""
    from oracle.backend_numpy import NumpyBackend
    old = fem._backend
    monkeypatch.setattr(spectral, "MIN_ROWS", 1000)
    try:
        fem.set_backend(NumpyBackend())
        p0, m0, it0, st0, _ = _run((19, 19, 19), 0)
        assert st0["harvests"] == 0 and st0["corrections"] == 0
        # (1) no multigrid here: dropped, the run is the plain run
        monkeypatch.delenv("PGD_SPECTRAL_ANY_SOLVER", raising=False)
        p1, m1, it1, st1, sp1 = _run((19, 19, 19), 6)
        assert st1["dropped_requests"] == 1 and st1["harvests"] == 0 and not sp1 and st1["corrections"] == 0
        _same_run(p1, m1, p0, m0)
        # (2) harvest through Jacobi-PCG solves
        monkeypatch.setenv("PGD_SPECTRAL_ANY_SOLVER", "1")
        p2, m2, it2, st2, sp2 = _run((19, 19, 19), 6)
        assert st2["harvests"] == 1 and st2["corrections"] == sum(int(v) for v in p2.num_fp_it)
        assert it2 < 0.92 * it0, (it2, it0)
        _same_run(p2, m2, p0, m0)
        sp = sp2[0]
        assert 1 <= sp.k <= 6 and all(r < spectral.RESIDUAL_BAR for r in sp.residuals) and sorted(sp.theta) == sp.theta
        V = p2.V[0]
        bverts = np.where(V.mesh().vertex_on_boundary())[0]
        for y in sp.Y:
            assert np.all(np.asarray(y.host())[bverts] == 0.0)
        # the lowest kept Ritz value is the lowest eigenvalue of the first spatial operator (free rows)
        from oracle import fem_numpy as F
        import scipy.sparse.linalg as spla
        c, e = V.mesh().coordinates(), V.mesh().cells()
        free = np.setdiff1d(np.arange(c.shape[0]), bverts)
        Fs = p2.get_Fsinit(p2.V, p2.bc, None)
        a_k = fem.assemble(Fs[1] * Fs[1] * fem.dx(p2.meshes[1]))
        a_m = fem.assemble(p2.param["mu"] * Fs[1] * Fs[1] * fem.dx(p2.meshes[1]))
        A = (a_k * F.assemble_atom(c, e, F.STIFF) + a_m * F.assemble_atom(c, e, F.MASS)).tocsr()[free][:, free]
        w = spla.eigsh(A.tocsc(), k=1, sigma=0, which="LM")[0]
        assert abs(sp.theta[0] - w[0]) <= 1e-8 * w[0]
        # (3) "auto": nothing before the space has seen AUTO_AFTER solves, then the harvest
        monkeypatch.setattr(spectral, "AUTO_AFTER", 4)
        monkeypatch.setattr(spectral, "AUTO_K", 5)
        fem.clear_caches()
        P = fem.Point
        p3 = PGDProblem(**problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), 19, 19, 19), 17, PGD_nmax=4, PGD_tol=1e-12))
        s0 = dict(spectral.STATS)
        p3.solve_PGD(_problem="linear", settings=dict(SET, spectral_start="auto"))
        solves = sum(int(v) for v in p3.num_fp_it)
        assert spectral.STATS["harvests"] - s0["harvests"] == 1 and spectral.STATS["corrections"] - s0["corrections"] == solves - 3
        assert [int(v) for v in p3.num_fp_it] == [int(v) for v in p0.num_fp_it]
        np.testing.assert_allclose(p3.amplitude, p0.amplitude, rtol=1e-7)
    finally:
        fem.set_backend(old) if old is not None else None
        fem.clear_caches()


@pytest.mark.gpu
def test_spectral_start_on_gpu(hip_backend):
    """64^3 x 17: the harvest runs through the V-cycle PCG (no Jacobi fallback), the Jacobi-PCG of the run needs fewer
    iterations, pass counts / amplitudes / modes are those of the run without it."""
    p0, m0, it0, st0, _ = _run((63, 63, 63), 0, nmax=5)
    mg0 = hip_backend.ctx.mg_stats()
    p1, m1, it1, st1, sp1 = _run((63, 63, 63), 12, nmax=5)
    mg1 = hip_backend.ctx.mg_stats()
    assert st1["harvests"] == 1 and st1["dropped_requests"] == 0 and len(sp1) == 1
    info = sp1[0].info
    assert info["lanczos_steps"] == 30 and mg1["solves"] - mg0["solves"] == 30 and mg1["fallbacks"] == mg0["fallbacks"]
    assert 6 <= sp1[0].k <= 12 and max(sp1[0].residuals) < spectral.RESIDUAL_BAR
    assert it1 < 0.9 * it0, (it1, it0)
    _same_run(p1, m1, p0, m0)
    print("spectral start at 64^3: Jacobi-PCG iterations %d -> %d, %d vectors, harvest %.3f s" % (it0, it1, sp1[0].k, info["seconds"]))
    fem.clear_caches()
