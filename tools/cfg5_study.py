"""What is cfg5's fixed-point non-convergence?  The 4-way problem (space x time x two parameters) at a size the oracle backend
solves DIRECTLY (sparse LU) in seconds, on the HIP engine in several solver configurations and on the oracle backend:

    python tools/cfg5_study.py [n_space_cells=16] [modes=16]

Prints per configuration the pass counts, which modes ran into max_fp_it, and how far amplitudes / modes are from the oracle's
on the prefix before the first stalled mode and after it."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pgdrome_amd import fem, problems
from pgdrome_amd.solver import PGDProblem


def run(backend, n, modes, knobs=(), rtol=1e-10, start_from_modes=True, rescale=True):
    fem.set_backend(backend)
    fem.clear_caches()
    for k, v in knobs:
        backend.ctx.tune(k, v)
    P = fem.Point
    spec = problems.transient_heat(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), n, n, n), 17, 9, PGD_nmax=modes)
    p = PGDProblem(**spec)
    p.start_from_modes = start_from_modes
    old = fem.WARM_START_RESCALE
    fem.WARM_START_RESCALE = rescale
    try:
        p.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "preconditioner": "jacobi", "relative_tolerance": rtol})
    finally:
        fem.WARM_START_RESCALE = old
        for k, v in knobs:
            backend.ctx.tune(k, 1 if k != 7 else 0)
    return dict(num_fp_it=[int(v) for v in p.num_fp_it], amplitude=np.array(p.amplitude), err=np.array(p.err_fp_it),
                modes=[[f.compute_vertex_values() for f in p.PGD_func[d]] for d in range(p.num_pgd_var)])


def compare(name, a, ref):
    stall_a = [i for i, k in enumerate(a["num_fp_it"]) if k >= 50]
    stall_r = [i for i, k in enumerate(ref["num_fp_it"]) if k >= 50]
    first = min(stall_r + stall_a + [len(ref["num_fp_it"])])
    pre = slice(0, first)
    amp_pre = np.abs(a["amplitude"][pre] / ref["amplitude"][pre] - 1).max() if first else 0.0
    mode_pre = max([np.linalg.norm(a["modes"][d][m] - ref["modes"][d][m]) / np.linalg.norm(ref["modes"][d][m])
                    for d in range(4) for m in range(first)] or [0.0])
    nm = min(len(a["amplitude"]), len(ref["amplitude"]))
    amp_all = np.abs(a["amplitude"][:nm] / ref["amplitude"][:nm] - 1).max()
    print("%-44s passes %s\n%44s stalled modes %s (oracle %s); before the first stalled mode (%d modes): counts equal %s, amplitude %.1e, "
          "modes %.1e; all %d modes: amplitude %.1e" % (name, a["num_fp_it"], "", stall_a, stall_r, first,
                                                      a["num_fp_it"][pre] == ref["num_fp_it"][pre], amp_pre, mode_pre, nm, amp_all), flush=True)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    modes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    from oracle.backend_numpy import NumpyBackend
    from pgdrome_amd.hip_backend import HipBackend
    ref = run(NumpyBackend(), n, modes)
    print("oracle backend (direct solves)               passes %s stalled %s" % (ref["num_fp_it"], [i for i, k in enumerate(ref["num_fp_it"]) if k >= 50]))
    print("   err_fp_it of the stalled modes:", [float("%.2e" % ref["err"][i]) for i, k in enumerate(ref["num_fp_it"]) if k >= 50])
    hip = HipBackend(0)
    compare("HIP, defaults", run(hip, n, modes), ref)
    compare("HIP, rtol 1e-13", run(hip, n, modes, rtol=1e-13), ref)
    compare("HIP, no Galerkin start / no rescale", run(hip, n, modes, start_from_modes=False, rescale=False), ref)
    compare("HIP, textbook kernels (CSR, unscaled PCG)", run(hip, n, modes, knobs=((3, 0), (10, 0))), ref)
    compare("HIP, lattice assembly off", run(hip, n, modes, knobs=((20, 0),)), ref)
    compare("HIP, single-sync off (two reductions)", run(hip, n, modes, knobs=((18, 0),)), ref)


main()
