"""CPU oracle for the PGD fixed-point hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is product code.  Only ``tests/``, ``bench.py``'s
``cpu_baseline`` leg and ``__graft_entry__.smoke()`` may import it, and there
only as the checker (never as the thing measured or shipped).  The product
package ``pgdrome_amd`` never imports this package and fails loudly when its
HIP library is missing.

Contents
--------
fem_numpy.py     P1 simplex meshes, closed-form element atoms, CSR assembly,
                 Dirichlet elimination, Jacobi-PCG - the FEM arithmetic that the
                 reference delegates to FEniCS 2019.1.0 (third-party, absent from
                 /root/reference and from this image).
backend_numpy.py the oracle as a backend of the host-side form frontend, so the
                 host logic can be exercised on CPU in ``-m "not gpu"`` tests.
c/               plain-C (OpenMP) restatement of CSR SpMV / Jacobi-PCG used for
                 the timed CPU baseline at BASELINE.json's full sizes (c_oracle.py binds it,
                 cpu_baseline.py is bench.py's cpu_baseline leg).
The enrichment and fixed-point loops themselves (pgdrome/solver.py:306-506, 508-881) need no
restatement here: the reference's own file is imported and run on backend_numpy by
tests/golden/make_fixtures.py, and its results are the fixtures.

Parity pinning: the control flow is pinned by fixtures
captured from the reference's own ``solve_PGD`` (tests/golden/make_fixtures.py);
the FEM arithmetic is pinned by analytic known-answer values and by the values
the reference's tests hold (FD_matrices entries, analytic truss, "one mode"
Laplace).  Bit-level parity with FEniCS itself is UNPINNED: FEniCS cannot be
run in this image and the reference stores no golden vectors.
"""
