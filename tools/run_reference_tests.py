"""Run the REFERENCE's own test modules with this repository's frontend standing in for dolfin.

    python tools/run_reference_tests.py [--backend oracle|hip] [module ...]

Build container only (needs /root/reference).  The modules under /root/reference/tests are loaded
unchanged; `dolfin` resolves to pgdrome_amd.fem (numpy oracle backend by default, the HIP engine with
--backend hip), `pgdrome.solver` / `pgdrome.model` are the reference's own modules.  Prints one line
per test method.  Nothing of the reference is copied: this only executes it.
"""
import argparse
import importlib.util
import io
import logging
import os
import sys
import time
import types
import unittest
import warnings
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

DEFAULT = ["tests/unit/test_autotest.py", "tests/unit/test_FD.py", "tests/unit/test_pgdclass.py",
           "tests/integration/test_heat1D.py", "tests/integration/test_heat1D_dimless.py",
           "tests/integration/test_elastic.py", "tests/integration/test_laplace.py",
           "tests/integration/test_solver_problem.py", "tests/unit/test_pgdclass_dolfin.py"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="oracle")
    ap.add_argument("modules", nargs="*", default=DEFAULT)
    args = ap.parse_args()
    from pgdrome_amd import fem
    if args.backend == "oracle":
        from oracle.backend_numpy import NumpyBackend
        fem.set_backend(NumpyBackend(direct_above=20000))     # test_solver_problem: 131 k-dof elasticity systems
    else:
        from pgdrome_amd.hip_backend import HipBackend
        fem.set_backend(HipBackend(0))
    sys.modules["dolfin"] = fem
    sys.modules["fenics"] = fem
    from pgdrome_amd import h5lite
    sys.modules["h5py"] = h5lite          # File(path, "r").get(name) -> array-like: what pgdrome/model.py uses of h5py
    logging.disable(logging.CRITICAL)
    warnings.filterwarnings("ignore")
    results = []
    for rel in args.modules:
        path = os.path.join("/root/reference", rel)
        name = "ref_" + os.path.basename(rel)[:-3]
        try:
            spec = importlib.util.spec_from_file_location(name, path)
            mod = importlib.util.module_from_spec(spec)
            with redirect_stdout(io.StringIO()):
                spec.loader.exec_module(mod)
        except Exception as e:                                   # module does not even import
            results.append((rel, "<import>", "ERROR", "%s: %s" % (type(e).__name__, e), 0.0))
            continue
        suite = unittest.defaultTestLoader.loadTestsFromModule(mod)
        for case in _flatten(suite):
            t0 = time.time()
            res = unittest.TestResult()
            with redirect_stdout(io.StringIO()):
                case.run(res)
            dt = time.time() - t0
            if res.errors or res.failures:
                tb = (res.errors or res.failures)[0][1].strip().splitlines()
                results.append((rel, case.id().split(".")[-1], "FAIL" if res.failures else "ERROR", tb[-1][:160], dt))
            elif res.skipped:
                results.append((rel, case.id().split(".")[-1], "SKIP", res.skipped[0][1], dt))
            else:
                results.append((rel, case.id().split(".")[-1], "PASS", "", dt))
    for rel, test, status, msg, dt in results:
        print("%-6s %-46s %-32s %6.1fs  %s" % (status, rel, test, dt, msg))
    print("passed %d of %d" % (sum(r[2] == "PASS" for r in results), len(results)))


def _flatten(suite):
    for item in suite:
        if isinstance(item, unittest.TestSuite):
            yield from _flatten(item)
        else:
            yield item


if __name__ == "__main__":
    main()
