"""Oracle (TEST INFRASTRUCTURE): ctypes access to oracle/c/libpgd_oracle.so."""
from __future__ import annotations

import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent / "c"
_lib = None


def load():
    global _lib
    if _lib is None:
        so = HERE / "libpgd_oracle.so"
        if not so.exists() or so.stat().st_mtime < (HERE / "pgd_oracle.c").stat().st_mtime:
            subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
        lib = C.CDLL(str(so))
        PD, PI = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        lib.orc_num_threads.restype = C.c_int
        lib.orc_set_threads.argtypes = [C.c_int]
        lib.orc_set_threads.restype = None
        lib.orc_set_threads(usable_cpus())
        lib.orc_spmv.argtypes = [C.c_int64, PI, PI, PD, PD, PD]
        lib.orc_spmv.restype = None
        lib.orc_pcg_jacobi.argtypes = [C.c_int64, PI, PI, PD, PD, PD, C.c_double, C.c_double, C.c_int,
                                       C.POINTER(C.c_int), PD]
        lib.orc_pcg_jacobi.restype = C.c_int
        _lib = lib
    return _lib


def usable_cpus():
    """CPUs this process may run on: affinity mask capped by the cgroup CPU quota."""
    import os
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return n


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def num_threads():
    return load().orc_num_threads()


def spmv(rp, cols, vals, x):
    y = np.empty_like(x)
    load().orc_spmv(x.size, _i(rp), _i(cols), _d(vals), _d(x), _d(y))
    return y


def pcg_jacobi(rp, cols, vals, b, x0=None, rtol=1e-10, atol=0.0, maxit=10000):
    x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
    it, rel = C.c_int(), C.c_double()
    rc = load().orc_pcg_jacobi(b.size, _i(rp), _i(cols), _d(vals), _d(b), _d(x), rtol, atol, maxit,
                               C.byref(it), C.byref(rel))
    if rc != 0:
        raise MemoryError("orc_pcg_jacobi")
    return x, it.value, rel.value
