"""Build libpgd_amd.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m pgdrome_amd.build [--force]

The shared library lands in pgdrome_amd/lib/ so that it travels with the
repository snapshot to the GPU box; hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / "csrc"
LIBDIR = HERE / "lib"
LIB = LIBDIR / "libpgd_amd.so"
SOURCES = ["pgd_ctx.hip", "pgd_vec.hip", "pgd_mesh.hip", "pgd_spmv.hip", "pgd_pcg.hip", "pgd_comm.hip", "pgd_mg.hip", "pgd_krylov.hip"]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-result",
         "-fno-gpu-rdc"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not LIB.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*.hip")) + list(CSRC.glob("*.h")) + [HERE.parent / "include" / "pgd_amd.h"]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> Path:
    if not force and not _stale():
        return LIB
    LIBDIR.mkdir(exist_ok=True)
    objdir = LIBDIR / "obj"
    objdir.mkdir(exist_ok=True)
    hipcc = _hipcc()

    def compile_one(src: str) -> Path:
        obj = objdir / (src.replace(".hip", ".o"))
        # (HIPCC_EXTRA: extra flags of one-off instrumented builds, e.g. -DPGD_STENCIL_TIMING for tools/stencil_timing.py)
        cmd = [hipcc, *FLAGS, *os.environ.get("HIPCC_EXTRA", "").split(), "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
