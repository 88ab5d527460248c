#!/usr/bin/env python
"""bench.py - PGD fixed-point iterations/sec + SpMV HBM GB/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one pass of the alternating-directions fixed-point loop
(reference pgdrome/solver.py:531: one FEM assemble+solve per separated
dimension, norms and the stop test), executed by ``PGDProblem.solve_PGD`` on
the workload the metric is quoted on: 3-D space (256^3 P1 dofs) x 1-D
parameter (128 P1 dofs), reaction-diffusion, Jacobi-PCG at rtol 1e-10.
Inputs are synthetic, generated on the host and resident in HBM before the
timed region.  For N > 1 the spatial system is row-sharded into z-slabs (fixed
global size: strong scaling) with a halo exchange inside every SpMV and fp64
all-reduces for the dots, over torch.distributed (RCCL).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



class _Done(Exception):
    pass


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=256, help="spatial dofs per axis (256 = the metric's config)")
    ap.add_argument("--n-mu", type=int, default=128)
    ap.add_argument("--rtol", type=float, default=1e-10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-reduction", action="store_true",
                    help="with --dist-driver: force the Chronopoulos-Gear recurrence that N > 1 uses")
    ap.add_argument("--dist-driver", action="store_true",
                    help="N=1 only: run the row-sharded (host-driven, RCCL) solver path with one rank")
    ap.add_argument("--python-driver", action="store_true",
                    help="sharded runs: drive the PCG recurrence from Python over torch.distributed instead of "
                         "the in-library loop (pgd_pcg_solve_sharded over RCCL)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    return ap.parse_args()


def main():
    args = parse()
    # stdout carries exactly ONE line, the JSON result: RCCL prints its version banner to stdout when a
    # communicator is created, so everything else this process (and the libraries it loads) writes to
    # file descriptor 1 is sent to stderr, and the saved descriptor is used for the result line only
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N > 1 with torch.distributed.run (one process per GPU)")
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.dist_driver
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem

    # Sharded runs: the library enqueues on torch's CURRENT stream so that RCCL collectives (which torch
    # orders against the current stream) and the kernels form one sequence.  The legacy default stream
    # has handle 0, which the C ABI reads as "create your own", so make an explicit stream current.
    stream = None
    if sharded:
        tstream = torch.cuda.Stream(device=local_rank)
        torch.cuda.set_stream(tstream)
        stream = tstream.cuda_stream
        assert stream, "expected a non-default HIP stream"
    be = fem.set_backend(HipBackend(local_rank, stream))

    n = args.n
    t_setup = time.time()
    P = fem.Point
    if sharded:
        from pgdrome_amd import dist as pdist
        comm = pdist.TorchComm(dist, be, True if args.single_reduction else None, in_library=not args.python_driver)
        space = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
    else:
        space = fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
    spec = problems.reaction_diffusion(space, args.n_mu, PGD_nmax=50, PGD_tol=1e-12)
    prob = PGDProblem(**spec)
    settings = {"linear_solver": "cg", "preconditioner": "jacobi", "relative_tolerance": args.rtol}
    # one-time work outside the timed region: mesh upload + topology, the four atoms
    space.atom(fem.STIFF)
    space.atom(fem.MASS)
    be.sync()
    t_setup = time.time() - t_setup

    def barrier():
        be.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    state = {"t0": None, "t1": None, "its0": 0, "its1": 0}
    W, K = args.warmup, args.steps

    def hook(passes):
        if passes == W:
            barrier()
            be.prof_enable(2)        # time the PCG instance of k_spmv_csr only
            state["its0"] = fem.STATS["pcg_iterations"]
            state["t0"] = time.perf_counter()
        elif passes == W + K:
            barrier()
            state["t1"] = time.perf_counter()
            state["its1"] = fem.STATS["pcg_iterations"]
            raise _Done()

    prob.pass_hook = hook
    if W == 0:
        hook(0)
    try:
        # one solve_PGD gives ~2.5 passes per mode; should it converge before W+K passes have run, the
        # enrichment simply starts again (same work per pass) until the timed window is complete
        for _ in range(1000):
            prob.solve_PGD(_problem="linear", settings=settings)
    except _Done:
        pass
    if state["t1"] is None:
        raise SystemExit("only %d passes ran, fewer than warmup+steps=%d" % (prob.fp_passes, W + K))
    prof = be.prof_read()
    be.prof_enable(False)
    elapsed = state["t1"] - state["t0"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_sp = n ** 3
    from pgdrome_amd.sizes import nnz_p1_box
    nnz = nnz_p1_box(n)
    pcg_its = state["its1"] - state["its0"]
    achieved = prof["bytes"] / prof["seconds"] / 1e9 if prof["seconds"] > 0 else 0.0
    # which form of the CSR product ran: column ids decoded from the mesh's pattern dictionary, or streamed
    patterns = be.ctx.mesh_dict_count(space.handle())
    max_row = be.ctx.mesh_info(space.handle())["max_row"]
    spmv_kernel = "k_spmv_csr<dot,store,64>"
    if patterns:
        spmv_kernel = ("k_spmv_csr_dict16<dot,store>" if max_row <= 16 else "k_spmv_csr_dict<dot,store,64>") + \
            " (%d relative column patterns)" % patterns
    sym = be.ctx.mesh_sym_info(space.handle())
    sym_bytes = None
    if sym["slots"]:
        # the SPD solves read the operator from its symmetric half storage: every off-diagonal value once
        spmv_kernel = ("k_spmv_sym_grid3<dot,store> (symmetric half storage, %d slots/row, z-march over the %d x %d vertex "
                       "grid with the x planes in LDS, %d relative patterns)" % (sym["slots"], sym["nx"], sym["ny"], patterns)
                       if sym["nx"] else "k_spmv_sym<dot,store,%d> (symmetric half storage, %d relative patterns)" % (sym["slots"], patterns))
        rows_local = n_sp // world if sharded else n_sp
        sym_bytes = rows_local * (8 * sym["slots"] + 8 + 8 + 2)     # slot values + x + y + pattern id
    out = {
        "metric": "PGD fixed-point iters/sec + SpMV HBM GB/s, 256^3 P1 space x 1D param",
        "value": K / elapsed, "unit": "fixed-point iterations/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "cfg4: 3D-space %d^3 P1 (BoxMesh, 6 tets/cube) x 1D-parameter %d P1, "
                               "-Laplace(u)+mu*u=1, Jacobi-PCG rtol %g" % (n, args.n_mu, args.rtol),
                   "spatial_dofs": n_sp, "nnz": nnz, "parallelism": "z-slab row sharding x%d" % world if sharded else "single GPU",
                   "sharded_pcg_driver": (("in-library loop, RCCL" if comm.in_library == "rccl" else
                                           "python loop, torch.distributed") if sharded else None),
                   "pcg_iterations_per_step": pcg_its / K, "modes_completed": len(prob.num_fp_it),
                   "setup_seconds_untimed": t_setup},
        "roofline": {"bound": "hbm", "kernel": spmv_kernel, "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                     "frac": achieved / 8000.0, "traffic": None,
                     "launches": prof["launches"], "avg_launch_us": 1e6 * prof["seconds"] / max(prof["launches"], 1),
                     "algorithmic_bytes_per_launch": prof["bytes"] / max(prof["launches"], 1)},
    }
    if sym_bytes:
        # `achieved` / `frac` use the CSR byte formula of SURVEY 8d whatever form the kernel reads (so frac can
        # exceed 1: the symmetric storage moves about 0.45 x those bytes); the kernel's own minimum traffic and the
        # rate it reaches on THAT are given beside it
        avg = prof["seconds"] / max(prof["launches"], 1)
        out["roofline"]["kernel_min_bytes_per_launch"] = sym_bytes
        out["roofline"]["kernel_min_bytes_GBps"] = sym_bytes / avg / 1e9 if avg > 0 else 0.0
        out["roofline"]["kernel_min_bytes_frac"] = out["roofline"]["kernel_min_bytes_GBps"] / 8000.0
        out["roofline"]["note"] = ("achieved/frac are priced with the CSR byte formula 12 nnz + 20 n of SURVEY 8d as required; "
                                   "frac > 1 means the kernel does not move those bytes (symmetric half storage: each "
                                   "off-diagonal value once) - see traffic (PMC) and kernel_min_bytes_* for what it does move")
    pmc = os.path.join(ROOT, "profiles", "pmc_spmv_latest.json")
    if n == 256 and world == 1 and sym["nx"] and os.path.exists(pmc):
        # HBM bytes per launch from the separate rocprofv3 --pmc passes (FETCH_SIZE doubled per the gfx950
        # correction + WRITE_SIZE); collected with tools/pmc_spmv.py, not in this process
        with open(pmc) as f:
            out["roofline"]["traffic"] = json.load(f)["hbm_bytes_per_launch"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(prob, spec, be, pcg_its / K, args)
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if sharded:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(prob, spec, be, pcg_its_per_step, args):
    """The oracle (CPU restatement of the reference algorithm, not FEniCS) timed on the host
    cores on a BOUNDED sample of the same workload: the first spatial system of the run,
    `sample` Jacobi-PCG iterations; one step costs (PCG iterations per step) x that."""
    from oracle import cpu_baseline as cb
    return cb.run(prob, spec, be, pcg_its_per_step, args.cpu_seconds)


if __name__ == "__main__":
    main()
