"""TEST INFRASTRUCTURE - the plumbing of an N-rank `bench.py` run rehearsed on a machine without GPUs.

    python tests/helpers/bench_rehearsal.py --gpus 2 --n 12 --n-mu 9 --steps 2 --warmup 1

Without WORLD_SIZE it calls bench.launch_ranks - the REAL launcher of bench.py (torch.distributed.run, free port, relay of rank 0's
line) - on THIS script; as a rank it does what bench.main does around the solves - process group, sharded mesh, the sharded solver
driver of pgdrome_amd/dist.py, the spectral start with its all-reduced vote, the timing window between barriers, the maximum over
the ranks, ONE line from rank 0 - over gloo, with the oracle backend doing the local arithmetic at a tiny size (which is why this
lives under tests/: bench.py itself touches oracle/ only in its cpu_baseline leg).  Its line says "data": "cpu rehearsal": never a
measurement."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench                                    # noqa: E402
from bench import _Done, _first_spatial_system, _free_port      # noqa: E402


def rehearse(args, world, rank, result_fd):
    import datetime
    import torch
    import torch.distributed as dist
    from oracle.backend_numpy import NumpyBackend
    from pgdrome_amd import dist as pdist, fem, problems, spectral
    from pgdrome_amd.solver import PGDProblem
    sharded = world > 1 or args.dist_driver
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    be = fem.set_backend(NumpyBackend())
    n = min(args.n, 24)
    P = fem.Point
    if sharded:
        comm = pdist.TorchComm(dist, be, True if args.single_reduction else None, in_library=False)
        space = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
    else:
        space = fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
    spec = problems.reaction_diffusion(space, min(args.n_mu, 17), PGD_nmax=50, PGD_tol=1e-12)
    prob = PGDProblem(**spec)
    settings = {"linear_solver": "cg", "preconditioner": args.preconditioner, "relative_tolerance": args.rtol}
    spectral_info = None
    if args.spectral_start > 0 and (not sharded or not args.no_spectral_sharded) and args.preconditioner == "jacobi":
        spectral.MIN_ROWS = 0
        settings["spectral_start"] = min(args.spectral_start, 6)
        sp, why = None, None
        try:
            A0, b0 = _first_spatial_system(prob)
            sp = spectral.get(fem, A0, b0, settings["spectral_start"], fem._Params(settings))
        except Exception as e:      # noqa: BLE001
            if not sharded:
                raise
            why = repr(e)[:300]
        if sharded:
            if not comm.allreduce_array([0.0 if sp is not None else 1.0])[0] == 0.0:
                spectral.clear()
                settings["spectral_start"] = 0
                sp = None
        spectral_info = {"vectors": sp.k if sp is not None else 0, "error": why}
    W, K = args.warmup, args.steps
    state = {"t0": None, "t1": None, "i0": 0, "i1": 0}

    def barrier():
        if world > 1:
            dist.barrier()

    def hook(passes):
        if passes == W:
            barrier()
            state["i0"], state["t0"] = fem.STATS["pcg_iterations"], time.perf_counter()
        elif passes == W + K:
            barrier()
            state["t1"], state["i1"] = time.perf_counter(), fem.STATS["pcg_iterations"]
            raise _Done()
    prob.pass_hook = hook
    if W == 0:
        hook(0)
    try:
        for _ in range(1000):
            prob.solve_PGD(_problem="linear", settings=settings)
    except _Done:
        pass
    elapsed = state["t1"] - state["t0"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    out = {"metric": "PGD fixed-point iters/sec + SpMV HBM GB/s, 256^3 P1 space x 1D param", "value": K / elapsed,
           "unit": "fixed-point iterations/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * elapsed / K,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "cpu rehearsal",
           "config": {"workload": "REHEARSAL of the run's plumbing on the CPU (gloo, oracle backend), %d^3 x %d: not a measurement" % (n, min(args.n_mu, 17)),
                      "parallelism": "z-slab row sharding x%d" % world if sharded else "single process",
                      "pcg_iterations_per_step": (state["i1"] - state["i0"]) / K, "modes_completed": len(prob.num_fp_it),
                      "spectral_start": spectral_info, "preconditioner": args.preconditioner,
                      "sharded_v_cycle_solves": (comm.stats.get("sharded_mg_solves", 0) if sharded else None)}}
    if rank == 0:
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if sharded:
        dist.barrier()
        dist.destroy_process_group()




def main():
    args = bench.parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(bench.launch_ranks(args, sys.argv[1:], script=__file__, need_gpus=False))
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench_rehearsal.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    rehearse(args, world, rank, result_fd)


if __name__ == "__main__":
    main()
