#!/usr/bin/env python
"""bench.py - PGD fixed-point iterations/sec + SpMV HBM GB/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
        N > 1 without a launcher around it (WORLD_SIZE unset): bench.py starts its own N ranks as child processes of
        `python -m torch.distributed.run` on 127.0.0.1 at a free port, before it touches HIP, and relays rank 0's line
        (`launch_ranks`; PGD_BENCH_FORCE_LAUNCHER=1 takes the same route at N = 1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
        the ranks themselves (what the driver's own launcher, or launch_ranks, runs)

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
    ap.add_argument("--spectral-start", type=int, default=48,
                    help='settings["spectral_start"]: Ritz vectors in the second level of the Galerkin start of the spatial solves '
                         "(pgdrome_amd/spectral.py; harvested once, outside the timed region, reported in config.spectral_start); 0: off")
    ap.add_argument("--no-spectral-sharded", action="store_true",
                    help="N > 1 (or --dist-driver): do NOT harvest / use the spectral start space on the row-sharded mesh (by default every N runs "
                         "the same algorithm - the harvest's solves go through the sharded V-cycle, dist.pcg_mg, driven from the host; a harvest "
                         "that fails on any rank switches it off on all of them and the run goes on as the plain sharded Jacobi-PCG)")
    ap.add_argument("--preconditioner", default="jacobi",
                    help='settings["preconditioner"] of the timed run: "jacobi" (the metric\'s Jacobi-PCG) or "amg" (the V-cycle of pgd_mg.hip; '
                         "on a sharded run the slab form of it, dist.pcg_mg) - a side measurement, never the headline")
    ap.add_argument("--direct-halo", action="store_true",
                    help="N > 1: the boundary planes of the PCG loop's search direction go straight into the neighbours' ghost planes "
                         "through IPC-mapped pointers (PGD_HALO_DIRECT=1, pgd_comm_push_*) instead of RCCL send / receive; opt-in - "
                         "never run between two different GPUs so far")
    ap.add_argument("--no-direct-probe", action="store_true",
                    help="N > 1 without --direct-halo: do not attach and check the direct halo / all-reduce (they are only CHECKED by default - "
                         "the solves use RCCL - so that a run on real multi-GPU hardware says whether they would work there)")
    ap.add_argument("--share-one-gpu", action="store_true",
                    help="REHEARSAL on a one-GPU box, never a measurement: the N ranks all use GPU 0, the exchange steps go through gloo "
                         "(RCCL refuses two ranks on one device) and the library's sharded loop through its callback binding; the "
                         "line says so in `data`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-reduction", action="store_true",
                    help="with --dist-driver: force the Chronopoulos-Gear recurrence that N > 1 uses")
    ap.add_argument("--dist-driver", action="store_true",
                    help="N=1 only: run the row-sharded (host-driven, RCCL) solver path with one rank")
    ap.add_argument("--python-driver", action="store_true",
                    help="sharded runs: drive the PCG recurrence from Python over torch.distributed instead of "
                         "the in-library loop (pgd_pcg_solve_sharded over RCCL)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the CPU baseline sample")
    ap.add_argument("--spmv-variant", type=int, default=-1, help="PGD_TUNE_SPMV_VARIANT for A/B runs (-1: library default)")
    ap.add_argument("--single-sync", type=int, default=-1, help="PGD_TUNE_PCG_SINGLE_SYNC for A/B runs (-1: library default)")
    ap.add_argument("--unit-diag", type=int, default=-1, help="PGD_TUNE_UNIT_DIAG for A/B runs (-1: library default)")
    ap.add_argument("--defer-x", type=int, default=-1, help="PGD_TUNE_PCG_DEFER_X for A/B runs (-1: library default)")
    ap.add_argument("--no-pmc", action="store_true", help="do not start rocprofv3 --pmc child processes for roofline.traffic")
    ap.add_argument("--no-csr-section", action="store_true", help="skip the timed CSR products (roofline.csr_product)")
    ap.add_argument("--no-general-paths", action="store_true",
                    help="skip the passes without the row-class dictionary / on the CSR kernels (config.general_paths)")
    ap.add_argument("--comm-timeout", type=float, default=float(os.environ.get("PGD_COMM_TIMEOUT_S", "60")),
                    help="N > 1: deadline (s) of every host-side wait of the sharded solve and of torch.distributed's collectives")
    ap.add_argument("--watchdog-seconds", type=float, default=1500.0,
                    help="N > 1: the whole process exits non-zero with all Python stacks on stderr after this long (0: off)")
    return ap.parse_args()


_T0 = time.time()


def _note(what):
    """One progress line on stderr (seconds since this process started): a long run shows where it is."""
    sys.stderr.write("bench.py: [%6.1f s] %s\n" % (time.time() - _T0, what))
    sys.stderr.flush()


def _free_port():
    """A TCP port nobody listens on right now (127.0.0.1): back-to-back runs on one node must not meet in TIME_WAIT."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _visible_gpus():
    """GPUs this process may use.  torch.cuda.device_count() reads the driver's device list without creating a HIP context
    (on this image), so the launcher below has not touched a GPU when it starts its children.  PGD_BENCH_ASSUME_GPUS overrides
    the count (tests of the launcher on a machine without a GPU)."""
    forced = os.environ.get("PGD_BENCH_ASSUME_GPUS")
    if forced:
        return int(forced)
    import torch
    return int(torch.cuda.device_count())


def launch_ranks(args, argv, script=None, need_gpus=True):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset - exactly how the driver starts N = 1): start
    the N ranks HERE, as CHILD processes of `python -m torch.distributed.run` (never an exec: this process stays what it is),
    one per GPU, rendezvous on 127.0.0.1 at a free port; relay the ONE JSON line rank 0 prints to our stdout and leave with the
    launcher's status.  Nothing in this function - or before it in main() - makes a HIP call; deadlines and the whole-process
    watchdog live in the children.  PGD_BENCH_LAUNCHER replaces the launcher command (tests: a stub)."""
    import shlex
    import subprocess
    n = args.gpus
    have = _visible_gpus() if need_gpus and not getattr(args, "share_one_gpu", False) else n          # (need_gpus False, another script: tests/helpers/bench_rehearsal.py)
    if have < n:
        sys.stderr.write("bench.py: --gpus %d asked for but %d GPU(s) visible to this process - not starting any rank\n" % (n, have))
        return 2
    port = _free_port()
    launcher = os.environ.get("PGD_BENCH_LAUNCHER")
    head = shlex.split(launcher) if launcher else [sys.executable, "-m", "torch.distributed.run"]
    # ("--": everything behind it is the script and ITS arguments - the launcher's parser would otherwise read an argument of ours
    # that abbreviates one of its own options, e.g. `--n`, as that option)
    cmd = head + ["--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port), "--",
                  os.path.abspath(script or __file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PGD_BENCH_LAUNCHED="1")
    env.pop("PGD_BENCH_FORCE_LAUNCHER", None)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env, cwd=ROOT)
    line = None
    for raw in child.stdout:                      # rank 0's result line; whatever else reaches the launcher's stdout goes to stderr
        text = raw.decode(errors="replace")
        if text.startswith("{") and '"metric"' in text:
            line = text
        else:
            sys.stderr.write(text)
    rc = child.wait()
    if line is not None:
        sys.stdout.write(line if line.endswith("\n") else line + "\n")
        sys.stdout.flush()
    elif rc == 0:
        sys.stderr.write("bench.py: the ranks ended with status 0 but printed no result line\n")
        rc = 1
    return rc


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or os.environ.get("PGD_BENCH_FORCE_LAUNCHER") == "1"):
        # before `import torch.distributed`, before any HIP call: this process only starts and relays
        raise SystemExit(launch_ranks(args, sys.argv[1:]))
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
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: one process per GPU (run `python bench.py --gpus N`, which starts "
                         "its own ranks, or torch.distributed.run with --nproc-per-node N)" % (args.gpus, world))
    if args.share_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    sharded = world > 1 or args.dist_driver
    if sharded:
        import datetime
        import faulthandler
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        os.environ["PGD_COMM_TIMEOUT_S"] = repr(args.comm_timeout)       # read by the library when it binds its communicator
        if args.watchdog_seconds > 0:
            # nothing of a multi-process run may hang silently: a rendezvous or a bind that never returns ends HERE, with every
            # thread's Python stack on stderr and a non-zero status (a fresh exit of this process; nothing is re-executed)
            faulthandler.dump_traceback_later(args.watchdog_seconds, exit=True)
        if args.share_one_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(4.0 * args.comm_timeout, 120.0)))
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank),
                                    timeout=datetime.timedelta(seconds=max(4.0 * args.comm_timeout, 120.0)))

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
    if args.spmv_variant >= 0:
        be.ctx.tune(13, args.spmv_variant)
    if args.defer_x >= 0:
        be.ctx.tune(16, args.defer_x)
    if args.unit_diag >= 0:
        be.ctx.tune(17, args.unit_diag)
    if args.single_sync >= 0:
        be.ctx.tune(18, args.single_sync)

    n = args.n
    _note("library bound; setup")
    t_setup = time.time()
    P = fem.Point
    if sharded:
        from pgdrome_amd import dist as pdist
        if args.direct_halo:
            os.environ["PGD_HALO_DIRECT"] = "1"         # read by pdist.sharded_box_mesh

        comm = pdist.TorchComm(dist, be, True if args.single_reduction else None, in_library=not args.python_driver)
        space = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
    else:
        space = fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), n - 1, n - 1, n - 1)
    spec = problems.reaction_diffusion(space, args.n_mu, PGD_nmax=50, PGD_tol=1e-12)
    prob = PGDProblem(**spec)
    settings = {"linear_solver": "cg", "preconditioner": args.preconditioner, "relative_tolerance": args.rtol}
    # one-time work outside the timed region: mesh upload + topology, the four atoms
    space.atom(fem.STIFF)
    space.atom(fem.MASS)
    be.sync()
    t_setup = time.time() - t_setup
    # ... and, on request, the spectral start space of the spatial solves: Ritz vectors of the first spatial operator, harvested
    # once per space and Dirichlet set (one-time work like the atoms; its seconds are reported on their own)
    spectral_info = None
    if args.spectral_start > 0 and (not sharded or not args.no_spectral_sharded) and args.preconditioner == "jacobi":
        from pgdrome_amd import spectral
        settings["spectral_start"] = args.spectral_start
        t_h = time.time()
        _note("harvest of the spectral start space")
        sp, why = None, None
        try:
            A0, b0 = _first_spatial_system(prob)
            sp = spectral.get(fem, A0, b0, args.spectral_start, fem._Params(settings))
            be.sync()
            del A0, b0
        except Exception as e:      # noqa: BLE001 - the run must go on without it (N > 1: the harvest is the only host-driven solver loop of the run)
            why = repr(e)[:300]
            spectral.clear()
            settings["spectral_start"] = 0
            sys.stderr.write("bench.py: rank %d: the spectral start space could not be harvested (%s): the run goes on without it\n" % (rank, why))
        if sharded:
            # every rank uses it or none does (a rank-local failure above must not leave the ranks with different start vectors:
            # the iteration counts, and with them the collectives of the solves, would differ)
            ok = comm.allreduce_array([0.0 if sp is not None else 1.0])[0] == 0.0
            if not ok:
                spectral.clear()
                settings["spectral_start"] = 0
                sp = None
        spectral_info = ({"vectors": sp.k, "asked": args.spectral_start, "lanczos_steps": sp.info["lanczos_steps"],
                          "inner_pcg_iterations_of_the_harvest": sp.info["inner_pcg_iterations"],
                          "harvest_seconds_untimed": time.time() - t_h, "hbm_bytes": 8.0 * sp.k * n ** 3,
                          "ritz_values_over_lowest": [t / sp.theta[0] for t in sp.theta],
                          "worst_relative_residual": max(sp.residuals),
                          "note": "settings[\"spectral_start\"]: second level of the Galerkin start of every spatial solve, x0 += Y (Y'AY)^-1 "
                                  "Y'(b - A x0) over Ritz vectors of the first spatial operator (inverse Lanczos from the first right-hand "
                                  "side through multigrid-PCG solves, once); config.without_spectral_start is the same workload without it"}
                         if sp is not None else {"vectors": 0, "asked": args.spectral_start, "note": "not available on this system", "error": why})

    _note("warm-up and timed passes")

    def barrier():
        be.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    state = {"t0": None, "t1": None, "its0": 0, "its1": 0, "pcg0": 0.0, "pcg1": 0.0}
    kc0 = kc1 = None
    W, K = args.warmup, args.steps

    def hook(passes):
        nonlocal kc0, kc1
        if passes == W:
            barrier()
            be.prof_enable(2)        # HIP events on the PCG instance of the product (fused dot, y stored) only
            if sharded:
                be.comm_prof(1)      # phase timing of the sharded loop: one iteration per chunk between HIP events
            kc0 = be.ctx.kernel_counts()
            state["its0"] = fem.STATS["pcg_iterations"]
            state["pcg0"] = fem.STATS["pcg_seconds"]
            state["ar0"] = comm.stats.get("allreduce_host", 0) if sharded else 0
            state["t0"] = time.perf_counter()
        elif passes == W + K:
            barrier()
            state["t1"] = time.perf_counter()
            state["ar1"] = comm.stats.get("allreduce_host", 0) if sharded else 0
            state["its1"] = fem.STATS["pcg_iterations"]
            state["pcg1"] = fem.STATS["pcg_seconds"]
            kc1 = be.ctx.kernel_counts()
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
    comm_phases = None
    if sharded:
        ph = be.comm_prof(0)
        ns = max(ph.pop("samples"), 1.0)
        host_wait = ph.pop("host_boundary_wait")
        mine = [1e6 * ph[k] / ns for k in sorted(ph)] + [1e6 * host_wait / max(state["its1"] - state["its0"], 1), ns]
        rows = [None] * world
        if world > 1:
            dist.all_gather_object(rows, mine)
        else:
            rows = [mine]
        names = ["%s_us" % k for k in sorted(ph)] + ["host_wait_at_chunk_boundaries_us_per_iteration", "samples"]
        comm_phases = {nm: {"max": max(r[i] for r in rows), "min": min(r[i] for r in rows)} for i, nm in enumerate(names)}
        comm_phases["note"] = ("per rank: HIP events around the phases of ONE iteration per chunk of 16 (compute stream); halo_wait = the "
                               "compute stream waiting for the ghost planes AFTER the interior rows' product (exposed halo time), allreduce "
                               "includes waiting for the slowest rank; max / min over the ranks")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_sp = n ** 3
    from pgdrome_amd.sizes import nnz_p1_box
    nnz = nnz_p1_box(n)
    pcg_its = state["its1"] - state["its0"]
    pcg_seconds = state["pcg1"] - state["pcg0"]
    launches = max(prof["launches"], 1)
    # event times minus what a pair of events adds to the kernel it brackets (calibrated by the library: t(n kernels) = overhead + n t)
    ev_over = prof.get("event_overhead", 0.0)
    avg_raw = prof["seconds"] / launches
    avg = max(avg_raw - ev_over, 1e-9) if prof["launches"] else 0.0
    rows_local = n_sp // world if sharded else n_sp
    # which product ran in the SPD solves (launch counters of the library, not a guess)
    kc = {k: kc1[k] - kc0[k] for k in kc1}
    patterns = be.ctx.mesh_dict_count(space.handle())
    sym = be.ctx.mesh_sym_info(space.handle())
    ran = max(("stencil_march", "diac_march", "dia_march", "dia_rows", "sym_rows", "csr_dict", "csr"), key=lambda k: kc[k])
    kernel_names = {
        "stencil_march": "k_spmv_stencil_march<dot,store> (the row-class dictionary of the scaled operator reduced to ONE stencil + eliminated "
                         "nodes, every row verified bit by bit: 8 couplings in scalar registers, a 64 x 16 patch of the %d x %d vertex grid "
                         "marching along z, four rows per thread, x planes in LDS; the class byte is read only where a plane's codes differ "
                         "from the plane below: 16 B per row)" % (sym["nx"], sym["ny"]),
        "diac_march": "k_spmv_diac_march2<dot,store> (symmetric half storage in diagonal form behind a lossless row-class dictionary: "
                      "one code byte per row, the classes' 8-tuples of slot values in LDS / registers; a 64 x 8 patch of the %d x %d "
                      "vertex grid marching along z, two rows per thread, x planes in LDS)" % (sym["nx"], sym["ny"]),
        "dia_march": "k_spmv_dia_march2<dot,store> (symmetric half storage in diagonal form: 8 slot arrays of n doubles, a 64 x 8 "
                     "patch of the %d x %d vertex grid marching along z, two rows per thread, x planes and the plane-below "
                     "couplings in LDS)" % (sym["nx"], sym["ny"]),
        "dia_rows": "k_spmv_dia_rows<dot,store> (symmetric half storage in diagonal form, row order)",
        "sym_rows": "k_spmv_sym<dot,store,%d> (symmetric half storage, %d relative patterns)" % (sym["slots"], patterns),
        "csr_dict": "k_spmv_csr_dict16<dot,store> (CSR values, column ids from %d relative patterns)" % patterns,
        "csr": "k_spmv_csr<dot,store,64>",
    }
    own = prof["own_bytes"] / launches                       # least bytes that kernel must move, per launch
    alg = prof["bytes"] / launches                           # 12 nnz + 20 n for the rows covered
    achieved = own / avg / 1e9 if avg > 0 else 0.0
    # the vector update of the single-sync recurrence (r, p and - every other launch - x in one kernel: 40 / 56 B per row)
    upd_n = prof.get("update_launches", 0)
    upd_avg_raw = prof["update_seconds"] / upd_n if upd_n else 0.0
    upd_avg = max(upd_avg_raw - ev_over, 1e-9) if upd_n else 0.0
    upd_bytes = prof["update_bytes"] / upd_n if upd_n else 0.0
    upd_achieved = upd_bytes / upd_avg / 1e9 if upd_avg > 0 else 0.0
    it_us = 1e6 * pcg_seconds / max(pcg_its, 1)
    out = {
        "metric": "PGD fixed-point iters/sec + SpMV HBM GB/s, 256^3 P1 space x 1D param",
        "value": K / elapsed, "unit": "fixed-point iterations/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not args.share_one_gpu else "synthetic; REHEARSAL, not a measurement: %d ranks share ONE GPU over gloo" % world,
        "config": {"workload": "cfg4: 3D-space %d^3 P1 (BoxMesh, 6 tets/cube) x 1D-parameter %d P1, "
                               "-Laplace(u)+mu*u=1, Jacobi-PCG rtol %g" % (n, args.n_mu, args.rtol),
                   "spatial_dofs": n_sp, "nnz": nnz, "parallelism": "z-slab row sharding x%d" % world if sharded else "single GPU",
                   "preconditioner": args.preconditioner,
                   "sharded_pcg_driver": (("in-library loop, RCCL" if comm.in_library == "rccl" else
                                           "in-library loop, exchange steps called back into torch.distributed" if comm.in_library == "callbacks" else
                                           "python loop, torch.distributed") if sharded else None),
                   "sharded_v_cycle_solves": (comm.stats.get("sharded_mg_solves", 0) if sharded else None),
                   "halo_overlap": (bool(be.comm_overlap(-2)) if sharded and getattr(comm, "in_library", None) == "rccl" else (False if sharded else None)),
                   "halo_overlap_available": (bool(getattr(comm, "halo_overlap", False)) if sharded else None),
                   "direct_halo": ({"attached_on_every_rank": bool(getattr(comm, "direct_halo", False)),
                                    "used_by_the_last_solve": bool(be.comm_push(-2)) if comm.in_library else False,
                                    "direct_allreduce_attached_on_every_rank": bool(getattr(comm, "direct_allreduce", False)),
                                    "direct_allreduce_used_by_the_last_solve": bool(be.comm_allreduce_direct(-2)) if comm.in_library else False,
                                    "probe": getattr(comm, "direct_probe", None)}
                                   if sharded else None),
                   "rccl_world": (be.comm_info()["world"] if sharded and comm.in_library == "rccl" else
                                  (dist.get_world_size() if sharded else None)),
                   "pcg_iterations_per_step": pcg_its / K, "modes_completed": len(prob.num_fp_it),
                   "us_per_pcg_iteration": 1e6 * pcg_seconds / max(pcg_its, 1),
                   "pcg_iteration_breakdown_us": ({"product": 1e6 * avg, "vector_update": 1e6 * upd_avg,
                                                   "scalars_kernel_launch_gaps_and_solve_setup": it_us - 1e6 * (avg + upd_avg)} if upd_n else
                                                  {"product": 1e6 * avg, "vector_kernels_reductions_and_solve_setup": it_us - 1e6 * avg}),
                   "seconds_in_pcg_solves": pcg_seconds, "seconds_timed": elapsed,
                   "product_launches_by_kernel": kc,
                   "launch_timing_samples_dropped_as_noops": prof.get("dropped_noop_samples"),
                   "row_class_classifications": be.ctx.classify_counts(),
                   # (inside the timed window: the harvest of the spectral start space, before it, issues hundreds)
                   "host_synchronised_allreduces_per_step": (((state.get("ar1", 0) - state.get("ar0", 0)) / max(K, 1)) if sharded else None),
                   "sharded_iteration_phases": comm_phases,
                   "comm_timeout_s": (args.comm_timeout if sharded else None),
                   "allreduces_outside_the_pcg_loop": (comm.stats.get("allreduce") if sharded else None),
                   "spectral_start": spectral_info,
                   "setup_seconds_untimed": t_setup},
        "roofline": {"bound": "hbm", "kernel": kernel_names[ran],
                     # PHYSICAL pricing: the bytes this kernel must move in the storage form it reads (diagonal form:
                     # 8 slots x 8 B + x + y = 80 B per row) / its average launch time (HIP events, this run)
                     "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                     "launches": prof["launches"], "avg_launch_us": 1e6 * avg,
                     "avg_launch_us_raw_events": 1e6 * avg_raw, "event_pair_overhead_us": 1e6 * ev_over,
                     "bytes_per_launch": own, "bytes_per_row": own / max(rows_local, 1),
                     # for orientation: the same rate against the chip's measured streaming ceiling (float4 copy 6.29 TB/s,
                     # MI355X_MICROARCH.md) instead of the 8 TB/s specification that `peak` / `frac` use
                     "measured_copy_ceiling_GBps": 6290.0, "frac_of_measured_copy_ceiling": achieved / 6290.0,
                     # SURVEY 8d's CSR formula priced on the same launch time: NOT a physical rate for this kernel (it does
                     # not stream the CSR arrays) - the CSR kernels' own measurement is in csr_product below
                     "csr_formula_bytes_per_launch": alg, "csr_formula_equivalent_GBps": alg / avg / 1e9 if avg > 0 else 0.0},
    }
    _note("timed region over: %.3f passes/s" % out["value"])
    if rank == 0 and world == 1 and not args.no_csr_section:
        out["roofline"]["csr_product"] = csr_section(be, prob, n_sp, nnz)
    general = None
    if rank == 0 and world == 1 and not sharded and not args.no_general_paths and ran in ("diac_march", "stencil_march"):
        import numpy as np
        n_done = min(len(prob.num_fp_it), min(len(f) for f in prob.PGD_func)) if getattr(prob, "PGD_func", None) else 0
        modes_ref = [[np.asarray(f[k].vector()[:]).copy() for f in prob.PGD_func] for k in range(n_done)]
        # (without the spectral start: its harvest leans on the multigrid structure, which the operators these paths stand for -
        # natural boundaries, variable coefficients, meshes without a lattice - do not have)
        _note("side section: general paths")
        general = general_paths(be, spec, dict(settings, spectral_start=0))
        out["config"]["general_paths"] = general
        _note("side section: multigrid preconditioner")
        out["config"]["multigrid_preconditioner"] = multigrid_path(be, spec, dict(settings, spectral_start=0), modes_ref)
        if settings.get("spectral_start"):
            _note("side section: the same passes without the spectral start")
            out["config"]["without_spectral_start"] = plain_start_path(be, spec, dict(settings, spectral_start=0), W, K)
    if rank == 0 and world == 1 and not sharded and not args.no_general_paths and n == 256:
        _note("side section: a rank with ghost planes (child process)")
        out["config"]["rank_with_ghost_planes"] = ghost_rank_rehearsal()
        _note("side section: the sharded V-cycle on one rank (child process)")
        out["config"]["sharded_v_cycle_one_rank"] = sharded_v_cycle_rank(args)
    _note("counters")
    pmc = pmc_traffic(own, upd_bytes) if rank == 0 and world == 1 and n == 256 and sym["nx"] and not args.no_pmc else {}
    out["roofline"].update(pmc.get("product", {}))
    if upd_n and upd_avg > avg:
        # The product no longer takes most of an iteration: `roofline` is the DOMINANT kernel's - the vector update - and the
        # product's keeps its fields, unchanged, under roofline.spmv.
        spmv = out["roofline"]
        out["roofline"] = {"bound": "hbm",
                           "kernel": "k_pcg1_update (single-sync recurrence: r -= alpha q, p = r + beta p, partial sums of r.r, and x += alpha p "
                                     "as a two-term update in every other launch; 16-byte accesses; 3 vectors read + 2 written, or 4 + 3: "
                                     "averages over both kinds of launch)",
                           "achieved": upd_achieved, "peak": 8000.0, "unit": "GB/s", "frac": upd_achieved / 8000.0, "traffic": None,
                           "launches": upd_n, "avg_launch_us": 1e6 * upd_avg, "avg_launch_us_raw_events": 1e6 * upd_avg_raw,
                           "event_pair_overhead_us": 1e6 * ev_over, "bytes_per_launch": upd_bytes,
                           "bytes_per_row": upd_bytes / max(rows_local, 1),
                           "measured_copy_ceiling_GBps": 6290.0, "frac_of_measured_copy_ceiling": upd_achieved / 6290.0,
                           "share_of_pcg_iteration": 1e6 * upd_avg / it_us if it_us > 0 else None,
                           "note": "of this kernel's streams p (134 MB at 256^3; written here, read by the product, read here again) is "
                                   "served by the 256 MiB Infinity Cache, and FETCH_SIZE / WRITE_SIZE count those hits: `achieved` is a rate "
                                   "relative to the HBM peak, not bytes that all came from HBM - it may exceed the chip's measured copy "
                                   "ceiling (measured_copy_ceiling_GBps)"}
        out["roofline"].update(pmc.get("update", {}))
        spmv["share_of_pcg_iteration"] = 1e6 * avg / it_us if it_us > 0 else None
        if ran in ("diac_march", "stencil_march"):
            # the same product in the plain diagonal form streams 72 B per row (k_spmv_dia_march2): timed in THIS run by
            # config.general_paths.plain_march (a variable coefficient or a graded mesh takes that path)
            spmv["plain_diagonal_form_bytes_per_row"] = 72.0
            pm = (general or {}).get("plain_march", {}).get("product_us")
            spmv["speedup_over_plain_diagonal_form_this_run"] = (1e-6 * pm / avg) if pm and avg > 0 else None
        out["roofline"]["spmv"] = spmv
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # (the CPU restatement has no spectral start: its pass is priced with the iteration count of the plain Galerkin start)
        plain = out["config"].get("without_spectral_start") or {}
        _note("cpu_baseline")
        out["cpu_baseline"] = cpu_baseline(prob, spec, be, plain.get("pcg_iterations_per_pass", pcg_its / K), args)
    # (a rehearsal with all ranks on ONE card: ranks + probe children must stay within the box's six processes per card)
    if sharded and world > 1 and not args.no_direct_probe and not args.direct_halo and not (args.share_one_gpu and 2 * world > 6):
        _note("probe of the direct halo / all-reduce (child processes)")
        probe = direct_probe(args, comm, space, dist, rank, world, local_rank)
        if rank == 0:
            out["config"]["direct_halo"]["probe"] = probe
    _note("done")
    if rank == 0:
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
        if args.watchdog_seconds > 0:
            import faulthandler
            faulthandler.cancel_dump_traceback_later()


def direct_probe(args, comm, space, dist, rank, world, local_rank):
    """N > 1, the solves on RCCL: what would the direct halo / direct all-reduce (pgd_comm_push_*, pgd_comm_allreduce_*) do on THIS
    hardware?  Their checked exchanges and a timing of 50 exchanges run in CHILD processes - one per rank, on the rank's GPU, a world of
    their own over gloo (tools/probe_direct.py) - after the timed region, with a deadline: nothing they do can touch the measurement or
    this process.  The same exchanges through the binding are timed here (they are what the solves have been doing all along).  Returns
    what rank 0's child printed + the binding's microseconds; {"error": ...} if a child failed."""
    import subprocess
    part = space.part
    n = space.num_vertices()
    res = {}
    try:
        res["microseconds_through_the_binding"] = comm._time_exchanges(n, part.own0, part.own1, part.lo_ghost, part.hi_ghost)
    except Exception as e:          # noqa: BLE001 - a probe never ends the run
        res["microseconds_through_the_binding"] = {"error": repr(e)[:200]}
    box = [_free_port() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    cmd = [sys.executable, os.path.join(ROOT, "tools", "probe_direct.py")] + [str(int(t)) for t in (
        rank, world, box[0], 0 if args.share_one_gpu else local_rank, n, part.own0, part.own1, part.lo_ghost, part.hi_ghost)]
    try:
        r = subprocess.run(cmd, capture_output=True, timeout=150, env=_own_world_env(MASTER_PORT=str(box[0])), cwd=ROOT)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if rank == 0:
            res.update(json.loads(line[-1]) if (r.returncode == 0 and line) else
                       {"error": "the probe's child ended with status %d: %s" % (r.returncode, r.stderr.decode()[-300:])})
    except Exception as e:          # noqa: BLE001
        res["error"] = repr(e)[:300]
    return res


def _own_world_env(**more):
    """The environment of a child process that forms a process group of its OWN, whatever launched us: none of the launcher's rank
    variables, and none of torch elastic's (TORCHELASTIC_USE_AGENT_STORE makes rank 0 a CLIENT of the agent's store at MASTER_PORT -
    at a port of our own nobody listens, and the child would wait for its whole rendezvous timeout)."""
    env = {k: v for k, v in os.environ.items()
           if not k.startswith("TORCHELASTIC_") and k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK",
                                                               "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "ROLE_NAME",
                                                               "PGD_TUNE", "PGD_BENCH_LAUNCHED", "PGD_BENCH_FORCE_LAUNCHER")}
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    env.update(more)
    return env


def ghost_rank_rehearsal():
    """The sharded iteration of a rank WITH ghost planes, on this one GPU (a child process: it needs a process group of its own):
    tools/bench_self_periodic.py makes ONE rank its own neighbour on both sides (PGD_TUNE_COMM_SELF_PERIODIC), so the solve has
    real ghost planes and RCCL send / receive inside its loop - everything of an N > 1 iteration but the wire and the other ranks -
    on the slab an 8-GPU rank of this workload owns (256 x 256 x 32).  None if the child could not run."""
    import subprocess
    try:
        env = _own_world_env()
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_self_periodic.py"), "--json"], capture_output=True,
                           timeout=120, env=env, cwd=ROOT)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return None
        res = json.loads(line[-1])
        res["note"] = ("one rank as its own neighbour on both sides (a problem periodic in z): real ghost planes, the halo exchange as RCCL "
                       "send / receive inside the iteration loop; microseconds per iteration over whole solves (setup included), best of two; "
                       "no wire, no other ranks - a lower bound of the time a rank of an 8-GPU run needs per iteration")
        return res
    except Exception:      # noqa: BLE001 - a side section must not take the line with it
        return None


def sharded_v_cycle_rank(args):
    """settings["preconditioner"] = "amg" through the SHARDED solver path (dist.pcg_mg: level 0 of the V-cycle on the rank's slab,
    levels >= 1 replicated behind an all-reduce, the PCG driven from the host over torch.distributed) with ONE rank, in a child
    process (a process group of its own): what a rank of an N-GPU run executes per iteration but the wire, the other ranks and the
    ghost planes.  None if the child could not run."""
    import subprocess
    try:
        env = _own_world_env()
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--dist-driver", "--python-driver", "--preconditioner", "amg",
                            "--steps", "6", "--warmup", "2", "--n", str(args.n), "--n-mu", str(args.n_mu), "--no-cpu-baseline", "--no-pmc",
                            "--no-csr-section", "--no-general-paths"], capture_output=True, timeout=120, env=env, cwd=ROOT)
        line = [l for l in r.stdout.decode().splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return None
        d = json.loads(line[-1])
        c = d["config"]
        return {"passes_per_s": d["value"], "ms_per_pass": d["ms_per_step"], "pcg_iterations_per_pass": c["pcg_iterations_per_step"],
                "us_per_pcg_iteration": c["us_per_pcg_iteration"], "solves_preconditioned_by_the_slab_v_cycle": c["sharded_v_cycle_solves"],
                "note": "one rank, no ghost planes; level 0 in the stencil march, the recurrence on the device slot bank driven from the host "
                        "(one synchronisation per iteration); compare config.multigrid_preconditioner (the unsharded in-library loop)"}
    except Exception:      # noqa: BLE001 - a side section must not take the line with it
        return None


def csr_section(be, prob, n_sp, nnz):
    """The north star's "Jacobi-PCG built from CSR SpMV": the CSR products of the library (plain k_spmv_csr and the
    column-dictionary form that is the default for the CSR path) timed in THIS run on the bench operator - the first
    spatial system of the problem - as the PCG instance (fused dot, y stored), >= 100 HIP-event samples each, priced with
    SURVEY 8d's formula 12 nnz + 20 n, which is what k_spmv_csr streams."""
    import numpy as np
    from pgdrome_amd import fem
    ctx = be.ctx
    op = _first_spatial_operator(prob)
    x = ctx.vec_from(np.random.default_rng(1234).uniform(-1, 1, n_sp))
    y = ctx.vec_alloc(n_sp)
    alg = 12.0 * nnz + 20.0 * n_sp
    res = {}
    try:
        ctx.tune(3, 0)                                # products from the CSR arrays, not the symmetric storage
        for name, dict_knob, own in (("k_spmv_csr<dot,store,64>", 0, alg), ("k_spmv_csr_dict16<dot,store>", 1, 8.0 * nnz + 22.0 * n_sp)):
            ctx.tune(2, dict_knob)
            ctx.flags_reset()
            for _ in range(3):
                ctx.spmv_dot_slot(op, x, y, x, 0, n_sp, 30)
            ctx.prof_enable(2)
            for _ in range(400):                      # one launch in four carries HIP events: 100 samples
                ctx.spmv_dot_slot(op, x, y, x, 0, n_sp, 30)
            p = ctx.prof_read()
            ctx.prof_enable(0)
            t = max(p["seconds"] / max(p["launches"], 1) - p.get("event_overhead", 0.0), 1e-9)
            res[name] = {"launches_timed": p["launches"], "launches": 400, "avg_launch_us": 1e6 * t,
                         "bytes_per_launch_survey_8d": alg, "achieved": alg / t / 1e9, "unit": "GB/s", "frac": alg / t / 8e12,
                         "kernel_min_bytes_per_launch": own, "kernel_min_bytes_frac": own / t / 8e12}
    finally:
        ctx.tune(3, 1)
        ctx.tune(2, 1)
        ctx.vec_free(x)
        ctx.vec_free(y)
        be.atom_free(op)
    res["note"] = ("frac of k_spmv_csr is physical (it streams values, column ids, row pointers, x, y = the 8d formula); the dictionary "
                   "form reads no column ids, so its physical fraction is kernel_min_bytes_frac")
    return res


def general_paths(be, spec, settings, passes=4, warm=1):
    """The engine WITHOUT what the headline leans on, in the same run on the same workload: (0) `row_class_dictionary` - the
    dictionary form of the product (PGD_TUNE_SPMV_STENCIL = 0: k_spmv_diac_march2, 17 B per row), which is what a uniform grid
    with natural boundaries or piecewise-constant coefficients gets; (1) `plain_march` - no row-class
    dictionary (PGD_TUNE_SPMV_ROW_CLASSES = 0): the product streams the 72 B per row of the scaled diagonal form, which is what
    any variable coefficient or graded mesh gets; (2) `csr` - no symmetric storage at all (PGD_TUNE_SPMV_SYM = 0): the north
    star's CSR SpMV (k_spmv_csr_dict16) inside textbook Jacobi-PCG, what a mesh without grid structure gets.  `passes` timed
    passes of solve_PGD each after `warm` untimed ones; products timed with HIP events like the headline's."""
    import time
    from pgdrome_amd import fem
    from pgdrome_amd.solver import PGDProblem
    res = {}
    for name, knob, reset in (("row_class_dictionary", (35, 0), (35, 1)), ("plain_march", (19, 0), (19, 1)), ("csr", (3, 0), (3, 1))):
        be.ctx.tune(*knob)
        try:
            prob = PGDProblem(**spec)
            st = {}

            def hook(n_pass, st=st):
                if n_pass == warm:
                    be.sync()
                    be.prof_enable(2)
                    st["k0"] = be.ctx.kernel_counts()
                    st["i0"], st["s0"], st["t0"] = fem.STATS["pcg_iterations"], fem.STATS["pcg_seconds"], time.perf_counter()
                elif n_pass == warm + passes:
                    be.sync()
                    st["t1"] = time.perf_counter()
                    st["i1"], st["s1"] = fem.STATS["pcg_iterations"], fem.STATS["pcg_seconds"]
                    st["k1"] = be.ctx.kernel_counts()
                    raise _Done()
            prob.pass_hook = hook
            try:
                for _ in range(100):
                    prob.solve_PGD(_problem="linear", settings=settings)
            except _Done:
                pass
            p = be.prof_read()
            be.prof_enable(False)
            its = max(st["i1"] - st["i0"], 1)
            kc = {k: st["k1"][k] - st["k0"][k] for k in st["k1"]}
            t = max(p["seconds"] / max(p["launches"], 1) - p.get("event_overhead", 0.0), 1e-9)
            own = p["own_bytes"] / max(p["launches"], 1)
            res[name] = {"passes_per_s": passes / (st["t1"] - st["t0"]), "passes": passes,
                         "us_per_pcg_iteration": 1e6 * (st["s1"] - st["s0"]) / its, "pcg_iterations_per_pass": its / passes,
                         "product_us": 1e6 * t, "product_kernel": max(kc, key=lambda k: kc[k]),
                         "product_bytes_per_launch": own, "product_frac_of_peak": own / t / 8e12 if t > 0 else None,
                         "knob": "pgd_tune(%d, %d)" % knob}
        finally:
            be.ctx.tune(*reset)
            be.prof_enable(False)
    res["note"] = ("same workload, same run, after the timed region, WITHOUT the spectral start space (its harvest needs the lattice structure "
                   "these paths stand in for the absence of): the headline's 16 B/row product needs a uniform lattice with constant "
                   "coefficients and a Dirichlet hull; row_class_dictionary is the rate with natural boundaries / piecewise-constant coefficients "
                   "(17 B/row), plain_march the rate of any variable coefficient or graded mesh (72 B/row), csr the rate on the CSR kernels "
                   "with textbook PCG (no grid structure at all)")
    return res


def plain_start_path(be, spec, settings, warm, passes):
    """The headline's workload and window WITHOUT the spectral start space (settings["spectral_start"] = 0): the Galerkin start
    over the previous iterate and the stored modes alone, as in rounds 1-3 - same warm-up, same number of timed passes."""
    import time
    from pgdrome_amd import fem
    from pgdrome_amd.solver import PGDProblem
    st = {}
    prob = PGDProblem(**spec)

    def hook(n_pass):
        if n_pass == warm:
            be.sync()
            st["i0"], st["s0"], st["t0"] = fem.STATS["pcg_iterations"], fem.STATS["pcg_seconds"], time.perf_counter()
        elif n_pass == warm + passes:
            be.sync()
            st["t1"] = time.perf_counter()
            st["i1"], st["s1"] = fem.STATS["pcg_iterations"], fem.STATS["pcg_seconds"]
            raise _Done()
    prob.pass_hook = hook
    if warm == 0:
        hook(0)
    try:
        for _ in range(1000):
            prob.solve_PGD(_problem="linear", settings=settings)
    except _Done:
        pass
    its = max(st["i1"] - st["i0"], 1)
    return {"passes_per_s": passes / (st["t1"] - st["t0"]), "passes": passes, "warmup": warm, "ms_per_pass": 1e3 * (st["t1"] - st["t0"]) / passes,
            "pcg_iterations_per_pass": its / passes, "us_per_pcg_iteration": 1e6 * (st["s1"] - st["s0"]) / its,
            "settings": {"spectral_start": 0}}


def multigrid_path(be, spec, settings, modes_ref, passes=8, warm=2):
    """The same workload with settings["preconditioner"] = "amg" (the reference forwards the key to its linear solver,
    solver.py:593-594): pgd_pcg_solve preconditioned by the matrix-free V-cycle of pgdrome_amd/csrc/pgd_mg.hip instead of the
    diagonal scaling.  NOT the headline - the north star names Jacobi-PCG - but what the engine does for this problem when asked:
    same systems, same stop test, same tolerance; the modes of the timed passes are compared with the headline run's."""
    import time
    import numpy as np
    from pgdrome_amd import fem
    from pgdrome_amd.solver import PGDProblem
    st = {}
    prob = PGDProblem(**spec)

    def hook(n_pass):
        if n_pass == warm:
            be.sync()
            st["i0"], st["s0"], st["m0"], st["t0"] = fem.STATS["pcg_iterations"], fem.STATS["pcg_seconds"], fem.STATS["mg_solves"], time.perf_counter()
        elif n_pass == warm + passes:
            be.sync()
            st["t1"] = time.perf_counter()
            st["i1"], st["s1"], st["m1"] = fem.STATS["pcg_iterations"], fem.STATS["pcg_seconds"], fem.STATS["mg_solves"]
            raise _Done()
    prob.pass_hook = hook
    try:
        for _ in range(100):
            prob.solve_PGD(_problem="linear", settings=dict(settings, preconditioner="amg"))
    except _Done:
        pass
    its = max(st["i1"] - st["i0"], 1)
    res = {"passes_per_s": passes / (st["t1"] - st["t0"]), "passes": passes, "ms_per_pass": 1e3 * (st["t1"] - st["t0"]) / passes,
           "pcg_iterations_per_pass": its / passes, "us_per_pcg_iteration": 1e6 * (st["s1"] - st["s0"]) / its,
           "solves_preconditioned_by_the_v_cycle": st["m1"] - st["m0"], "seconds_in_pcg_solves": st["s1"] - st["s0"],
           "settings": {"preconditioner": "amg"}, "knob": "pgd_tune(40, 1)",
           "note": "geometric multigrid V(1,1) on the stencil form (Galerkin stencils of the nested P1 lattices, damped Jacobi), "
                   "same stop test and tolerance as the headline's Jacobi-PCG; single GPU only"}
    if modes_ref:
        # completed modes of this run against the headline run's (rank-1 products: the sign of a mode's factors is not fixed)
        n_cmp = min(len(modes_ref), len(prob.PGD_func[0]) if getattr(prob, "PGD_func", None) else 0)
        worst = 0.0
        for k in range(n_cmp):
            a = [np.asarray(f[k].vector()[:]) for f in prob.PGD_func]
            b = modes_ref[k]
            sgn = 1.0 if float(np.dot(a[0], b[0])) >= 0 else -1.0
            num = np.linalg.norm(a[0] * sgn - b[0]) / max(np.linalg.norm(b[0]), 1e-300)
            worst = max(worst, float(num))
        res["modes_compared_with_the_headline_run"] = n_cmp
        res["worst_relative_l2_difference_of_a_spatial_mode"] = worst
    return res


def _first_spatial_operator(prob):
    from pgdrome_amd import fem
    V = prob.V[0]
    bcs = prob.bc
    Fs = prob.get_Fsinit(prob.V, bcs, None)
    u, v = fem.TrialFunction(V), fem.TestFunction(V)
    a = prob.lhs_fct(u, v, Fs, prob.meshes, prob.dom, prob.param, prob.prob[0], 0)
    A = fem.assemble(a)
    if bcs[0] != 0:
        for bc in fem._bc_list(bcs[0]):
            A.apply_dirichlet(bc)
    return A.op()


def _first_spatial_system(prob):
    """(A, b) of the first spatial solve of the run: the operator and right-hand side the callbacks give for the initial factors."""
    from pgdrome_amd import fem
    V = prob.V[0]
    bcs = prob.bc
    Fs = prob.get_Fsinit(prob.V, bcs, None)
    u, v = fem.TrialFunction(V), fem.TestFunction(V)
    a = prob.lhs_fct(u, v, Fs, prob.meshes, prob.dom, prob.param, prob.prob[0], 0)
    l = prob.rhs_fct(u, v, Fs, prob.meshes, prob.dom, prob.param, prob.load, [[] for _ in Fs], prob.prob[0], 0, 0)
    A, b = fem.assemble(a), fem.assemble(l)
    fem._apply_bcs_system(A, b, bcs[0] if bcs[0] != 0 else None)
    return A, b


def pmc_traffic(own_bytes, upd_bytes):
    """HBM-side bytes per launch of the PCG product and of the vector update.  bench.py cannot read PMC counters of its own
    process, so it starts `rocprofv3 --pmc` CHILD processes (one counter per pass, as MI355X_MICROARCH.md prescribes) on
    tools/pmc_spmv_sym.py - the same kernel instances (12 iterations of pgd_pcg_solve on the same 256^3 operator shape) - and
    applies the calibrated factors of profiles/r02a_pmc_calibration_and_march.json (FETCH_SIZE x 2 for 8 and 16 B/lane
    streaming loads, measured on a known 1 GiB stream; WRITE_SIZE exact).  Falls back to the committed profile when no
    profiler can be started.  Returns {"product": {...}, "update": {...}} with `traffic`, `traffic_source`,
    `traffic_over_kernel_min`."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    fallback = os.path.join(ROOT, "profiles", "pmc_spmv_latest.json")
    kernels = {"product": (lambda name: ("k_spmv_dia" in name or "k_spmv_stencil" in name) and "<true, true" in name, own_bytes),
               "update": (lambda name: "k_pcg1_update" in name, upd_bytes)}
    res = {k: {"traffic": None, "traffic_source": None} for k in kernels}
    rp = shutil.which("rocprofv3")
    nested = any(k.startswith(("ROCPROF", "ROCP_", "ROCPROFILER")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if rp and not nested:
        env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCPROF", "ROCP_", "HSA_TOOLS")) and k != "LD_PRELOAD"}
        env["TMPDIR"] = "/tmp"
        vals = {k: {} for k in kernels}
        try:
            for counter in ("FETCH_SIZE", "WRITE_SIZE"):
                d = tempfile.mkdtemp(prefix="pgd_pmc_", dir="/tmp")
                subprocess.run([rp, "--pmc", counter, "--output-format", "csv", "-d", d, "--", sys.executable,
                                os.path.join(ROOT, "tools", "pmc_spmv_sym.py"), "256", "grid", "0"],
                               check=True, timeout=75, env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                got = {k: [] for k in kernels}
                for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                    with open(f) as fh:
                        for row in csv.DictReader(fh):
                            if row.get("Counter_Name") != counter:
                                continue
                            for k, (match, _) in kernels.items():
                                if match(row.get("Kernel_Name", "")):
                                    got[k].append(float(row["Counter_Value"]))
                shutil.rmtree(d, ignore_errors=True)
                for k, (_, least) in kernels.items():
                    if not got[k]:
                        continue
                    v = sum(got[k]) / len(got[k])
                    if least > 0 and v * 1024.0 < 64.0 * least:       # the counter is reported in KiB on ROCm 7.x
                        v *= 1024.0
                    vals[k][counter] = v
            for k, (_, least) in kernels.items():
                if len(vals[k]) == 2 and least > 0:
                    res[k]["traffic"] = 2.0 * vals[k]["FETCH_SIZE"] + vals[k]["WRITE_SIZE"]
                    res[k]["traffic_source"] = ("rocprofv3 --pmc child processes of this run (tools/pmc_spmv_sym.py 256 grid): FETCH_SIZE x 2 "
                                                "(calibrated, profiles/r02a_pmc_calibration_and_march.json) + WRITE_SIZE; counts Infinity-Cache hits")
                    res[k]["traffic_over_kernel_min"] = res[k]["traffic"] / least
            if res["product"]["traffic"] is not None:
                return res
        except Exception as e:      # noqa: BLE001 - the bench line must still be printed
            res["product"]["traffic_error"] = repr(e)[:200]
    if os.path.exists(fallback):
        with open(fallback) as f:
            saved = json.load(f)
        for k, (_, least) in kernels.items():
            key = "hbm_bytes_per_launch" if k == "product" else "update_hbm_bytes_per_launch"
            if saved.get(key) and least > 0:
                res[k]["traffic"] = saved[key]
                res[k]["traffic_source"] = "profiles/pmc_spmv_latest.json (separate rocprofv3 --pmc passes, not this run)"
                res[k]["traffic_over_kernel_min"] = res[k]["traffic"] / least
    return res


def cpu_baseline(prob, spec, be, pcg_its_per_step, args):
    """The oracle (CPU restatement of the reference algorithm, not FEniCS) timed on the host
    cores on a BOUNDED sample of the same workload: the first spatial system of the run,
    `sample` Jacobi-PCG iterations; one step costs (PCG iterations per step) x that."""
    from oracle import cpu_baseline as cb
    return cb.run(prob, spec, be, pcg_its_per_step, args.cpu_seconds)


if __name__ == "__main__":
    main()
