"""The sharded iteration of ONE rank with ghost planes on both sides, on a single GPU: the rank is its own neighbour
(PGD_TUNE_COMM_SELF_PERIODIC), so the halo exchange is real RCCL send / receive, the product is an interior launch + the two
boundary planes, and the phases can be timed - everything of an N > 1 iteration but the wire and the other ranks.

    PGD_DEBUG_PCG=1 python tools/bench_self_periodic.py [--planes 32] [--nxy 256]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from pgdrome_amd import fem
from pgdrome_amd.hip_backend import HipBackend

ap = argparse.ArgumentParser()
ap.add_argument("--planes", type=int, default=32, help="owned planes (256 / 8 = the slab of an 8-GPU rank at 256^3)")
ap.add_argument("--nxy", type=int, default=256)
ap.add_argument("--only", default="", help="with --json: just this variant (profiling)")
ap.add_argument("--json", action="store_true", help="one JSON line on stdout: microseconds per iteration of whole solves (bench.py's side section)")
args = ap.parse_args()
if args.json:                      # RCCL prints its banner to stdout: the result line goes to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
import json
import time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
tstream = torch.cuda.Stream(device=0)
torch.cuda.set_stream(tstream)
be = fem.set_backend(HipBackend(0, tstream.cuda_stream))
ctx = be.ctx
nx = ny = args.nxy
VARIANTS = (("stream_ordered_one_march", ((45, 1 << 40), (46, 1))), ("stream_ordered_interior_plus_boundary", ((45, 1 << 40), (46, 0))),
            ("overlapped_on_the_halo_stream", ((45, 0), (46, 1))),
            # the boundary planes stored straight into the ghost planes + a posted sequence number (pgd_comm_push_*): no RCCL kernel
            ("direct_halo_one_march", ((45, 1 << 40), (46, 1), (49, 1), ("push", 1))),
            # (49 = 0: the planes leave from a launch of their own, k_halo_push, instead of the update kernel)
            ("direct_halo_push_kernel", ((45, 1 << 40), (46, 1), (49, 0), ("push", 1))),
            # ... and the loop's sums through the mailbox kernel (pgd_comm_allreduce_attach): neither k_pcg1_sums nor an all-reduce of the binding
            ("direct_halo_and_allreduce", ((45, 1 << 40), (46, 1), (49, 1), ("push", 1), ("ar", 1))))


def slab(planes):
    nzl = planes + 2
    mesh = fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, (nzl - 1) / (nx - 1.0)), nx - 1, ny - 1, nzl - 1)
    coords = mesh.coordinates()
    h = ctx.mesh_upload(coords, mesh.cells())
    n, plane = coords.shape[0], nx * ny
    ak, am = ctx.atom_assemble(h, 1), ctx.atom_assemble(h, 0)
    hull = np.where((coords[:, 0] <= 1e-12) | (coords[:, 0] >= 1 - 1e-12) | (coords[:, 1] <= 1e-12) | (coords[:, 1] >= 1 - 1e-12))[0].astype(np.int32)
    b = np.random.default_rng(9).uniform(-1, 1, n)
    b[hull] = 0.0
    return h, n, plane, plane, n - plane, ak, am, hull, ctx.vec_from(b)


def timed(S, variants):
    h, n, plane, own0, own1, ak, am, hull, bv = S
    res, xs = {}, {}
    for variant, tune in variants:
        ctx.comm_unbind()
        ctx.comm_bind_rccl(ctx.comm_unique_id(), 0, 1)
        ok = ctx.comm_overlap(1)
        ctx.tune(44, 1)
        pushing = False
        for knob, value in tune:
            if knob == "push":
                blob = ctx.comm_push_export(n, own0, own1, plane, plane)
                pushing = ctx.comm_push_attach(blob, blob)
            elif knob == "ar":
                ar_ok = ctx.comm_allreduce_attach([blob])
            else:
                ctx.tune(knob, value)
        best = None
        for rep in range(3):
            o2 = ctx.op_combine(h, [ak, am], [1.0, 3.0], hull)      # a fresh operator per solve, as the fixed-point loop has
            xv = ctx.vec_alloc(n)
            be.sync()
            t0 = time.perf_counter()
            it, rel = ctx.pcg_solve_sharded(o2, bv, xv, own0, own1, plane, plane, 1e-10, 0.0, 10000)
            be.sync()
            dt = time.perf_counter() - t0
            if rep == 2:
                xs[variant] = ctx.vec_download(xv)
            ctx.vec_free(xv)
            ctx.atom_free(o2)
            if rep and (best is None or dt < best):
                best = dt
        res[variant] = {"us_per_iteration": 1e6 * best / max(it, 1), "iterations": it, "second_stream_used": bool(ctx.comm_overlap(-2)),
                        "second_stream_available": bool(ok)}
        if variant.endswith("_and_allreduce"):
            res[variant].update(direct_allreduce_attached=bool(ar_ok), direct_allreduce_used=bool(ctx.comm_allreduce_direct(-2)))
        if variant.startswith("direct_halo"):
            res[variant].update(direct_halo_attached=bool(pushing), direct_halo_used=bool(ctx.comm_push(-2)),
                                same_bits_as_the_rccl_exchange=bool("stream_ordered_one_march" in xs and
                                                                    np.array_equal(xs[variant], xs["stream_ordered_one_march"])))
    return res


def release(S):
    h, n, plane, own0, own1, ak, am, hull, bv = S
    ctx.comm_unbind()
    ctx.vec_free(bv)
    for a in (ak, am):
        ctx.atom_free(a)
    ctx.mesh_free(h)


if args.json:
    S = slab(args.planes)
    res = {"owned_planes": args.planes, "plane": [nx, ny], "rows": int(S[4] - S[3])}
    res.update(timed(S, [v for v in VARIANTS if not args.only or v[0] == args.only]))
    release(S)
    if args.planes == 32 and nx == 256 and not args.only:
        # the slabs of a 4- and a 2-GPU rank of the same grid: what the iteration costs there (one march / second stream)
        res["larger_slabs"] = {}
        for planes in (64, 128):
            S = slab(planes)
            res["larger_slabs"][str(planes)] = {k: v["us_per_iteration"] for k, v in timed(S, (VARIANTS[0], VARIANTS[2], VARIANTS[3])).items()}
            release(S)
    os.write(result_fd, (json.dumps(res) + "\n").encode())
    dist.destroy_process_group()
    sys.exit(0)
S = slab(args.planes)
h, n, plane, own0, own1, ak, am, hull, bv = S
op = ctx.op_combine(h, [ak, am], [1.0, 3.0], hull)
for variant in ("rccl", "rccl+overlap"):
    ctx.comm_unbind()
    ctx.comm_bind_rccl(ctx.comm_unique_id(), 0, 1)
    ctx.comm_overlap(1 if variant == "rccl+overlap" else 0)
    ctx.tune(44, 1)
    ctx.tune(45, 0)          # the second stream whatever the slab's size
    for rep in range(2):
        xv = ctx.vec_alloc(n)
        if rep:
            ctx.comm_prof(1)
        it, rel = ctx.pcg_solve_sharded(op, bv, xv, own0, own1, plane, plane, 1e-10, 0.0, 10000)
        ctx.vec_free(xv)
    ph = ctx.comm_prof(0)
    ns = max(ph.pop("samples"), 1.0)
    print(variant, "iterations", it, "phases (us, one iteration per chunk between HIP events, each incl. the event pair's own ~4.8 us):",
          {k: round(1e6 * v / ns, 1) for k, v in ph.items() if k != "host_boundary_wait"}, flush=True)
dist.destroy_process_group()
