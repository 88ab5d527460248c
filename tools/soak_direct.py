"""Soak of the direct halo / direct all-reduce between PROCESSES on one GPU: BASELINE config 5 (256^3 x time x two parameters) with the
space sharded over `world` processes, the first `modes` enrichment steps - some hundred thousand exchanges - and the pass counts of the
unsharded run to hold it against ([4, 6, 5, 6, 12, 8, 50, 10, ...]).

    python tools/soak_direct.py [world=2] [modes=8]
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp


def worker(rank, world, port, modes, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PGD_HALO_DIRECT"] = "1"
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgdrome_amd import dist as pdist, fem, problems
        from pgdrome_amd.hip_backend import HipBackend
        from pgdrome_amd.solver import PGDProblem
        torch.cuda.set_device(0)
        ts = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(ts)
        be = fem.set_backend(HipBackend(0, ts.cuda_stream))
        comm = pdist.TorchComm(dist, be, in_library=True)
        P = fem.Point
        mesh = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), 255, 255, 255)
        p = PGDProblem(**problems.transient_heat(mesh, 256, 64, PGD_nmax=modes))
        t0 = time.time()
        p.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "relative_tolerance": 1e-10})
        be.sync()
        if rank == 0:
            q.put(dict(world=world, num_fp_it=p.num_fp_it, amplitude=[float(a) for a in p.amplitude], seconds=time.time() - t0,
                       pcg_iterations=fem.STATS["pcg_iterations"], direct_halo=bool(comm.direct_halo), direct_allreduce=bool(comm.direct_allreduce),
                       used=[bool(be.comm_push(-2)), bool(be.comm_allreduce_direct(-2))]))
    finally:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    modes = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, world, port, modes, q)) for r in range(world)]
    [p.start() for p in ps]
    out = q.get(timeout=1000)
    [p.join(timeout=120) for p in ps]
    print(json.dumps(out))
