import os, sys, collections, traceback
sys.path.insert(0, '/root/repo')
import torch.multiprocessing as mp

def worker(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pgdrome_amd.hip_backend import HipBackend; import torch
    from pgdrome_amd import dist as pdist, fem, problems, spectral
    from pgdrome_amd.solver import PGDProblem
    
    torch.cuda.set_device(0); ts = torch.cuda.Stream(device=0); torch.cuda.set_stream(ts); be = fem.set_backend(HipBackend(0, ts.cuda_stream))
    comm = pdist.TorchComm(dist, be, True)
    P = fem.Point
    mesh = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), 127, 127, 127)
    counts = collections.Counter()
    for name in ("allreduce_array", "allreduce_sum", "allreduce_maxloc"):
        orig = getattr(comm, name)
        def wrap(*a, _o=orig, _n=name, **k):
            st = traceback.extract_stack(limit=6)
            key = _n + " <- " + " <- ".join("%s:%d" % (f.name, f.lineno) for f in reversed(st[:-1]) if "fem.py" in f.filename or "solver.py" in f.filename or "spectral.py" in f.filename)[:150]
            counts[key] += 1
            return _o(*a, **k)
        setattr(comm, name, wrap)
    p = PGDProblem(**problems.reaction_diffusion(mesh, 17, PGD_nmax=50, PGD_tol=1e-12))
    class Done(Exception): pass
    W, K = 5, 20
    def hook(n):
        if n == W: counts.clear()
        if n == W + K: raise Done()
    p.pass_hook = hook
    try:
        for _ in range(100):
            p.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "relative_tolerance": 1e-10, "spectral_start": 48})
    except Done:
        pass
    if rank == 0:
        print("modes so far", len(p.num_fp_it), "total per pass", sum(counts.values()) / K)
        for k, v in counts.most_common(16):
            print("%6.2f per pass  %s" % (v / K, k))
    dist.barrier(); dist.destroy_process_group()

if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    ps = [ctx.Process(target=worker, args=(r, 2, 29791)) for r in range(2)]
    [p.start() for p in ps]; [p.join() for p in ps]
