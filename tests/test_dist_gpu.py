"""The sharded PCG on the GPU.

* world_size 1 over RCCL: the in-library loop (pgd_pcg_solve_sharded) bound to a real RCCL communicator
  (unique id, ncclCommInitRank, the library's ring-shift self-test = ncclSend/ncclRecv to itself, in-stream
  ncclAllReduce on the slot bank), and the Python-driven loop over torch.distributed (zero-copy torch views
  of library-owned memory, shared HIP stream).
* 2 and 3 processes sharing GPU 0 with REAL halo exchanges (two ranks cannot share one GPU under RCCL, so
  the transport is gloo staged through the host): the same in-library C++ loop with the two communication
  steps bound to callbacks, and the Python-driven loop.
The N > 1 exchange logic is also covered on the CPU by tests/test_dist_cpu.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _collect(procs, q, n_results, timeout):
    """Start the worker processes, take `n_results` items from the queue, join; whatever happens, no worker outlives the test
    (a worker blocked in a collective would keep the test runner from exiting)."""
    for pr in procs:
        pr.start()
    try:
        out = [q.get(timeout=timeout) for _ in range(n_results)]
        for pr in procs:
            pr.join(timeout=120)
            assert pr.exitcode == 0, pr.exitcode
        return out
    finally:
        for pr in procs:
            if pr.is_alive():
                pr.kill()
                pr.join(timeout=30)


def test_sharded_driver_world1_matches_library_pcg():
    import torch
    import torch.distributed as dist
    from pgdrome_amd import dist as pdist, fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    old = fem._backend
    try:
        P = fem.Point
        shape = (24, 20, 28)
        # library-driven PCG
        be1 = fem.set_backend(HipBackend(0))
        fem.clear_caches()
        ref = PGDProblem(**problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 17, PGD_nmax=3))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
        its_ref = fem.STATS["pcg_iterations"]
        # host-driven PCG on torch's stream, scalars all-reduced through RCCL
        with pytest.raises(RuntimeError):          # default stream (handle 0) = library's own stream: refused
            pdist.TorchComm(dist, HipBackend(0, torch.cuda.current_stream().cuda_stream or None))
        tstream = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(tstream)
        be2 = fem.set_backend(HipBackend(0, tstream.cuda_stream))
        fem.clear_caches()
        os.environ["PGD_HALO_OVERLAP"] = "1"        # (the second communicator is set up on request only)
        for in_library in (True, False):
            fem.clear_caches()
            comm = pdist.TorchComm(dist, be2, in_library=in_library)
            if in_library:     # the library opened its own RCCL communicator and passed its ring-shift check
                assert comm.in_library == "rccl" and be2.comm_info() == {"kind": "rccl", "rank": 0, "world": 1}
                # the halo communicator (ncclCommSplit) on its own stream passed its event-ordered ring shift as well
                assert comm.halo_overlap is True and be2.comm_overlap() is True
            else:
                assert comm.in_library is None
            mesh = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), *shape)
            assert mesh.part.lo_ghost == 0 and mesh.part.hi_ghost == 0
            p = PGDProblem(**problems.reaction_diffusion(mesh, 17, PGD_nmax=3))
            fem.STATS["pcg_iterations"] = 0
            p.solve_PGD(_problem="linear")
            assert p.num_fp_it == ref.num_fp_it
            np.testing.assert_allclose(p.amplitude, ref.amplitude, rtol=1e-9)
            for m in range(ref.PGD_modes):
                got = p.PGD_func[0][m].compute_vertex_values()
                assert np.linalg.norm(got - ref_x[m]) <= 1e-8 * np.linalg.norm(ref_x[m])
            assert fem.STATS["pcg_iterations"] > 100
            if not in_library:
                assert comm.stats["allreduce"] > 100
            else:
                # PGD_HALO_OVERLAP=1 is an opt-in that takes effect (it used to set up the second communicator and
                # leave the threshold at 2^40 rows): the last sharded solve took the second stream
                assert be2.comm_overlap(-2) is True
        be2.comm_unbind()
        assert be2.comm_info()["kind"] == "none"
        # zero-copy view really aliases the library's memory
        v = be2.vec_from(np.arange(5.0))
        t = be2.vec_tensor(v)
        t += 1.0
        torch.cuda.synchronize()
        assert np.array_equal(be2.vec_to_host(v), np.arange(5.0) + 1.0)
    finally:
        os.environ.pop("PGD_HALO_OVERLAP", None)
        torch.cuda.set_stream(torch.cuda.default_stream(0))
        fem.set_backend(old)
        fem.clear_caches()
        dist.destroy_process_group()


def test_halo_exchange_with_itself_overlapped_or_not_is_the_same_solve():
    """ONE rank that is its own neighbour on both sides (PGD_TUNE_COMM_SELF_PERIODIC: a problem periodic in z): the sharded
    solve with real ghost planes, boundary-plane launches and RCCL send / receive INSIDE the iteration loop on a single GPU -
    (a) through the callback binding (the exchange as two device copies on the compute stream), (b) over RCCL on the compute
    stream, (c) over RCCL on the halo communicator and its stream, overlapped with the interior rows' product: the same
    launches in the same order, so iteration counts, residuals and x must be IDENTICAL; the solution checked through the CSR
    kernels with the ghost planes filled on the host."""
    import torch
    import torch.distributed as dist
    from pgdrome_amd import dist as pdist, fem
    from pgdrome_amd.hip_backend import HipBackend

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    old = fem._backend
    try:
        tstream = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(tstream)
        be = fem.set_backend(HipBackend(0, tstream.cuda_stream))
        fem.clear_caches()
        ctx = be.ctx
        nx, ny, nzl = 256, 256, 34                       # 32 owned planes between two ghost planes: the slab of an 8-GPU rank at 256^3
        mesh = fem.BoxMesh(fem.Point(0, 0, 0), fem.Point(1, 1, (nzl - 1) / 255.0), nx - 1, ny - 1, nzl - 1)
        coords = mesh.coordinates()
        h = ctx.mesh_upload(coords, mesh.cells())
        n, plane = coords.shape[0], nx * ny
        own0, own1 = plane, n - plane
        ak, am = ctx.atom_assemble(h, 1), ctx.atom_assemble(h, 0)
        hull = np.where((coords[:, 0] <= 1e-12) | (coords[:, 0] >= 1 - 1e-12) | (coords[:, 1] <= 1e-12) | (coords[:, 1] >= 1 - 1e-12))[0].astype(np.int32)
        op = ctx.op_combine(h, [ak, am], [1.0, 3.0], hull)
        rng = np.random.default_rng(9)
        b = rng.uniform(-1, 1, n)
        b[hull] = 0.0
        bv = ctx.vec_from(b)

        def cb_halo(vec, o0, o1, lo_g, hi_g):
            t = be.vec_tensor(vec)
            t[0:lo_g].copy_(t[o1 - lo_g:o1])
            t[o1:o1 + hi_g].copy_(t[o0:o0 + hi_g])

        def cb_allreduce(first, count):
            pass

        uid = ctx.comm_unique_id()
        out = {}
        for variant in ("callbacks", "rccl", "rccl+overlap"):
            ctx.comm_unbind()
            if variant == "callbacks":
                ctx.comm_bind_callbacks(cb_halo, cb_allreduce, 0, 1)
            else:
                ctx.comm_bind_rccl(uid if variant == "rccl" else ctx.comm_unique_id(), 0, 1)
                assert ctx.comm_overlap(1 if variant == "rccl+overlap" else 0) == (variant == "rccl+overlap")
            ctx.tune(44, 1)                              # (a binding starts from the defaults)
            ctx.tune(45, 0)                              # the second stream whatever the slab's size
            xv = ctx.vec_alloc(n)
            k0 = ctx.kernel_counts()
            it, rel = ctx.pcg_solve_sharded(op, bv, xv, own0, own1, plane, plane, 1e-10, 0.0, 10000)
            k1 = ctx.kernel_counts()
            # overlapped: the interior rows march while the planes travel, the two boundary planes follow in row order; in stream
            # order ALL owned planes march in one launch, the ghost planes staged as data (k_stencil_ghost)
            assert k1["stencil_march"] > k0["stencil_march"]
            assert (k1["dia_rows"] > k0["dia_rows"]) == (variant == "rccl+overlap"), (variant, k0, k1)
            assert ctx.comm_overlap(-2) == (variant == "rccl+overlap")
            x = ctx.vec_download(xv)
            out[variant] = (it, rel, x)
            assert it > 50 and rel <= 1e-10
            # the library returns x with current ghost planes: its own far boundary planes
            assert np.array_equal(x[:plane], x[own1 - plane:own1]) and np.array_equal(x[own1:], x[own0:own0 + plane])
            # the residual on the owned rows through the CSR kernels
            yv = ctx.vec_alloc(n)
            ctx.tune(3, 0)
            ctx.spmv(op, xv, yv, own0, own1)
            ctx.tune(3, 1)
            r = (b - ctx.vec_download(yv))[own0:own1]
            assert np.linalg.norm(r) <= 1.05e-10 * np.linalg.norm(b[own0:own1])
            for v in (xv, yv):
                ctx.vec_free(v)
        a, c, o = out["callbacks"], out["rccl"], out["rccl+overlap"]
        assert a[0] == c[0] and a[1] == c[1] and np.array_equal(a[2], c[2]), (a[0], c[0], a[1], c[1])      # the same launches
        # (three launches instead of one: y is bit-identical, the fused dots are grouped differently - iterates equal to rounding)
        assert abs(o[0] - a[0]) <= 1 and np.linalg.norm(o[2] - a[2]) <= 1e-9 * np.linalg.norm(a[2])
        ctx.comm_unbind()
    finally:
        torch.cuda.set_stream(torch.cuda.default_stream(0))
        fem.set_backend(old)
        fem.clear_caches()
        dist.destroy_process_group()


def _shared_gpu_worker(rank, world, port, shape, q, in_library, problem="reaction"):
    """One of several ranks that all use GPU 0: HIP kernels for the local arithmetic, gloo (staged
    through the host) for the exchange steps."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgdrome_amd import dist as pdist, fem, problems
        from pgdrome_amd.hip_backend import HipBackend
        from pgdrome_amd.solver import PGDProblem
        torch.cuda.set_device(0)
        tstream = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(tstream)
        be = fem.set_backend(HipBackend(0, tstream.cuda_stream))
        comm = pdist.TorchComm(dist, be, in_library=in_library)
        assert comm.in_library == ("callbacks" if in_library else None)
        P = fem.Point
        mesh = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), *shape)
        if problem == "convection":
            p = PGDProblem(**problems.convection_diffusion(mesh, 7, 6, PGD_nmax=3))
        elif problem == "elastic":
            p = PGDProblem(**problems.elastic_block(mesh, 7, PGD_nmax=3))
        elif problem == "heat4":
            p = PGDProblem(**problems.transient_heat(mesh, 9, 5, PGD_nmax=4))
        elif problem == "p2":
            p = PGDProblem(**problems.reaction_diffusion(mesh, 9, PGD_nmax=3, degree=2))
        else:
            p = PGDProblem(**problems.reaction_diffusion(mesh, 17, PGD_nmax=3))
        p.solve_PGD(_problem="linear")
        if problem == "elastic":
            view = fem._block_layout(mesh, 1, 3).shard_view()
            modes_x = [pdist.gather_owned(comm, view, f.vector()[:]) for f in p.PGD_func[0]]
        else:
            modes_x = [pdist.gather_owned(comm, mesh, f.compute_vertex_values()) for f in p.PGD_func[0]]
        if rank == 0:
            q.put(dict(num_fp_it=p.num_fp_it, pcg_iterations=fem.STATS["pcg_iterations"], bicgstab_iterations=fem.STATS.get("bicgstab_iterations", 0), amplitude=p.amplitude, modes_x=modes_x, stats=dict(comm.stats),
                       kernels=be.ctx.kernel_counts(), direct_halo=bool(comm.direct_halo),
                       direct_halo_used=bool(be.comm_push(-2)) if comm.in_library else False,
                       direct_allreduce=bool(comm.direct_allreduce),
                       direct_allreduce_used=bool(be.comm_allreduce_direct(-2)) if comm.in_library else False))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,in_library", [(2, True), (3, True), (2, False)])
def test_sharded_solve_with_real_halos_on_one_gpu(world, in_library):
    """Row-sharded solve with the HIP kernels and REAL halo exchanges: `world` processes share GPU 0
    and exchange through gloo - with the iteration loop inside the library (communication by callbacks)
    or driven from Python.  Must reproduce the unsharded GPU run (same modes, same iteration counts)."""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    shape = (24, 20, 29)
    old = fem._backend
    fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 17, PGD_nmax=3))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, shape, q, in_library)) for r in range(world)]
    out = _collect(procs, q, 1, 300)[0]
    assert out["num_fp_it"] == ref.num_fp_it
    np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-8)
    for m in range(ref.PGD_modes):
        assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-7 * np.linalg.norm(ref_x[m])
    assert out["stats"]["halo"] > 100


def test_sharded_solve_marches_on_row_classes():
    """Two ranks whose slabs are large enough for the z-march (128 x 128 planes, 50 owned planes each): the interior rows'
    product of the in-library sharded loop runs on the row-class dictionary of each rank's LOCAL scaled operator
    (k_spmv_diac_march2), the boundary planes in row order, the x update lags; the run must reproduce the unsharded one."""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    shape = (127, 127, 99)
    old = fem._backend
    fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 17, PGD_nmax=3))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    outs = {}
    saved = os.environ.get("PGD_TUNE")
    try:
        # the scalar step of an iteration inside the update kernel (default) and as a launch of its own; the true residual norm
        # measured only near the end (default) and in every iteration: the same iterates, the same bits
        # (+ 46=0: interior march and boundary planes in row order instead of one march over all owned planes)
        for fold, tune in ((1, "29=1"), (0, "29=0"), (2, "29=1,30=0"), (3, "29=0,30=0"), (4, "29=1,46=0")):
            os.environ["PGD_TUNE"] = tune + os.environ.get("PGD_TEST_EXTRA_TUNE", "")      # (e.g. ",35=0": the dictionary form of the interior product)
            q = ctx.Queue()
            port = _free_port()
            procs = [ctx.Process(target=_shared_gpu_worker, args=(r, 2, port, shape, q, True)) for r in range(2)]
            out = outs[fold] = _collect(procs, q, 1, 600)[0]
            assert out["kernels"]["diac_march"] + out["kernels"]["stencil_march"] > 100
            one_march = fold != 4 and out["kernels"]["stencil_march"] > 100
            assert (out["kernels"]["dia_rows"] < 20) if one_march else (out["kernels"]["dia_rows"] > 100), out["kernels"]
            assert out["num_fp_it"] == ref.num_fp_it
            np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-8)
            for m in range(ref.PGD_modes):
                assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-7 * np.linalg.norm(ref_x[m])
    finally:
        if saved is None:
            os.environ.pop("PGD_TUNE", None)
        else:
            os.environ["PGD_TUNE"] = saved
    for other in (0, 2, 3):
        assert outs[other]["amplitude"] == outs[1]["amplitude"]
        for m in range(ref.PGD_modes):
            assert np.array_equal(outs[other]["modes_x"][m], outs[1]["modes_x"][m])


def test_sharded_solve_on_slabs_of_the_bench_plane():
    """Two ranks with 34 planes of 256 x 256 vertices each - the plane of the bench grid, a slab like the ranks of a multi-GPU
    run own: here the product is one stencil march over all owned planes with the ghost planes staged as data (the coded march
    + the boundary planes in row order where the operator is not one stencil), the scalar step sits in the update kernel and the
    true residual norm is measured only in the exact phase.  Must reproduce the unsharded run."""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    # (PGD_TEST_SLAB_SHAPE=255,255,255 runs the same check on the whole bench grid split in two - a one-off, minutes not seconds)
    shape = tuple(int(t) for t in os.environ.get("PGD_TEST_SLAB_SHAPE", "255,255,67").split(","))
    old = fem._backend
    fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 17, PGD_nmax=3))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = int(os.environ.get("PGD_TEST_SLAB_WORLD", "2"))      # (at most 5: the GPU box allows six processes on the card)
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, shape, q, True)) for r in range(world)]
    out = _collect(procs, q, 1, 600)[0]
    # (the product: ONE march over all owned planes in the stencil form of each rank's own planes - verified on them, the ghost
    # planes staged as data - or, on the dictionary form, the interior rows' march + the boundary planes in row order)
    assert out["kernels"]["diac_march"] + out["kernels"]["stencil_march"] > 100
    assert out["kernels"]["dia_rows"] < 20 if out["kernels"]["stencil_march"] > 100 else out["kernels"]["dia_rows"] > 100
    assert out["num_fp_it"] == ref.num_fp_it
    np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-8)
    for m in range(ref.PGD_modes):
        assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-7 * np.linalg.norm(ref_x[m])


def test_four_way_separation_with_a_sharded_space_on_the_gpu():
    """BASELINE config 5 in small with the HIP kernels: space x time x two parameters, the space row-sharded over two processes on GPU 0
    (in-library loop), the other three dimensions whole on every rank.  Must reproduce the unsharded run."""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    shape = (12, 10, 15)
    old = fem._backend
    fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.transient_heat(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 9, 5, PGD_nmax=4))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, 2, port, shape, q, True, "heat4")) for r in range(2)]
    out = _collect(procs, q, 1, 600)[0]
    assert out["pcg_iterations"] > 50
    assert out["num_fp_it"] == ref.num_fp_it
    np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-7)
    for m in range(ref.PGD_modes):
        assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-6 * np.linalg.norm(ref_x[m])


def test_p2_on_a_sharded_mesh_on_the_gpu():
    """P2 on the row-sharded box with the HIP kernels (two processes on GPU 0): the slab's nodes are numbered plane by plane, the halo
    has different sizes per direction, the P2 atoms are assembled on the slab's dof mesh and the loop driven from Python solves on
    them.  Must reproduce the unsharded P2 run at the vertices."""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    shape = (10, 8, 13)
    old = fem._backend
    fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 9, PGD_nmax=3, degree=2))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, 2, port, shape, q, True, "p2")) for r in range(2)]
    out = _collect(procs, q, 1, 600)[0]
    assert out["pcg_iterations"] > 50
    assert out["num_fp_it"] == ref.num_fp_it
    np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-7)
    for m in range(ref.PGD_modes):
        assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-6 * np.linalg.norm(ref_x[m])


@pytest.mark.parametrize("in_library", [True, False])
def test_vector_valued_space_on_a_sharded_mesh_on_the_gpu(in_library):
    """problems.elastic_block - a VECTOR-valued P1 space - on the row-sharded box with the HIP kernels (two processes on GPU 0): the
    blocked layout's dofs are partitioned like its nodes, times three; the in-library sharded loop (or the loop driven from Python)
    takes the blocked operator through its CSR products.  Must reproduce the unsharded run."""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    shape = (14, 10, 17)
    old = fem._backend
    fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.elastic_block(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 7, PGD_nmax=3))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.vector()[:].copy() for f in ref.PGD_func[0]]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, 2, port, shape, q, in_library, "elastic")) for r in range(2)]
    out = _collect(procs, q, 1, 600)[0]
    assert out["pcg_iterations"] > 50
    assert out["num_fp_it"] == ref.num_fp_it
    np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-7)
    for m in range(ref.PGD_modes):
        assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-6 * np.linalg.norm(ref_x[m])


def test_nonsymmetric_spatial_systems_on_a_sharded_mesh_on_the_gpu():
    """A convective term on the row-sharded spatial dimension with the HIP kernels (two processes on GPU 0): TorchComm.bicgstab over
    the library's vector primitives and CSR product must reproduce the unsharded run (in-library BiCGStab, csrc/pgd_krylov.hip)."""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    shape = (20, 18, 23)
    old = fem._backend
    fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.convection_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 7, 6, PGD_nmax=3))
        ref.solve_PGD(_problem="linear")
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_gpu_worker, args=(r, 2, port, shape, q, True, "convection")) for r in range(2)]
    out = _collect(procs, q, 1, 600)[0]
    assert out["bicgstab_iterations"] > 10
    assert out["num_fp_it"] == ref.num_fp_it
    np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-7)
    for m in range(ref.PGD_modes):
        assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-6 * np.linalg.norm(ref_x[m])


@pytest.mark.parametrize("world,shape,tune,ar", [(2, (127, 127, 99), "", "1"), (2, (127, 127, 99), "49=0", "0"), (3, (24, 20, 29), "", "0"),
                                                  (3, (24, 20, 29), "", "1")])
def test_direct_halo_between_processes_on_one_gpu(world, shape, tune, ar, monkeypatch):
    """PGD_HALO_DIRECT=1: the boundary planes of the search direction go straight into the NEIGHBOUR PROCESS's ghost planes through
    hipIpcMemHandle-mapped pointers, a sequence number is posted behind them and the product waits for its own (pgd_comm_push_*) -
    `world` processes on GPU 0, everything else of the exchange over gloo as in the tests above.  Only the transport of the planes
    differs: the run must be bit for bit the one with the binding's exchange.  (Planes with an even number of rows - the first
    shape - send from the update kernel itself, PGD_TUNE_PUSH_IN_UPDATE; "49=0" and odd planes from k_halo_push.)"""
    import torch.multiprocessing as mp
    if tune:
        monkeypatch.setenv("PGD_TUNE", tune)
    # (ar = "1": the loop's five sums through the ranks' IPC-mapped mailboxes as well - pgd_comm_allreduce_attach - added in rank order:
    # with two ranks the same bits as any all-reduce, with three equal to rounding)
    monkeypatch.setenv("PGD_ALLREDUCE_DIRECT", ar)
    ctx = mp.get_context("spawn")
    outs = {}
    saved = os.environ.get("PGD_HALO_DIRECT")
    try:
        for direct in ("0", "1"):
            os.environ["PGD_HALO_DIRECT"] = direct
            q = ctx.Queue()
            port = _free_port()
            procs = [ctx.Process(target=_shared_gpu_worker, args=(r, world, port, shape, q, True)) for r in range(world)]
            outs[direct] = _collect(procs, q, 1, 600)[0]
    finally:
        if saved is None:
            os.environ.pop("PGD_HALO_DIRECT", None)
        else:
            os.environ["PGD_HALO_DIRECT"] = saved
    a, b = outs["0"], outs["1"]
    assert not a["direct_halo"] and not a["direct_halo_used"]
    assert b["direct_halo"] and b["direct_halo_used"], b
    assert b["stats"]["halo"] < a["stats"]["halo"] - 100           # the products' exchanges no longer come through the callback
    assert b["direct_allreduce"] == (ar == "1") and b["direct_allreduce_used"] == (ar == "1")
    if ar == "1":
        assert b["stats"]["allreduce"] < a["stats"]["allreduce"] - 100    # ... nor the iterations' all-reduces
    assert a["num_fp_it"] == b["num_fp_it"]
    if ar == "0" or world == 2:
        assert a["amplitude"] == b["amplitude"]
        for xa, xb in zip(a["modes_x"], b["modes_x"]):
            assert np.array_equal(xa, xb)
    else:
        np.testing.assert_allclose(a["amplitude"], b["amplitude"], rtol=1e-9)
        for xa, xb in zip(a["modes_x"], b["modes_x"]):
            assert np.linalg.norm(xa - xb) <= 1e-8 * np.linalg.norm(xa)


FAULT_CASES = (("iteration 7", 15, 7), ("stage 1", 33, 1), ("stage 2", 33, 2), ("stage 3", 33, 3), ("stage 4", 33, 4))


def _faulty_worker(rank, world, port, shape, q):
    """Rank 1 fails rank-locally (injected) at one point of a sharded solve after the other: inside the loop, right after the
    setup vote, between the setup's halo exchanges, between the setup all-reduce and the loop, after the loop.  Then one
    solve without a fault: the contexts are still good."""
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = []
    try:
        from pgdrome_amd import dist as pdist, fem, problems
        from pgdrome_amd.hip_backend import HipBackend
        from pgdrome_amd.solver import PGDProblem
        torch.cuda.set_device(0)
        tstream = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(tstream)
        be = fem.set_backend(HipBackend(0, tstream.cuda_stream))
        comm = pdist.TorchComm(dist, be, in_library=True)
        be.comm_timeout(120.0)
        P = fem.Point
        mesh = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), *shape)
        assert comm.direct_halo == (os.environ.get("PGD_HALO_DIRECT") == "1")
        for name, knob, value in FAULT_CASES + (("none", 0, 0),):
            p = PGDProblem(**problems.reaction_diffusion(mesh, 17, PGD_nmax=2))
            if rank == 1 and knob:
                be.ctx.tune(knob, value)
            t0 = time.time()
            try:
                p.solve_PGD(_problem="linear")
                msg = "no error"
            except Exception as e:      # noqa: BLE001 - the point of the test
                msg = "%s: %s" % (type(e).__name__, e)
            out.append((name, msg, time.time() - t0, getattr(p, "PGD_modes", None)))
            dist.barrier()
    finally:
        q.put((rank, out))
        dist.destroy_process_group()


@pytest.mark.parametrize("direct", ["0", "1"])
def test_a_rank_failing_anywhere_in_a_solve_takes_the_others_out_with_an_error(direct, monkeypatch):
    """(direct = "1": with the direct halo - a rank that failed locally still pushes its planes and posts its numbers.)
    A rank-local failure at ANY point of the in-library sharded solve after its setup vote must end the solve on EVERY rank
    with an error, promptly: the failing rank keeps issuing the protocol's collectives (NaN payloads) and votes at the next
    agreement - before the first chunk, after every chunk, at the end - where all ranks leave together (PGD_ERR_PEER on the
    healthy ones), instead of leaving its neighbours blocked in a halo exchange or an all-reduce."""
    import torch.multiprocessing as mp
    monkeypatch.setenv("PGD_HALO_DIRECT", direct)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_faulty_worker, args=(r, 2, port, (16, 12, 21), q)) for r in range(2)]
    got = dict(_collect(procs, q, 2, 400))
    assert len(got[0]) == len(got[1]) == len(FAULT_CASES) + 1
    for (name, m0, t0, _), (_, m1, t1, _) in zip(got[0][:-1], got[1][:-1]):
        assert "injected fault" in m1, (name, m1)
        assert "another rank failed" in m0 and "error -8" in m0, (name, m0)
        assert t0 < 60 and t1 < 60, (name, t0, t1)
    assert got[0][-1][1] == got[1][-1][1] == "no error" and got[0][-1][3] == 2


def _stalled_worker(port, q):
    """One rank whose stream stops making progress in front of the first chunk (a kernel that spins for 4 s): the library's
    deadline (1 s here) must end the solve with PGD_ERR_TIMEOUT and a one-line diagnosis."""
    import time
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["PGD_COMM_TIMEOUT_ACTION"] = "raise"      # (the default, os._exit(3), is covered on the CPU: tests/test_dist_cpu.py)
    torch.cuda.set_device(0)
    # (RCCL binding: its collectives are queued on the stream and the host waits only at the agreements - where the deadline is;
    # the callback binding of the other tests blocks inside gloo instead)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    msg, dt = "no error", 0.0
    try:
        from pgdrome_amd import dist as pdist, fem, problems
        from pgdrome_amd.hip_backend import HipBackend
        from pgdrome_amd.solver import PGDProblem
        torch.cuda.set_device(0)
        tstream = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(tstream)
        be = fem.set_backend(HipBackend(0, tstream.cuda_stream))
        comm = pdist.TorchComm(dist, be, in_library=True)
        assert comm.in_library == "rccl"
        P = fem.Point
        mesh = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), 16, 12, 21)
        p = PGDProblem(**problems.reaction_diffusion(mesh, 17, PGD_nmax=1))
        be.comm_timeout(1.0)
        be.ctx.tune(34, 4000)
        t0 = time.time()
        try:
            p.solve_PGD(_problem="linear")
        except Exception as e:      # noqa: BLE001
            msg = "%s: %s (code %s)" % (type(e).__name__, e, getattr(e, "code", None))
        dt = time.time() - t0
        torch.cuda.synchronize()          # the bounded stall drains before the process ends
    finally:
        q.put((msg, dt))
        dist.destroy_process_group()


def test_a_stream_without_progress_runs_into_the_deadline():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    pr = ctx.Process(target=_stalled_worker, args=(_free_port(), q))
    msg, dt = _collect([pr], q, 1, 300)[0]
    assert "code -7" in msg and "no progress" in msg and "rank 0/1" in msg and "last collective issued" in msg, msg
    assert 0.9 < dt < 30, dt


def _shared_gpu_mg_worker(rank, world, port, shape, q):
    """Ranks sharing GPU 0, settings["preconditioner"] = "amg" on the sharded mesh: the slab pieces of the V-cycle are the HIP
    kernels (pgd_mg_slab_*), halos and the level-1 all-reduce go through gloo (staged through the host)."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pgdrome_amd import dist as pdist, fem, problems
        from pgdrome_amd.hip_backend import HipBackend
        from pgdrome_amd.solver import PGDProblem
        torch.cuda.set_device(0)
        tstream = torch.cuda.Stream(device=0)
        torch.cuda.set_stream(tstream)
        be = fem.set_backend(HipBackend(0, tstream.cuda_stream))
        comm = pdist.TorchComm(dist, be, in_library=False)
        P = fem.Point
        mesh = pdist.sharded_box_mesh(comm, P(0, 0, 0), P(1, 1, 1), *shape)
        p = PGDProblem(**problems.reaction_diffusion(mesh, 17, PGD_nmax=3))
        i0 = fem.STATS["pcg_iterations"]
        p.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "preconditioner": "amg", "relative_tolerance": 1e-10})
        modes_x = [pdist.gather_owned(comm, mesh, f.compute_vertex_values()) for f in p.PGD_func[0]]
        if rank == 0:
            q.put(dict(num_fp_it=p.num_fp_it, amplitude=p.amplitude, modes_x=modes_x, stats=dict(comm.stats),
                       pcg_iterations=fem.STATS["pcg_iterations"] - i0, mg_solves=fem.STATS.get("mg_solves", 0),
                       kernels=be.ctx.kernel_counts()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("world,shape,tune", [(2, (40, 36, 45), ""), (3, (31, 33, 38), ""), (2, (40, 36, 45), "42=16"), (3, (70, 65, 40), "")])
def test_sharded_v_cycle_on_one_gpu(world, shape, tune):
    """The V-cycle on a row-sharded lattice with the HIP kernels and real exchanges (`world` processes share GPU 0): the PGD run
    under settings["preconditioner"] = "amg" reproduces the UNSHARDED run under the same setting - pass counts, amplitudes,
    modes to 1e-7 - with the same number of PCG iterations (+-1 per solve: the dots are grouped by rank), every spatial solve
    preconditioned by the cycle.  (tune "42=16" / the 71 x 66 planes: level 0 of the slabs in the stencil march of the product -
    ghost planes staged as data - instead of the plain slab kernels.)"""
    import torch.multiprocessing as mp
    from pgdrome_amd import fem, problems
    from pgdrome_amd.hip_backend import HipBackend
    from pgdrome_amd.solver import PGDProblem
    old = fem._backend
    be = fem.set_backend(HipBackend(0))
    fem.clear_caches()
    try:
        P = fem.Point
        ref = PGDProblem(**problems.reaction_diffusion(fem.BoxMesh(P(0, 0, 0), P(1, 1, 1), *shape), 17, PGD_nmax=3))
        i0, m0 = fem.STATS["pcg_iterations"], be.ctx.mg_stats()
        ref.solve_PGD(_problem="linear", settings={"linear_solver": "cg", "preconditioner": "amg", "relative_tolerance": 1e-10})
        its_ref, m1 = fem.STATS["pcg_iterations"] - i0, be.ctx.mg_stats()
        ref_x = [f.compute_vertex_values() for f in ref.PGD_func[0]]
        solves = sum(int(v) for v in ref.num_fp_it)
        assert m1["solves"] - m0["solves"] == solves and m1["fallbacks"] == m0["fallbacks"]
    finally:
        fem.set_backend(old)
        fem.clear_caches()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    saved = os.environ.get("PGD_TUNE")
    try:
        if tune:
            os.environ["PGD_TUNE"] = tune
        procs = [ctx.Process(target=_shared_gpu_mg_worker, args=(r, world, port, shape, q)) for r in range(world)]
        out = _collect(procs, q, 1, 300)[0]
    finally:
        if saved is None:
            os.environ.pop("PGD_TUNE", None)
        else:
            os.environ["PGD_TUNE"] = saved
    assert out["num_fp_it"] == ref.num_fp_it and out["stats"].get("sharded_mg_solves", 0) == solves
    if tune or min(shape[0], shape[1]) >= 63:
        assert out["kernels"]["stencil_march"] > 2 * out["pcg_iterations"]        # two level-0 passes per cycle in the march
    assert abs(out["pcg_iterations"] - its_ref) <= solves, (out["pcg_iterations"], its_ref)
    assert out["pcg_iterations"] <= 25 * solves
    np.testing.assert_allclose(out["amplitude"], ref.amplitude, rtol=1e-8)
    for m in range(ref.PGD_modes):
        assert np.linalg.norm(out["modes_x"][m] - ref_x[m]) <= 1e-7 * np.linalg.norm(ref_x[m])
