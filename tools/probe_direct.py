"""The direct halo / direct all-reduce of the sharded PCG loop (pgd_comm_push_*, pgd_comm_allreduce_*) put through their checked
exchanges and timed between the GPUs of THIS machine - in processes of their own, so that whatever they do cannot touch the
measurement of the run that started them (bench.py at N > 1 starts one of these per rank, after its timed region, and waits with a
deadline).  Rank 0 prints one JSON object.

    python tools/probe_direct.py RANK WORLD PORT DEVICE N OWN0 OWN1 LO_GHOST HI_GHOST
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rank, world, port, dev, n, own0, own1, lo_g, hi_g = (int(t) for t in sys.argv[1:10])
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import datetime

import torch
import torch.distributed as dist

torch.cuda.set_device(dev)
dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=90))
try:
    from pgdrome_amd import dist as pdist, fem
    from pgdrome_amd.hip_backend import HipBackend
    ts = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(ts)
    be = fem.set_backend(HipBackend(dev, ts.cuda_stream))
    comm = pdist.TorchComm(dist, be, in_library=True)          # gloo: the library's loop bound to callbacks - only the direct paths are probed
    comm.enable_direct_halo(n, own0, own1, lo_g, hi_g, use=False, time_binding=False)
    out = dict(comm.direct_probe or {})
    out["devices"] = "rank r on the device its bench rank uses; %d processes" % world
    if rank == 0:
        print(json.dumps(out), flush=True)
finally:
    try:
        dist.barrier()
    except Exception:       # noqa: BLE001
        pass
    dist.destroy_process_group()
