"""Stands in for `python -m torch.distributed.run` in tests/test_bench_launcher.py: records how bench.py's launcher called it
(argv, the rendezvous environment, which heavy modules the PARENT had loaded) and plays N ranks: rank 0's JSON line on stdout,
noise around it, an exit status of choice."""
import json
import os
import sys

rec = {"argv": sys.argv[1:], "env": {k: os.environ.get(k) for k in ("MASTER_ADDR", "MASTER_PORT", "PGD_BENCH_LAUNCHED",
                                                                     "PGD_BENCH_FORCE_LAUNCHER", "HSA_ENABLE_IPC_MODE_LEGACY")},
       "cwd": os.getcwd()}
with open(os.environ["PGD_STUB_RECORD"], "w") as f:
    json.dump(rec, f)
print("NCCL version 2.x (banner on stdout, as RCCL prints it)")
mode = os.environ.get("PGD_STUB_MODE", "ok")
if mode != "silent":
    n = int(sys.argv[sys.argv.index("--nproc-per-node") + 1])
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": n}))
print("trailing noise")
sys.exit(int(os.environ.get("PGD_STUB_RC", "0")))
