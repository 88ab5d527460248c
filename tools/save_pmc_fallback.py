"""Write profiles/pmc_spmv_latest.json - bench.py's fallback for roofline.traffic when no profiler child can be started -
from the PMC traffic a bench line of this round measured itself.

    python tools/save_pmc_fallback.py gpurun_out/r02v_bench.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
line = open(sys.argv[1]).read().strip().splitlines()[-1]
d = json.loads(line)
r = d["roofline"]
prod = r.get("spmv", r)
out = {"hbm_bytes_per_launch": prod["traffic"], "kernel": prod["kernel"][:160],
       "kernel_min_bytes_per_launch": prod["bytes_per_launch"],
       "source": "%s: %s" % (os.path.basename(sys.argv[1]), prod["traffic_source"])}
if "spmv" in r and r.get("traffic"):
    out["update_hbm_bytes_per_launch"] = r["traffic"]
    out["update_kernel"] = r["kernel"][:160]
    out["update_kernel_min_bytes_per_launch"] = r["bytes_per_launch"]
assert "rocprofv3 --pmc child processes of this run" in prod["traffic_source"], "the line carries fallback numbers itself"
with open(os.path.join(ROOT, "profiles", "pmc_spmv_latest.json"), "w") as f:
    json.dump(out, f, indent=1)
    f.write("\n")
print(json.dumps(out, indent=1))
