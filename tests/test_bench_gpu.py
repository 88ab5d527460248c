"""bench.py's one-line contract on a small instance of the bench problem (104^3: above the 2^20 rows that switch the
single-sync recurrence on): the fields the driver reads, a physical roofline of the dominant kernel with the product's
beside it, the CSR products of the north star timed in the same run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n", "104", "--steps", "3", "--warmup", "1", "--no-pmc",
           "--cpu-seconds", "1"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                      # exactly ONE line on stdout
    d = json.loads(lines[0])
    for key, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str)):
        assert isinstance(d[key], typ), key
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f64"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] / 1e3 - 1.0) < 1e-9
    assert "workload" in d["config"] and d["config"]["us_per_pcg_iteration"] > 0
    assert d["config"]["product_launches_by_kernel"]["diac_march"] > 0          # the row-class dictionary is what ran
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    spmv = r.get("spmv", r)                            # the product's entry: under roofline.spmv when the vector update dominates
    assert "k_spmv_diac_march2" in spmv["kernel"] and spmv["bytes_per_row"] == 17.0 and 0 < spmv["frac"] <= 1.0
    csr = spmv["csr_product"]
    assert csr["k_spmv_csr<dot,store,64>"]["launches_timed"] >= 100 and 0 < csr["k_spmv_csr<dot,store,64>"]["frac"] <= 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
