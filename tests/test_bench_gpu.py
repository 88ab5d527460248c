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
    assert d["config"]["product_launches_by_kernel"]["stencil_march"] > 0       # the stencil form of the row-class dictionary is what ran
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9) <= 1e-6 * r["achieved"]
    spmv = r.get("spmv", r)                            # the product's entry: under roofline.spmv when the vector update dominates
    assert "k_spmv_stencil_march" in spmv["kernel"] and spmv["bytes_per_row"] == 16.0 and 0 < spmv["frac"] <= 1.0
    csr = spmv["csr_product"]
    assert csr["k_spmv_csr<dot,store,64>"]["launches_timed"] >= 100 and 0 < csr["k_spmv_csr<dot,store,64>"]["frac"] <= 1.0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == d["unit"]
    # the engine without the row-class dictionary and on the CSR kernels, timed in the same run
    gp = d["config"]["general_paths"]
    assert gp["plain_march"]["product_kernel"] in ("dia_march", "dia_rows") and gp["csr"]["product_kernel"] == "csr_dict"
    assert gp["row_class_dictionary"]["product_kernel"] == "diac_march"
    assert 0 < gp["csr"]["passes_per_s"] <= gp["plain_march"]["passes_per_s"] * 1.2 and gp["plain_march"]["product_us"] > 0
    assert spmv["speedup_over_plain_diagonal_form_this_run"] > 1.0
    # the same workload with settings["preconditioner"] = "amg": every spatial solve through the V-cycle, the same modes
    mg = d["config"]["multigrid_preconditioner"]
    assert mg["solves_preconditioned_by_the_v_cycle"] >= mg["passes"] and mg["pcg_iterations_per_pass"] < 40
    assert mg["passes_per_s"] > d["value"] and mg["modes_compared_with_the_headline_run"] >= 1
    assert mg["worst_relative_l2_difference_of_a_spatial_mode"] <= 1e-6
    assert isinstance(d["config"]["launch_timing_samples_dropped_as_noops"], int)
    assert "Infinity Cache" in r.get("note", "") or "spmv" not in r


def test_bench_line_of_the_sharded_driver_has_the_phase_timings():
    """bench.py --dist-driver (one rank through the in-library sharded loop over RCCL): the fields a multi-GPU line carries -
    per-rank max / min of the iteration's phases, the communicator's world size, the deadline."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--n", "104", "--steps", "3", "--warmup", "1", "--no-pmc",
           "--no-cpu-baseline", "--dist-driver"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    c = d["config"]
    assert c["sharded_pcg_driver"] == "in-library loop, RCCL" and c["rccl_world"] == 1 and c["comm_timeout_s"] == 60.0
    ph = c["sharded_iteration_phases"]
    for k in ("halo_wait_us", "allreduce_us", "product_interior_us", "product_boundary_us", "update_us", "local_sums_us"):
        assert ph[k]["max"] >= ph[k]["min"] >= 0.0, k
    assert ph["samples"]["min"] >= 10 and ph["product_interior_us"]["max"] > 1.0 and ph["update_us"]["max"] > 1.0
    assert ph["allreduce_us"]["max"] > 0.0


def test_rank_with_ghost_planes_rehearsal():
    """bench.py's side section config.rank_with_ghost_planes (attached at the metric's size): tools/bench_self_periodic.py --json -
    one rank as its own neighbour, real RCCL send / receive inside the sharded loop, three ways of running the iteration."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "bench_self_periodic.py"), "--json"]
    res = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29549"))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["owned_planes"] == 32 and d["rows"] == 256 * 256 * 32
    one, split, over = (d[k] for k in ("stream_ordered_one_march", "stream_ordered_interior_plus_boundary", "overlapped_on_the_halo_stream"))
    assert not one["second_stream_used"] and not split["second_stream_used"] and over["second_stream_used"] and over["second_stream_available"]
    assert one["iterations"] == split["iterations"] > 100 and abs(over["iterations"] - one["iterations"]) <= 1
    assert 10.0 < one["us_per_iteration"] <= 1.05 * split["us_per_iteration"]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--preconditioner", "amg"], ["--direct-halo"]])
def test_two_ranks_of_the_bench_on_one_gpu(extra):
    """`python bench.py --gpus 2 --share-one-gpu`: bench.main's N > 1 code - the launcher, the sharded mesh, the spectral start's vote,
    the timing window between barriers, the maximum over the ranks, the gathered phase timings - with the HIP kernels, two ranks on
    GPU 0, the exchange steps over gloo.  A rehearsal (the line says so), at a small size."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-one-gpu", "--n", "64", "--n-mu", "33",
                        "--steps", "4", "--warmup", "2"] + extra, capture_output=True, env=env, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = [l for l in r.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["value"] > 0 and "REHEARSAL" in d["data"]
    c = d["config"]
    assert c["parallelism"].endswith("x2") and c["pcg_iterations_per_step"] > 3
    if "--direct-halo" in extra:
        assert all(v for k, v in c["direct_halo"].items() if k != "probe") and all(c["direct_halo"]["probe"].values()), c["direct_halo"]
    elif extra:
        assert c["sharded_v_cycle_solves"] > 0
    else:
        assert c["spectral_start"]["vectors"] > 0 and c["sharded_iteration_phases"]
        # (without --direct-halo the direct paths are only PROBED - in child processes, after the timed region)
        pr = c["direct_halo"]["probe"]
        assert "error" not in pr and pr["direct_halo_passed_its_checks_on_every_rank"] and pr["direct_allreduce_passed_its_checks_on_every_rank"], pr
        assert pr["microseconds"]["direct_halo"] > 0 and pr["microseconds_through_the_binding"]["halo_through_the_binding"] > 0, pr
        assert not c["direct_halo"]["used_by_the_last_solve"], c["direct_halo"]
