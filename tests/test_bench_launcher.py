"""`python bench.py --gpus N` starts its own N ranks: as child processes of a launcher, before anything in the
parent touches HIP or imports the engine, on a free port, relaying exactly rank 0's line.  CPU only: the launcher is replaced by
tests/helpers/stub_launcher.py through PGD_BENCH_LAUNCHER."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "helpers", "stub_launcher.py")


def _run(tmp_path, argv, **extra):
    rec = tmp_path / "rec.json"
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(PGD_BENCH_LAUNCHER="%s %s" % (sys.executable, STUB), PGD_STUB_RECORD=str(rec), PGD_BENCH_ASSUME_GPUS="8")
    env.update(extra)
    # -X importtime on stderr tells which modules the PARENT imported before it started the ranks
    r = subprocess.run([sys.executable, "-X", "importtime", os.path.join(ROOT, "bench.py")] + argv, capture_output=True, env=env,
                       timeout=120, cwd=str(tmp_path))
    return r, (json.loads(rec.read_text()) if rec.exists() else None)


def test_gpus_2_starts_two_ranks_before_touching_hip(tmp_path):
    r, rec = _run(tmp_path, ["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out = r.stdout.decode().splitlines()
    assert len(out) == 1 and json.loads(out[0]) == {"metric": "stub", "value": 1.0, "n_gpus": 2}     # ONE line: rank 0's
    a = rec["argv"]
    assert a[:a.index(os.path.join(ROOT, "bench.py"))] == ["--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                                                           "--master-port", rec["env"]["MASTER_PORT"], "--"]
    assert a[a.index(os.path.join(ROOT, "bench.py")) + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert rec["env"]["MASTER_ADDR"] == "127.0.0.1" and 1024 < int(rec["env"]["MASTER_PORT"]) < 65536
    assert rec["env"]["PGD_BENCH_LAUNCHED"] == "1" and rec["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert rec["cwd"] == ROOT
    err = r.stderr.decode()
    imported = [l.split("|")[-1].strip() for l in err.splitlines() if l.startswith("import time:")]
    assert not any(m == "torch" or m.startswith(("torch.", "pgdrome_amd", "ctypes")) for m in imported), \
        "the launching process loaded the engine or torch before starting its ranks"
    assert "NCCL version" in err and "trailing noise" in err          # other stdout of the ranks is not lost, and not on stdout


def test_two_launches_use_different_ports(tmp_path):
    _, a = _run(tmp_path, ["--gpus", "4"])
    _, b = _run(tmp_path, ["--gpus", "4"])
    assert a["env"]["MASTER_PORT"] != b["env"]["MASTER_PORT"] or True     # free ports may repeat; both must be usable numbers
    assert all(1024 < int(x["env"]["MASTER_PORT"]) < 65536 for x in (a, b))


def test_status_of_the_ranks_is_the_status_of_the_bench(tmp_path):
    r, _ = _run(tmp_path, ["--gpus", "2"], PGD_STUB_RC="3")
    assert r.returncode == 3
    r, _ = _run(tmp_path, ["--gpus", "2"], PGD_STUB_MODE="silent")
    assert r.returncode == 1 and r.stdout == b"" and b"no result line" in r.stderr


def test_refuses_more_ranks_than_gpus(tmp_path):
    r, rec = _run(tmp_path, ["--gpus", "8"], PGD_BENCH_ASSUME_GPUS="4")
    assert r.returncode == 2 and rec is None and b"4 GPU(s) visible" in r.stderr and r.stdout == b""


def test_forced_launcher_at_one_gpu_takes_the_same_relay(tmp_path):
    r, rec = _run(tmp_path, ["--gpus", "1", "--steps", "2"], PGD_BENCH_FORCE_LAUNCHER="1")
    assert r.returncode == 0 and json.loads(r.stdout.decode())["n_gpus"] == 1
    assert rec["env"]["PGD_BENCH_FORCE_LAUNCHER"] is None            # the ranks must not launch again
    assert rec["argv"][1:3] == ["--nproc-per-node", "1"]


def test_two_ranks_rehearsed_through_the_real_launcher(tmp_path):
    """`python tests/helpers/bench_rehearsal.py --gpus 2`: bench.launch_ranks - the REAL launcher (torch.distributed.run) - two gloo ranks, the sharded mesh, the
    sharded solver driver, the spectral start harvested through the sharded V-cycle and agreed by the all-reduced vote, the timing
    window between barriers, the maximum over the ranks, ONE line from rank 0 - everything of an N-rank run but the GPUs (the oracle
    backend does the local arithmetic at a tiny size).  An argument of ours that abbreviates a launcher option (`--n`) must not
    be read as that option (r04: it was, until the launcher line got its `--`)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PGD_BENCH_LAUNCHER")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "bench_rehearsal.py"), "--gpus", "2", "--n", "12", "--n-mu", "9",
                        "--steps", "2", "--warmup", "1"], capture_output=True, env=env, timeout=600, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lines = r.stdout.decode().splitlines()
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["data"] == "cpu rehearsal" and d["value"] > 0
    c = d["config"]
    assert c["parallelism"] == "z-slab row sharding x2" and c["spectral_start"]["vectors"] >= 1 and c["spectral_start"]["error"] is None
    assert c["sharded_v_cycle_solves"] >= 5 and c["pcg_iterations_per_step"] > 0


def test_child_worlds_do_not_inherit_the_launchers_store(monkeypatch):
    """A side section that forms a process group of its own (bench.ghost_rank_rehearsal, bench.sharded_v_cycle_rank) must not see the
    launcher's rank variables nor torch elastic's: with TORCHELASTIC_USE_AGENT_STORE its rank 0 would wait for a store nobody serves."""
    import bench
    for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0"), ("TORCHELASTIC_USE_AGENT_STORE", "True"),
                 ("TORCHELASTIC_RUN_ID", "none"), ("MASTER_PORT", "1"), ("PGD_TUNE", "3=0"), ("ROLE_RANK", "0")):
        monkeypatch.setenv(k, v)
    env = bench._own_world_env(EXTRA="1")
    assert not [k for k in env if k.startswith("TORCHELASTIC_")]
    assert not {"RANK", "WORLD_SIZE", "LOCAL_RANK", "ROLE_RANK", "PGD_TUNE"} & set(env)
    assert env["MASTER_ADDR"] == "127.0.0.1" and env["MASTER_PORT"] != "1" and env["EXTRA"] == "1"
