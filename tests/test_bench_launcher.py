"""bench.py's multi-GPU launcher (argument handling only; no GPU needed)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=env, timeout=300)


def test_more_gpus_than_visible_is_an_error():
    import torch
    have = torch.cuda.device_count()
    n = max(have + 1, 2)   # --gpus 1 never launches ranks
    r = _run(["--gpus", str(n)])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert f"--gpus {n} asked for but only {have} GPU(s) are visible" in r.stderr


def test_rank_count_must_match_gpus():
    # started as a rank of a 1-rank job but told --gpus 2: refuse before touching a GPU
    r = _run(["--gpus", "2"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "the launcher started 1 rank(s)" in r.stderr
    r = _run(["--gpus", "0"])
    assert r.returncode != 0 and "at least 1" in r.stderr


def test_launcher_starts_one_rank_per_gpu(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    import torch
    seen = {}
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    rc = bench.launch_ranks(4, ["--gpus", "4", "--steps", "5"])
    assert rc == 7                                   # the children's exit code is the launcher's
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [BENCH, "--gpus", "4", "--steps", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_strong_scaling_keys_of_a_multi_gpu_line():
    """An N > 1 line carries the strong-scaling number (the tile-sharded film of the same run) beside the weak-scaling
    `value`, under keys of its own that say which is which; an N = 1 line, or a run without the sharded leg, has none."""
    sys.path.insert(0, ROOT)
    import bench
    sharded = {"samples_per_sec": 1.5e9, "workload": "sponza_class 3840x2160, 8 passes, tiles of 64x64 dealt round-robin, film gathered on rank 0"}
    assert bench.strong_scaling_fields(1, sharded) == {}
    assert bench.strong_scaling_fields(4, None) == {}
    f = bench.strong_scaling_fields(4, sharded)
    assert f["value_strong"] == 1.5e9 and f["unit_strong"] == "samples/s" and f["samples_per_sec_strong"] == f["value_strong"]
    assert f["scaling_of"] == {"value": "weak", "value_strong": "strong"}
    assert f["strong_workload"] == sharded["workload"]
    line = {"metric": "m", "value": 1.0, "scaling": "weak", **f}
    assert json.loads(json.dumps(line))["value_strong"] == 1.5e9
