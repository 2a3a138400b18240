"""CPU tests of bench.py's multi-rank launch path: `python bench.py --gpus N` must start N ranks itself
(a child torchrun, before any GPU call), report n_gpus from the world size, and refuse -- loudly -- to run
ranks that would share a device."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_spawns_that_many_ranks():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing-only"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, from rank 0"
    # ... and nothing else on stdout: the chatter of the libraries below (gloo's rank lines here, RCCL's version banner on
    # a GPU node) is kept on stderr
    assert r.stdout.strip() == lines[0], r.stdout[:500]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["plumbing_only"] is True
    assert out["value"] == 0.0                      # plumbing only: never a measurement
    assert out["all_reduce_sum"] == 1 + 2           # every rank took part in the collective
    assert [rk[0] for rk in out["ranks"]] == [0, 1]


def test_more_ranks_than_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("needs a node with fewer than 2 devices")
    r = _run(["--gpus", "2", "--steps", "1"])
    assert r.returncode != 0
    assert "2 ranks" in r.stderr and "device" in r.stderr
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines()), "no JSON line may be printed"


def test_world_size_must_match_gpus_flag():
    r = _run(["--gpus", "1", "--plumbing-only"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)
