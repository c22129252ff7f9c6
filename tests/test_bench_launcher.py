"""`python bench.py --gpus N` as typed: with no torchrun around it the parent starts the N ranks itself, forwards
rank 0's one JSON line and exits with the worst child's code.  CPU: the launch / rendezvous / reduction plumbing over gloo
(--launcher-selftest does no GPU work).  GPU: the real bench through the same spelling, gloo on one card, nccl when there are two."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


@pytest.mark.timeout(180)
@pytest.mark.parametrize("n", [1, 2, 3, 8])             # 8 = the node size the driver's scaling run uses (BASELINE.json configs[2] / [4])
def test_bench_starts_its_own_ranks(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launcher-selftest"], env=_env(AQ_DIST_BACKEND="gloo"),
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                       # ONE line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["max_over_ranks"] == float(n)


@pytest.mark.timeout(180)
def test_a_failing_rank_fails_the_launch_and_prints_no_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=_env(AQ_DIST_BACKEND="gloo", AQ_SELFTEST_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=170)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert r.stdout.strip() == ""
    assert "exit codes" in r.stderr


@pytest.mark.timeout(60)
def test_under_torchrun_the_world_size_must_match():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=50)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus_2_as_typed_gloo_on_one_card():
    """Two ranks sharing cuda:0 (LOCAL_RANK is taken modulo the device count), gloo for the gather: the N > 1 code path of the bench
    end to end, started by `python bench.py --gpus 2` itself."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "8", "--pool", "2", "--no-autotune",
                        "--no-cpu-baseline", "--parity-steps", "0"], env=_env(AQ_DIST_BACKEND="gloo"), capture_output=True, text=True, timeout=880)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["scaling"] == "weak"
    assert out["value"] > 0 and out["config"]["detections_gathered"] > 0
    # what the first real multi-GPU run needs to be readable: the world size the backend saw, per-rank rates, the gather's share
    ranks = out["config"]["ranks"]
    assert ranks["dist_world_size"] == 2 and ranks["backend"] == "gloo" and out["config"]["collectives"] == "gloo"
    assert 0 < ranks["per_rank_tiles_per_s"]["min"] <= ranks["per_rank_tiles_per_s"]["max"] and ranks["per_rank_tiles_per_s"]["slowest_rank"] in (0, 1)
    assert 0.0 <= ranks["gather_share_of_timed_region"] < 1.0
    assert out["value"] <= 2 * ranks["per_rank_tiles_per_s"]["max"] * 1.001


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_gpus_2_as_typed_nccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL)")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--batch", "8", "--pool", "2", "--no-autotune",
                        "--no-cpu-baseline", "--parity-steps", "0"], env=_env(), capture_output=True, text=True, timeout=880)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["n_gpus"] == 2 and out["config"]["detections_gathered"] > 0
