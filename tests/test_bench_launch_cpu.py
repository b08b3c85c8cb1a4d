"""`python bench.py --gpus N` must START by itself: with no launcher environment it re-executes under
torch.distributed.run, every rank rendezvouses on 127.0.0.1, the time is the max over the ranks and rank 0 prints ONE JSON line.
Checked here over gloo without a GPU (GENNBV_BENCH_DRY=1 skips the hot path, nothing else)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True, timeout=300)


def test_gpus2_self_launches_and_prints_one_json_line():
    r = _run({"GENNBV_BENCH_DRY": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["max_rank_seconds"] >= 0.02  # rank 1 sleeps 20 ms: the max over the ranks, not rank 0's own time
    assert "torch.distributed.run" in r.stderr  # the launcher line goes to stderr, never in front of the JSON


def test_gpus1_needs_no_launcher():
    r = _run({"GENNBV_BENCH_DRY": "1"}, "--gpus", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    assert json.loads(r.stdout.strip().splitlines()[-1])["n_gpus"] == 1


def test_without_gpus_the_real_bench_fails_loudly():
    r = _run({}, "--gpus", "1", "--steps", "1", "--warmup", "0")
    assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)
