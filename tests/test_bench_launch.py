"""
`python bench.py --gpus N` must really start N ranks (the driver's command form has no launcher in front of it).
CPU rehearsal: --dry-run goes through the same spawn -> torch.distributed.run -> rendezvous -> barrier ->
max-over-ranks -> one JSON line from rank 0, on gloo, without a GPU and without kernels (value null, "dry_run": true).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=300, env=env)


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_flag_starts_that_many_ranks(n):
    proc = _run("--gpus", str(n), "--steps", "3", "--warmup", "0", "--dry-run")
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, proc.stdout  # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == n and rec["dry_run"] is True and rec["value"] is None
    assert rec["config"]["global_batch"] == n * 16
    # the job's clock is its slowest rank: rank r sleeps (1 + r) ms per step
    assert rec["ms_per_step"] >= n * 1.0


def test_single_rank_dry_run_and_mismatch():
    proc = _run("--steps", "2", "--warmup", "0", "--dry-run")
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert json.loads(proc.stdout.strip())["n_gpus"] == 1
    # under a launcher that started a different number of ranks than --gpus says: refuse, do not mislabel
    proc = _run("--gpus", "4", "--steps", "2", "--dry-run", env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert proc.returncode != 0 and "--gpus 4" in proc.stderr


def test_failing_rank_fails_the_job():
    # rank 1 dies: the launcher takes the job down, the parent must not print a line and must fail
    proc = _run("--gpus", "2", "--dry-run", "--steps", "2", env_extra={"GPT_BENCH_FAIL_RANK": "1"})
    assert proc.returncode != 0
    assert not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
