"""CPU: `python bench.py --gpus N` is an N-rank job.  The driver's recorded command for one GPU is a plain `python3 bench.py
--gpus 1 ...`; with N > 1 and no launcher around it bench.py must start the ranks itself (before it touches the GPU) and
print ONE line whose n_gpus is the world size the process group saw.  --dry-launch does the rendezvous alone (gloo here),
so the launch path is checked without a GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(argv, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    return r, lines


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_starts_n_ranks_by_itself(n):
    r, lines = run(["--gpus", str(n), "--dry-launch"], {"M2V_DIST_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, "exactly one JSON line (rank 0's): %r" % r.stdout
    assert lines[0]["n_gpus"] == n and lines[0]["ranks_seen"] == n and lines[0]["ranks_counted"] == n
    assert lines[0]["launched_by"] == "bench.py" and lines[0]["backend"] == "gloo"


def test_one_gpu_needs_no_launcher():
    r, lines = run(["--gpus", "1", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 1 and lines[0]["launched_by"] == "caller"


def test_under_torch_distributed_run_the_ranks_are_not_started_twice():
    """the driver's N > 1 form: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env["M2V_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["launched_by"] == "caller"


def test_gpus_argument_must_match_the_launcher():
    r, _ = run(["--gpus", "2", "--dry-launch"], {"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


def test_a_dying_rank_takes_the_job_down():
    """rank 1 fails before the rendezvous: the parent reports a non-zero code instead of waiting for ever"""
    r, lines = run(["--gpus", "2", "--dry-launch"], {"M2V_DIST_BACKEND": "gloo", "M2V_BENCH_TEST_FAIL_RANK": "1"}, timeout=120)
    assert r.returncode != 0
