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


def _alive(pid):
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[0] != "Z"      # a zombie nobody has reaped yet is not running
    except OSError:
        return False


def test_a_rank_that_ignores_sigterm_is_killed(tmp_path):
    """rank 1 fails, rank 0 is stuck (as in a collective the failed rank never joins) and ignores SIGTERM: the launcher escalates
    to SIGKILL instead of looping for ever (round-3 advisor), and the stuck process is really gone afterwards"""
    pidfile = str(tmp_path / "deaf.pid")
    r, _ = run(["--gpus", "2", "--dry-launch"], {"M2V_DIST_BACKEND": "gloo", "M2V_BENCH_TEST_FAIL_RANK": "1", "M2V_BENCH_TEST_DEAF_RANK": "0",
                                                 "M2V_BENCH_TEST_PIDFILE": pidfile, "M2V_BENCH_GRACE": "2,2"}, timeout=120)
    assert r.returncode != 0
    assert not _alive(int(open(pidfile).read()))


def test_terminating_the_launcher_ends_its_ranks(tmp_path):
    """SIGTERM to `bench.py --gpus 2` itself: no rank is left behind holding a GPU"""
    import signal
    import time
    pidfile = str(tmp_path / "deaf.pid")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"M2V_DIST_BACKEND": "gloo", "M2V_BENCH_TEST_DEAF_RANK": "1", "M2V_BENCH_TEST_PIDFILE": pidfile, "M2V_BENCH_GRACE": "2,2"})
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"], env=env, cwd=ROOT,
                         stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    t0 = time.time()
    while not os.path.exists(pidfile) and time.time() - t0 < 90:
        time.sleep(0.1)
    assert os.path.exists(pidfile), "the ranks never started"
    time.sleep(0.3)
    pid = int(open(pidfile).read())
    p.send_signal(signal.SIGTERM)
    assert p.wait(timeout=60) != 0
    assert not _alive(pid)


def test_more_ranks_than_visible_gpus_is_refused_before_anything_starts():
    """`bench.py --gpus 8` on a node that shows fewer GPUs must not start eight processes that then fight over them: the launcher counts
    the visible devices WITHOUT touching the HIP runtime (the *_VISIBLE_DEVICES list, else the KFD topology) and says what it found"""
    r, lines = run(["--gpus", "4", "--steps", "1"], {"HIP_VISIBLE_DEVICES": "0,1"}, timeout=60)
    assert r.returncode != 0 and not lines
    assert "--gpus 4 but 2 GPU(s) visible" in r.stderr and "nothing was started" in r.stderr


def test_visible_gpus_and_sensors_read_without_the_runtime(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "3,5,6")
    assert bench.visible_gpus() == 3
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "")
    assert bench.visible_gpus() == 0
    monkeypatch.delenv("HIP_VISIBLE_DEVICES")
    monkeypatch.delenv("ROCR_VISIBLE_DEVICES")
    n = bench.visible_gpus()                       # the KFD topology of this host (no GPU here: 0, or None where /sys/class/kfd is absent)
    assert n is None or n >= 0
    s = bench.gpu_sensors(99)
    assert s is None                               # no such card: nothing invented
