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
    # ... and the one line also carries config c5: the two strips legs ran as fresh ranks after the c4 ranks had exited
    st = lines[0]["strips"]
    assert set(st) == {"rccl", "peer"}
    for name in ("rccl", "peer"):
        assert st[name]["dry_launch"] is True and st[name]["ranks_seen"] == n and st[name]["mode"] == "strips" and st[name]["transport"] == name


def test_a_strips_leg_that_hangs_or_fails_never_costs_the_c4_line():
    """the peer leg never comes back (env switch): ended by its PIDs when the bound passes, reported as a timeout INSIDE the intact c4
    line, exit code 0; the RCCL leg beside it is fine.  Then a leg whose ranks exit with an error: a stated exit code, same line."""
    import time
    t0 = time.time()
    r, lines = run(["--gpus", "2", "--dry-launch"], {"M2V_DIST_BACKEND": "gloo", "M2V_BENCH_TEST_HANG_LEG": "peer", "M2V_BENCH_LEG_TIMEOUT": "25"}, timeout=400)
    took = time.time() - t0
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["ranks_counted"] == 2
    st = lines[0]["strips"]
    assert st["rccl"].get("dry_launch") is True and "error" not in st["rccl"]
    assert st["peer"]["error"] == "timeout" and 25 <= st["peer"]["seconds"] < 60          # (the bound is generous: a loaded box imports torch slowly)
    assert took < 240
    r, lines = run(["--gpus", "2", "--dry-launch"], {"M2V_DIST_BACKEND": "gloo", "M2V_BENCH_TEST_FAIL_LEG": "rccl"}, timeout=240)
    assert r.returncode == 0 and len(lines) == 1
    assert lines[0]["strips"]["rccl"]["error"] == "exit code 5" and lines[0]["strips"]["peer"].get("dry_launch") is True


def test_strips_legs_can_be_switched_off():
    r, lines = run(["--gpus", "2", "--dry-launch", "--strips-legs", "off"], {"M2V_DIST_BACKEND": "gloo"})
    assert r.returncode == 0 and len(lines) == 1 and "strips" not in lines[0]


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
    # the driver's form: every rank started ONE child per strips leg on a rendezvous port of the leg's own
    st = lines[0]["strips"]
    assert st["rccl"]["dry_launch"] is True and st["rccl"]["ranks_seen"] == 2 and st["peer"]["ranks_seen"] == 2 and st["peer"]["transport"] == "peer"


def test_under_torch_distributed_run_a_hanging_leg_is_ended_by_every_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    env.update({"M2V_DIST_BACKEND": "gloo", "M2V_BENCH_TEST_HANG_LEG": "rccl", "M2V_BENCH_LEG_TIMEOUT": "25"})
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29633", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       capture_output=True, text=True, env=env, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["ranks_counted"] == 2
    assert lines[0]["strips"]["rccl"]["error"] == "timeout" and lines[0]["strips"]["peer"].get("dry_launch") is True


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
    assert bench.gpu_sensors(None) is None and bench.gpu_sensors("/nonexistent/card") is None      # no such card: nothing invented


def test_sensors_come_from_the_card_at_the_hip_devices_pci_address(tmp_path):
    """A lease that shows HIP one GPU of a node's eight still lists eight cards in sysfs: the card is picked by the PCI address
    m2v_device_pci_bus_id reports, never by its position (round 5 read card 0 - somebody else's idle GPU: 111 MHz, 249.0 W flat)."""
    sys.path.insert(0, ROOT)
    import bench_util as bu
    pci = tmp_path / "devices"
    drm = tmp_path / "drm"
    drm.mkdir()
    addrs = ["0000:05:00.0", "0000:c1:00.0", "0000:e5:00.0"]
    for k, a in enumerate(addrs):
        d = pci / a
        (d / "hwmon" / "hwmon3").mkdir(parents=True)
        busy = a == "0000:c1:00.0"
        (d / "pp_dpm_sclk").write_text("0: 132Mhz %s\n1: 2400Mhz %s\n" % ("" if busy else "*", "*" if busy else ""))
        (d / "pp_dpm_mclk").write_text("0: 900Mhz *\n")
        (d / "hwmon" / "hwmon3" / "power1_average").write_text("%d\n" % (910000000 if busy else 249000000))
        (d / "hwmon" / "hwmon3" / "temp1_input").write_text("61000\n")
        (drm / ("card%d" % k)).mkdir()
        os.symlink(str(d), str(drm / ("card%d" % k) / "device"))
        (drm / ("card%d-DP-1" % k)).mkdir()                  # a connector: not a card
    card = bu.sysfs_card_of("0000:C1:00.0", drm_root=str(drm))
    assert card is not None and os.path.realpath(card).endswith("0000:c1:00.0")
    s = bu.gpu_sensors(card)
    assert s == {"sclk_mhz": 2400, "mclk_mhz": 900, "power_w": 910.0, "temp_c": 61.0}
    assert bu.sysfs_card_of("0000:aa:00.0", drm_root=str(drm)) is None and bu.sysfs_card_of(None, drm_root=str(drm)) is None
    idle = bu.gpu_sensors(bu.sysfs_card_of(addrs[0], drm_root=str(drm)))
    assert idle["sclk_mhz"] == 132
    # the verdict on samples taken while the loop ran
    assert bu.sensors_verdict([s] * 3 + [dict(s, power_w=905.0)] * 7) == (True, None)
    ok, why = bu.sensors_verdict([idle] * 20)
    assert not ok and "idle" in why
    ok, why = bu.sensors_verdict([dict(s)] * 20)
    assert not ok and "constant" in why
    assert bu.sensors_verdict([]) == (False, "no sensor file readable")


def test_queue_placement_decision_table():
    """fake probe timings in, submission form out (bench_util.settle_queue_placement): overlap at once; one stream shared, repaired by
    the second new stream; never repaired by new streams, then by a stream of another priority; nothing helps -> blocking calls;
    repair not allowed (--split given)."""
    sys.path.insert(0, ROOT)
    import bench_util as bu
    G = bu.PLACEMENT_OVERLAP_GAIN

    def settle(fly_times, t_block, t_one, allowed=True):
        it, acts = iter(fly_times), []
        sub, rec = bu.settle_queue_placement(lambda: next(it), lambda: t_block, lambda: t_one, acts.append, repair_allowed=allowed)
        return sub, rec, acts
    sub, rec, acts = settle([0.90], 0.97, 1.00)
    assert sub == "in_flight" and acts == [] and rec["new_streams"] == 0 and rec["priority"] == 0
    sub, rec, acts = settle([1.00, 0.99, 0.91], 0.97, 1.00)
    assert sub == "in_flight" and acts == ["new_stream", "new_stream"] and rec["new_streams"] == 2 and len(rec["probe_ms_per_step"]["in_flight"]) == 3
    sub, rec, acts = settle([1.00, 1.00, 1.00, 1.00, 0.93], 0.97, 1.00)
    assert sub == "in_flight" and acts == ["new_stream"] * bu.PLACEMENT_MAX_NEW_STREAMS + ["priority"] and rec["priority"] == 1
    sub, rec, acts = settle([1.00] * 5, 0.97, 1.00)
    assert sub == "blocking" and len(acts) == bu.PLACEMENT_MAX_NEW_STREAMS + 1
    sub, rec, acts = settle([1.00], 0.97, 1.00, allowed=False)
    assert sub == "blocking" and acts == []
    # a power-capped GPU: no form overlaps by 4 %, but in flight beats the blocking form as it is - left alone (a re-streaming cannot be undone)
    sub, rec, acts = settle([0.993, 1.029], 1.021, 1.024)
    assert sub == "in_flight" and acts == [] and rec["probe_ms_per_step"]["in_flight"] == [993.0]
    sub, rec, acts = settle([1.010, 0.95], 1.021, 1.024)           # ... within 1.5 % of it: repaired as before
    assert acts == ["new_stream"] and sub == "in_flight"
    # the threshold itself: exactly at the gain counts as overlapping, a hair above does not
    assert bu.placement_next_action(G * 1.0, 1.0, 0, False) == "keep" and bu.placement_next_action(G * 1.0 + 1e-9, 1.0, 0, False) == "new_stream"
    assert bu.placement_next_action(1.0, 1.0, bu.PLACEMENT_MAX_NEW_STREAMS, False) == "priority"
    assert bu.placement_next_action(1.0, 1.0, bu.PLACEMENT_MAX_NEW_STREAMS, True) == "keep"


def test_a_strips_legs_line_becomes_an_entry_and_a_failure_a_stated_error():
    """bench_launch.summarize_leg / error_text: rank 0's stdout of a strips leg (noise lines of the backend in front of the JSON line allowed) -> the
    entry of `strips`; no line, a non-zero exit code or a timeout -> {"error": ...} quoting the most telling line of rank 0's stderr"""
    sys.path.insert(0, ROOT)
    import bench_launch as bl
    line = {"value": 1234.5, "unit": "MPixels/s", "ms_per_step": 0.25, "n_gpus": 8, "scaling": "strong", "sequences_in_flight": 3, "steps": 40,
            "in_flight_form": "one thread", "one_sequence_at_a_time": {"value": 1000.0, "ms_per_step": 0.31}, "in_flight_output_rank_0": {"value": 1200.0},
            "in_flight_output_rank_rotating": {"value": 1234.5}, "in_flight_long_sequence": {"gops": 40, "ms_per_90_frames": 0.24},
            "config": {"transport": "peer+rccl", "transport_asked_for": "peer", "strip_loop": "native", "strip_loop_why": None, "gop_steps_ran_as": "peer",
                       "peer": [{"peer_sequences": 50, "giveups": 0, "fell_back": False}, {"peer_sequences": 49, "giveups": 1, "fell_back": True}]},
            "exchange_ms_per_step": {"halo_exposed": 0.0, "halo_total": 0.0, "gather_and_assembly": 0.05}, "roofline": {"frac": 0.11},
            "parity_check": {"identical_to_oracle": True}, "per_rank_ms_per_step": {"columns": ["k_mb_P"], "ranks": [[0.2]] * 8}}
    out = [b"[Gloo] Rank 0 is connected to 7 peer ranks\n", (json.dumps(line) + "\n").encode()]
    e = bl.summarize_leg(out, 0, False, 33.3)
    assert e["value"] == 1234.5 and e["ms_per_sequence"] == 0.25 and e["n_gpus"] == 8 and e["identical_to_oracle"] is True and e["seconds"] == 33.3
    assert e["transport"] == "peer+rccl" and e["peer_sequences"] == 99 and e["giveups"] == 1 and e["fell_back"] is True
    assert e["in_flight_long_sequence"]["gops"] == 40 and e["roofline_frac"] == 0.11 and "error" not in e
    stderr = [b"/opt/amdgpu/share/libdrm/amdgpu.ids: No such file or directory\n", b"Traceback (most recent call last):\n",
              b"bench.py --mode strips: rank 3 failed: m2v_strip_encode_end failed (-5): out of memory\n", b"  state: {...}\n"]
    said = bl.error_text(stderr)
    assert "rank 3 failed" in said
    assert bl.summarize_leg(out, 1, False, 5.0, said) == {"error": "exit code 1", "seconds": 5.0, "rank0_said": said}
    assert bl.summarize_leg([], 0, False, 5.0)["error"] == "no line from rank 0"
    t = bl.summarize_leg(out, 124, True, 180.2, said)
    assert t["error"] == "timeout" and t["rank0_said"] == said and t["bound_s"] == bl.LEG_TIMEOUT_S
    assert bl.error_text([]) is None and bl.error_text([b"all fine\n"]) == "all fine"

