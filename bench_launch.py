"""bench_launch.py — how `bench.py --gpus N` becomes an N-rank job, and how its ONE line also carries config c5.

* launch_ranks(): `python bench.py --gpus N` with no launcher around it starts the N ranks itself.  That process has not imported torch and
  has made no HIP call (a process that initialised the GPU must never exec GPU work); it starts N children of bench.py with RANK /
  LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays or keeps rank 0's stdout, and tears the job down when a rank dies or a deadline passes.
* The strips legs: the driver's multi-GPU command is `bench.py --gpus N` in its default mode - config c4, independent sequences.  Config
  c5 (one frame sharded by macroblock rows, the north star's one collective config) is `--mode strips`, which the driver never passes.
  So after the c4 measurement is in hand, and before its line is printed, every N > 1 job also runs `--mode strips` twice - halo through
  RCCL, then `--transport peer` - each as a batch of N FRESH child processes under a hard wall-clock bound, and attaches what they report
  as `strips: {rccl: {...}, peer: {...}}` to the c4 line, which is printed ONCE, LAST.  A strips leg that hangs, crashes or mismatches
  costs its own entry (`{"error": ...}`), never the c4 line and never the exit code.
    - bench.py started alone (`--gpus N`, WORLD_SIZE unset): the launcher runs the c4 batch, then the two strips batches, itself.
    - bench.py started per rank by torch.distributed.run (the driver's form): every rank, its c4 work done and its encoder handles
      closed, starts ONE child for each leg (a child process, never an exec: the rank has touched the GPU) with its own RANK on a fresh
      rendezvous port that rank 0 picked, waits for it under the same bound, ends it by its exact PID when the bound passes.
"""
import json
import os
import signal
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(ROOT, "bench.py")

LEG_TIMEOUT_S = float(os.environ.get("M2V_BENCH_LEG_TIMEOUT", "180"))       # wall clock per strips leg, start of the children to their exit
LEGS = (("rccl", ["--transport", "rccl"]), ("peer", ["--transport", "peer"]))


def free_port():
    with socket.socket() as sk:                          # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _end(procs, grace_kill):
    """SIGTERM, then - a rank blocked in an RCCL collective or a driver call may ignore that - SIGKILL: exact PIDs, never a pattern"""
    for pr in procs:
        if pr.poll() is None:
            pr.terminate()
    t_kill = time.time() + grace_kill
    while time.time() < t_kill and any(pr.poll() is None for pr in procs):
        time.sleep(0.05)
    for pr in procs:
        if pr.poll() is None:
            pr.kill()
    for pr in procs:
        try:
            pr.wait(timeout=10)
        except subprocess.TimeoutExpired:
            pass


def _tee_stderr(pipe, keep, limit=4096):
    """a child's stderr passed through to ours as it comes, the last few KB kept (what an error entry of `strips` quotes)"""
    for raw in pipe:
        try:
            sys.stderr.buffer.write(raw)
            sys.stderr.buffer.flush()
        except Exception:  # noqa: BLE001
            pass
        keep.append(raw)
        while sum(len(x) for x in keep) > limit and len(keep) > 1:
            keep.pop(0)


def error_text(keep):
    """the most telling line of a rank's stderr: the last one that names a failure, else the last one at all"""
    lines = b"".join(keep).decode(errors="replace").splitlines()
    for ln in reversed(lines):
        if any(w in ln for w in ("failed", "Error", "error", "rror:")) and "amdgpu.ids" not in ln:
            return ln.strip()[-400:]
    return lines[-1].strip()[-400:] if lines else None


def launch_ranks(nranks, argv, relay=True, deadline_s=None, extra_env=None, keep_stderr=None):
    """Starts `bench.py argv` as `nranks` ranks and waits for them.  -> (worst exit code, rank 0's stdout lines as bytes, timed_out).
    relay: rank 0's stdout is passed through as it arrives (the ONE JSON line); otherwise it is only kept.  The other ranks' stdout goes
    to stderr.  A rank that dies takes the job down: the survivors get M2V_BENCH_GRACE seconds to finish on their own, then SIGTERM,
    then SIGKILL.  deadline_s: the same clean-up when the whole batch is still running after that long (timed_out = True).  The
    clean-up also runs when the launcher itself is interrupted or terminated, so no rank is left behind holding a GPU."""
    port = free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), M2V_BENCH_LAUNCHED_BY="bench.py")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, BENCH] + argv, env=env, cwd=ROOT, stdout=subprocess.PIPE if r == 0 else sys.stderr,
                                      stderr=subprocess.PIPE if (r == 0 and keep_stderr is not None) else None))
    if keep_stderr is not None:
        threading.Thread(target=_tee_stderr, args=(procs[0].stderr, keep_stderr), daemon=True).start()
    line0 = procs[0].stdout
    worst, live = 0, set(range(nranks))
    kept = []

    def reader():
        for raw in line0:
            kept.append(raw)
            if relay:
                sys.stdout.buffer.write(raw)
                sys.stdout.buffer.flush()
    t = threading.Thread(target=reader, daemon=True)
    t.start()

    def on_term(signum, frame):
        raise KeyboardInterrupt
    in_main = threading.current_thread() is threading.main_thread()
    old_term = signal.signal(signal.SIGTERM, on_term) if in_main else None
    grace, grace_kill = (float(x) for x in os.environ.get("M2V_BENCH_GRACE", "20,10").split(","))
    t_start = time.time()
    deadline, stage, timed_out = None, 0, False          # stage 0: waiting, 1: SIGTERM sent, 2: SIGKILL sent
    try:
        while live:
            for r in list(live):
                rc = procs[r].poll()
                if rc is None:
                    continue
                live.discard(r)
                if rc != 0:
                    worst = worst or rc
                    if deadline is None:
                        sys.stderr.write("bench.py: rank %d exited with code %d; stopping the other ranks in %.0f s\n" % (r, rc, grace))
                        deadline = time.time() + grace
            if live and deadline_s is not None and not timed_out and time.time() - t_start > deadline_s:
                sys.stderr.write("bench.py: the ranks are still running after %.0f s (the bound of this leg): ending them\n" % deadline_s)
                timed_out, deadline = True, time.time() - 1.0
                worst = worst or 124
            if deadline is not None and time.time() > deadline and stage < 2:
                for r in live:
                    (procs[r].terminate if stage == 0 else procs[r].kill)()
                stage += 1
                deadline = time.time() + grace_kill
            time.sleep(0.05)
    except KeyboardInterrupt:
        worst = worst or 130
        _end(procs, grace_kill)
    finally:
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
    t.join(timeout=10.0)
    return worst, kept, timed_out


# ---------------------------------------------------------------------------------------------------------------------
# the strips legs
# ---------------------------------------------------------------------------------------------------------------------
def leg_argv(args, transport_flags):
    """the command line of one strips leg: the job's size and warm-up, the leg's own step count, two sequences in flight from one thread
    with the output rank rotating; the whole-stream oracle check stays on"""
    a = ["--gpus", str(args.gpus), "--mode", "strips", "--steps", str(args.strips_steps), "--warmup", str(min(args.warmup, 10)),
         "--gops", str(args.gops), "--prewarm", str(min(args.prewarm, 0.5)), "--rotate-dst", "--strips-legs", "off",
         "--long-gops", str(40 if args.long_gops < 0 else args.long_gops)] + transport_flags
    if args.dry_launch:
        a.append("--dry-launch")
    return a


def summarize_leg(lines, rc, timed_out, seconds, stderr_tail=None):
    """rank 0's stdout of a strips leg -> the entry of `strips` in the c4 line: the numbers, or a stated error"""
    line = None
    for raw in reversed(lines):
        try:
            txt = raw.decode() if isinstance(raw, bytes) else raw
            if txt.lstrip().startswith("{"):
                line = json.loads(txt)
                break
        except (ValueError, UnicodeDecodeError):
            continue
    if timed_out:
        d = {"error": "timeout", "bound_s": LEG_TIMEOUT_S, "seconds": round(seconds, 1)}
        if stderr_tail:
            d["rank0_said"] = stderr_tail
        return d
    if rc != 0 or line is None:
        d = {"error": "exit code %d" % rc if rc != 0 else "no line from rank 0", "seconds": round(seconds, 1)}
        if stderr_tail:
            d["rank0_said"] = stderr_tail
        return d
    if line.get("dry_launch"):
        return {"dry_launch": True, "ranks_seen": line.get("ranks_seen"), "mode": line.get("mode"), "transport": line.get("transport"),
                "seconds": round(seconds, 1)}
    cfg = line.get("config", {})
    ex = line.get("exchange_ms_per_step", {})
    peer = cfg.get("peer")
    return {"value": line.get("value"), "unit": line.get("unit"), "ms_per_sequence": line.get("ms_per_step"), "n_gpus": line.get("n_gpus"),
            "scaling": line.get("scaling"), "sequences_in_flight": line.get("sequences_in_flight"), "in_flight_form": line.get("in_flight_form"),
            "one_sequence_at_a_time": line.get("one_sequence_at_a_time"), "in_flight_output_rank_0": line.get("in_flight_output_rank_0"),
            "in_flight_output_rank_rotating": line.get("in_flight_output_rank_rotating"),
            "in_flight_long_sequence": line.get("in_flight_long_sequence"),
            "transport": cfg.get("transport"), "transport_asked_for": cfg.get("transport_asked_for"), "strip_loop": cfg.get("strip_loop"),
            "strip_loop_why": cfg.get("strip_loop_why"), "gop_steps_ran_as": cfg.get("gop_steps_ran_as"),
            "peer_sequences": sum(p["peer_sequences"] for p in peer) if peer else None,
            "giveups": sum(p["giveups"] for p in peer) if peer else None,
            "fell_back": any(p["fell_back"] for p in peer) if peer else None,
            "per_rank_ms_per_step": line.get("per_rank_ms_per_step"), "kernel_ms_per_step": line.get("kernel_ms_per_step"),
            "halo_exposed_ms": ex.get("halo_exposed"), "halo_total_ms": ex.get("halo_total"), "gather_and_assembly_ms": ex.get("gather_and_assembly"),
            "roofline_frac": (line.get("roofline") or {}).get("frac"),
            "identical_to_oracle": (line.get("parity_check") or {}).get("identical_to_oracle"),
            "steps": line.get("steps"), "seconds": round(seconds, 1)}


def legs_from_launcher(args):
    """bench.py is the launcher (it has made no HIP call): each leg is a batch of N fresh ranks under the bound"""
    out = {}
    for name, flags in LEGS:
        t0 = time.time()
        try:
            err_keep = []
            rc, lines, timed_out = launch_ranks(args.gpus, leg_argv(args, flags), relay=False, deadline_s=LEG_TIMEOUT_S,
                                                extra_env={"M2V_BENCH_GRACE": "5,5", "M2V_BENCH_LEG": name}, keep_stderr=err_keep)
            out[name] = summarize_leg(lines, rc, timed_out, time.time() - t0, error_text(err_keep))
        except Exception as ex:  # noqa: BLE001   (a leg never costs the c4 line)
            out[name] = {"error": "launcher: %r" % (ex,), "seconds": round(time.time() - t0, 1)}
    return out


def legs_from_rank(args, rank, world, ports):
    """bench.py is rank `rank` of a job somebody else started (torch.distributed.run): ONE child per leg, same RANK / LOCAL_RANK /
    WORLD_SIZE, the rendezvous port of the leg (ports: picked by rank 0, the same on every rank).  Every rank enforces the bound on its
    own child.  -> {leg: entry} on rank 0, {} elsewhere."""
    out = {}
    for (name, flags), port in zip(LEGS, ports):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), M2V_BENCH_LAUNCHED_BY="bench.py (a rank's child)", M2V_BENCH_LEG=name)
        for k in [k for k in env if k.startswith("TORCHELASTIC_") or k in ("GROUP_RANK", "ROLE_RANK", "ROLE_NAME", "GROUP_WORLD_SIZE", "ROLE_WORLD_SIZE")]:
            env.pop(k)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        t0 = time.time()
        try:
            pr = subprocess.Popen([sys.executable, BENCH] + leg_argv(args, flags), env=env, cwd=ROOT,
                                  stdout=subprocess.PIPE if rank == 0 else sys.stderr, stderr=subprocess.PIPE if rank == 0 else None)
            lines, err_keep = [], []
            rd = None
            if rank == 0:
                rd = threading.Thread(target=lambda: lines.extend(pr.stdout), daemon=True)
                rd.start()
                threading.Thread(target=_tee_stderr, args=(pr.stderr, err_keep), daemon=True).start()
            timed_out = False
            try:
                pr.wait(timeout=LEG_TIMEOUT_S)
            except subprocess.TimeoutExpired:
                timed_out = True
                _end([pr], 5.0)
            if rd is not None:
                rd.join(timeout=10.0)
            if rank == 0:
                out[name] = summarize_leg(lines, pr.returncode if pr.returncode is not None else -9, timed_out, time.time() - t0, error_text(err_keep))
        except Exception as ex:  # noqa: BLE001
            if rank == 0:
                out[name] = {"error": "rank 0: %r" % (ex,), "seconds": round(time.time() - t0, 1)}
    return out


def legs_wanted(args, world):
    """the strips legs ride on the DEFAULT job only: N > 1, sequences mode, the metric's configuration"""
    if args.strips_legs == "off" or os.environ.get("M2V_BENCH_STRIPS_LEGS") == "0":
        return False
    return world > 1 and args.mode == "sequences" and args.config == "c3" and not args.ablate


def attach_and_print(line_bytes, strips):
    """the c4 line, once, last, with `strips` added (if the line cannot be parsed it goes out as it came)"""
    for raw in line_bytes:
        txt = raw.decode(errors="replace")
        if txt.lstrip().startswith("{"):
            try:
                d = json.loads(txt)
                d["strips"] = strips
                txt = json.dumps(d) + "\n"
            except ValueError:
                pass
        sys.stdout.write(txt)
    sys.stdout.flush()


# ---------------------------------------------------------------------------------------------------------------------
# --dry-launch: the rendezvous alone
# ---------------------------------------------------------------------------------------------------------------------
def dry_launch(args, rank, world):
    """--dry-launch: the rendezvous alone, no encoder - runs without a GPU (gloo), which is how tests/ checks on CPU that
    `bench.py --gpus N` really is an N-rank job.  Every rank contributes 1 to an all-reduce; rank 0 prints the line."""
    import torch
    import torch.distributed as dist
    backend = os.environ.get("M2V_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if os.environ.get("M2V_BENCH_TEST_FAIL_RANK") == str(rank):        # tests/test_bench_launch.py: a rank that dies early
        return 3
    if os.environ.get("M2V_BENCH_TEST_DEAF_RANK") == str(rank):        # ... and one that is stuck and ignores SIGTERM
        signal.signal(signal.SIGTERM, signal.SIG_IGN)
        with open(os.environ["M2V_BENCH_TEST_PIDFILE"], "w") as f:
            f.write(str(os.getpid()))
        time.sleep(600)
        return 0
    if os.environ.get("M2V_BENCH_TEST_HANG_LEG") and os.environ.get("M2V_BENCH_TEST_HANG_LEG") == os.environ.get("M2V_BENCH_LEG"):
        time.sleep(600)                                                 # ... and a strips leg that never comes back
        return 0
    if os.environ.get("M2V_BENCH_TEST_FAIL_LEG") and os.environ.get("M2V_BENCH_TEST_FAIL_LEG") == os.environ.get("M2V_BENCH_LEG"):
        return 5
    seen, total = 1, 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, rank=rank, world_size=world)
        t = torch.ones(1, dtype=torch.int64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t)
        seen, total = dist.get_world_size(), int(t.item())
        strips = None
        if legs_wanted(args, world) and os.environ.get("M2V_BENCH_LAUNCHED_BY", "caller") == "caller":
            # the driver's form of the job, rehearsed without a GPU: every rank starts its own child per leg
            ports = torch.tensor([free_port() for _ in LEGS] if rank == 0 else [0] * len(LEGS), dtype=torch.int64,
                                 device="cuda" if backend == "nccl" else "cpu")
            dist.broadcast(ports, src=0)
            strips = legs_from_rank(args, rank, world, [int(p) for p in ports.tolist()])
        dist.barrier()
        dist.destroy_process_group()
    else:
        strips = None
    if rank == 0:
        d = {"dry_launch": True, "n_gpus": seen, "ranks_seen": seen, "ranks_counted": total, "gpus_arg": args.gpus,
             "backend": backend if world > 1 else None, "mode": args.mode, "transport": args.transport,
             "launched_by": os.environ.get("M2V_BENCH_LAUNCHED_BY", "caller")}
        if strips is not None:
            d["strips"] = strips
        print(json.dumps(d))
        sys.stdout.flush()
    return 0 if total == world == seen else 1
