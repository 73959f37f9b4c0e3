"""bench_strips.py — `bench.py --mode strips`: BASELINE config c5, ONE 2048x2048 (XL=YL=7) I+P sequence cut into `world` strips of
macroblock rows; the +-6 luma / +-3 chroma reference rows cross xGMI once per GOP step (RTL:1446-1448; slices independent RTL:2704-2715;
closed GOPs RTL:2656).  Strong scaling: the total work is fixed.

What is timed, K steps each, barrier + synchronize + max over ranks around every leg:
  * one sequence at a time: one blocking m2v_strip_encode after the other (what rounds 3-5 reported);
  * sequences in flight (the default, --strip-inflight 2): that many handles taking turns from ONE thread through
    m2v_strip_encode_begin / _end - a peer communicator (landing block) per handle with --transport peer - over ONE base communicator;
    with --rotate-dst sequence i is assembled on rank i mod world, so no single rank carries every gather.  This is `value`.
  * --strip-threads K (opt-in, the round-5 form): K host threads, a handle and a communicator stack each.
The line carries what the sequences mode carries (roofline of this rank's P-frame launches, CPU baseline, whole-stream check against
the oracle on rank 0) plus the halo / gather time per step and every rank's kernel time."""
import json
import os
import sys
import time

from bench_util import FPGA_MPIXELS, HBM_PEAK_GBS, compare_with_per_gop_oracle, rtl_sim_probe


def bench_strips(args, cfg, M, torch, dist, rank, local_rank, world, dev):
    Ws = Hs = 2048
    PFRAMES, VL, Q = cfg.PFRAMES, cfg.VL, cfg.Q
    gop = PFRAMES + 1
    nframes = args.gops * gop
    backend = dist.get_backend() if dist is not None else None
    clip = M.synth.clip_torch(Ws, Hs, nframes, clip_index=0, device=dev)        # every rank holds the same clip
    inflight = args.strip_inflight if args.strip_inflight > 0 else (3 if args.transport == "peer" else 2)
    K = max(1, args.strip_threads if args.strip_threads > 1 else inflight)
    threads_form = args.strip_threads > 1
    # The loop: native (m2v_strip_encode*: the GOP steps and the exchange issued from C++) whenever the ranks have a communicator the
    # library can drive - RCCL between GPUs; for one rank nothing; with the 1-GPU test hook (gloo, shared device) and --transport peer a
    # communicator over torch.distributed (parallel.dist_comm) under the peer transport, whose landing blocks then cross the PROCESS
    # boundary through hipIpc handles.  parallel.encode_strips (the Python statement of the same call order, point-to-point ops through
    # torch.distributed) is what the hook runs otherwise, and the agreed fallback should librccl refuse to initialise.
    loop, why = "native", None
    over_dist = world > 1 and backend != "nccl" and args.transport == "peer" and os.environ.get("M2V_STRIP_LOOP") != "python"
    if os.environ.get("M2V_STRIP_LOOP") == "python" or (world > 1 and backend != "nccl" and not over_dist):
        loop, why = "python", "M2V_STRIP_LOOP=python" if os.environ.get("M2V_STRIP_LOOP") == "python" else "backend %s" % backend
        K = 1
    encs = [M.Mpeg2Encoder(7, 7, VL, Q, device=local_rank) for _ in range(K)]
    enc = encs[0]
    # communicators: `bases` (threads form: one per stack; else ONE), `comms[k]` = what handle k exchanges through (a peer communicator
    # of its own on top of its base, or the base itself)
    bases, comms, peer_on = [], [None] * K, False
    if loop == "native" and world > 1:
        # (StripComm.rccl is collective: a failure to make the id on rank 0 reaches every rank through its broadcast and all of them
        # land in the except branch together; a rank stuck in ncclCommInitRank because another one never arrived ends its process
        # after init_timeout, and the launcher tears the job down.  The vote below runs after every rank has returned from it.)
        ok = 1
        try:
            for _ in range(K if threads_form else 1):
                if over_dist:
                    bases.append(M.parallel.dist_comm(M.StripComm, dist, world))
                else:
                    bases.append(M.StripComm.rccl(rank, world, local_rank, dist, init_timeout=float(os.environ.get("M2V_RCCL_INIT_TIMEOUT", "180"))))
        except Exception as ex:  # noqa: BLE001
            ok, why = 0, "communicator: %s" % ex
        t = torch.tensor([ok], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            loop, why = "python", why or "another rank could not initialise its communicator"
            for b in bases:
                b.close()
            bases, K = [], 1
            for e in encs[1:]:
                e.close()
            encs = encs[:1]
            comms = [None]
        else:
            comms = [bases[k if threads_form else 0] for k in range(K)]
            if args.transport == "peer":
                # the peer transport on top (the base keeps moving sizes and strips, and the halo if a wait ever runs out of budget);
                # creating it is collective (the landing blocks' IPC handles are all-gathered through the base).  A rank that cannot -
                # the vote again - leaves every rank on the plain base.
                ok, made, peer_why = 1, [], None
                try:
                    for k in range(K):
                        # (a landing buffer holds one GOP step's rows of ALL GOPs: sized for the longest sequence this run encodes)
                        made.append(M.StripComm.peer(comms[k], rank, local_rank, halo_bytes=max(args.gops, args.long_gops) * 3 * VL * Ws + 4096))
                except Exception as ex:  # noqa: BLE001
                    ok, peer_why = 0, "m2v_comm_init_peer: %s" % ex
                t = torch.tensor([ok], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
                dist.all_reduce(t, op=dist.ReduceOp.MIN)
                if int(t.item()) == 1:
                    comms, peer_on = made, True
                else:
                    why = peer_why if not ok else "another rank could not set the peer transport up"
                    for pc in made:
                        pc.close()
    comm = comms[0]
    torch.cuda.synchronize()
    out = None
    rotate = bool(args.rotate_dst) and loop == "native" and world > 1
    cap = M.parallel.strip_output_bound(nframes, Ws, Hs)
    if loop == "native":
        # an output buffer per handle on rank 0 - on every rank when the output rank rotates
        d_outs = [torch.empty(cap, dtype=torch.uint8, device=dev) if (rank == 0 or rotate) else None for _ in range(K)]

        def step(timings=None, k=0):
            return M.parallel.encode_strips_native(encs[k], comms[k], rank, world, clip, 128, 128, PFRAMES, d_outs[k])
    else:
        eng = M.parallel.GpuStripEngine(enc, clip, 128, 128, PFRAMES, dev)
        d_outs = [None]

        def step(timings=None, k=0):
            return M.parallel.encode_strips(eng, rank, world, dist, timings=timings)

    def run_turns(steps, rot):
        """exactly `steps` sequences, K of them under way at any time, from this one thread: sequence i goes to handle i mod K as soon as
        that handle's previous sequence has been collected; rot: sequence i is assembled on rank i mod world"""
        busy, last = [None] * K, None
        for i in range(steps):
            h = i % K
            if busy[h] is not None:
                r = M.parallel.encode_strips_native_end(encs[h], d_outs[h], rank, busy[h])
                last = r if r is not None else last
            dst = i % world if rot else 0
            M.parallel.encode_strips_native_begin(encs[h], comms[h], rank, world, clip, 128, 128, PFRAMES, d_outs[h], dst=dst)
            busy[h] = dst
        for k in range(K):
            h = (steps + k) % K
            if busy[h] is not None:
                r = M.parallel.encode_strips_native_end(encs[h], d_outs[h], rank, busy[h])
                last = r if r is not None else last
                busy[h] = None
        return last

    def describe():
        """what this rank was running, for the failure path of ANY rank (stderr: rank 0's stdout carries the JSON line only)"""
        d = {"rank": rank, "ranks_seen": dist.get_world_size() if dist is not None else 1, "strip_loop": loop, "strip_loop_why": why,
             "transport": comm.kind if comm is not None else None, "dist_backend": backend, "device": dev, "handles": K}
        if peer_on:
            d["peer"] = [c.peer_stats() for c in comms]
        if loop == "native":
            try:
                d["strip_graph"] = enc.strip_graph_stats()
                d["last_error"] = [e._L.m2v_last_error(e._h).decode() for e in encs]
            except Exception as ex:  # noqa: BLE001
                d["strip_graph"] = "unreadable: %s" % ex
        return d

    def guarded(fn, *a):
        try:
            return fn(*a)
        except BaseException as ex:
            sys.stderr.write("bench.py --mode strips: rank %d failed: %s\n  state: %s\n" % (rank, ex, json.dumps(describe())))
            sys.stderr.flush()
            raise

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, *a):
        barrier()
        t0 = time.perf_counter()
        res = guarded(fn, *a)
        barrier()
        return time.perf_counter() - t0, res

    for _ in range((20 if args.prewarm > 0 else 0) + args.warmup):     # fixed count: every rank takes part in the halo exchange
        out = guarded(step)
    # ---- one sequence at a time ----
    dt, out = timed(lambda: [step() for _ in range(args.steps)][-1])
    # ---- sequences in flight from one thread (turns), the output rank fixed and - if asked for - rotating ----
    dt_turns = dt_rot = dt_thr = None
    turns_identical = None
    if loop == "native" and K > 1 and not threads_form:
        for e in encs:
            e.set_option("split_streams", 1)  # in flight the sequences themselves are what overlaps: one stream each (tools/strip_solo_turns.py)
        guarded(run_turns, max(2 * K, args.warmup), False)
        dt_turns, last = timed(run_turns, args.steps, False)
        if rank == 0 and out is not None and last is not None:
            turns_identical = bool(torch.equal(last, out)) and all(torch.equal(d_outs[k][:out.numel()], out) for k in range(min(K, args.steps)))
        if rotate:
            guarded(run_turns, max(2 * K, 2 * world), True)
            dt_rot, _ = timed(run_turns, args.steps, True)
        for e in encs:
            e.set_option("split_streams", cfg.LIB_DEFAULT_SPLIT_STREAMS)
    # ---- the same with K host threads (opt-in): thread k runs steps k, k + K, ... on stack k (the same split on every rank) ----
    if loop == "native" and threads_form:
        import threading
        errs = []

        def worker(k, count):
            try:
                for _ in range(count):
                    step(None, k)
            except BaseException as ex:  # noqa: BLE001
                errs.append((k, ex))
        for phase in ("warm", "timed"):
            counts = [len(range(k, args.steps if phase == "timed" else max(4 * K, args.warmup), K)) for k in range(K)]
            th = [threading.Thread(target=worker, args=(k, counts[k])) for k in range(K)]
            barrier()
            t0 = time.perf_counter()
            for x in th:
                x.start()
            for x in th:
                x.join()
            barrier()
            dt_thr = time.perf_counter() - t0
            if errs:
                sys.stderr.write("bench.py --mode strips: rank %d, sequences in flight: %r\n  state: %s\n" % (rank, errs, json.dumps(describe())))
                raise errs[0][1]
    graph_stats = enc.strip_graph_stats() if loop == "native" else None      # the timed steps: one recorded hipGraph launch each?
    host_us_timed = enc.strip_stats().get("host_us_per_step") if loop == "native" else None
    # one more pass with per-launch HIP events (option profile) and the exchange bracketed by events on the engine's stream
    enc.set_option("profile", 1)
    timings = {}
    guarded(step, timings)
    out = guarded(step, timings) if loop == "python" else guarded(step)
    if loop == "native":
        timings = enc.strip_stats()
    launches, ms_p, px_p = enc.kernel_stats(0)
    _, ms_i, _ = enc.kernel_stats(1)
    _, ms_asm, _ = enc.kernel_stats(3)
    _, ms_fin, _ = enc.kernel_stats(2)
    _, ms_scan, _ = enc.kernel_stats(4)
    enc.set_option("profile", 0)
    host_us, host_out = timings.get("host_us_per_step"), timings.get("host_us_per_step_outside_comm")
    per_rank = None
    if dist is not None:
        on = dev if backend == "nccl" else "cpu"
        mine = torch.tensor([ms_p, ms_i, ms_scan + ms_asm, timings.get("halo_exposed", 0.0)], dtype=torch.float64, device=on)     # (this rank's, before the max)
        t = torch.tensor([dt, timings.get("halo_exposed", 0.0), timings.get("halo_total", 0.0), timings.get("gather", 0.0), dt_turns or 0.0, dt_rot or 0.0,
                          dt_thr or 0.0], dtype=torch.float64, device=on)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, timings["halo_exposed"], timings["halo_total"], timings["gather"], m_turns, m_rot, m_thr = (float(v) for v in t.tolist())
        dt_turns, dt_rot, dt_thr = (m if x is not None else None for x, m in ((dt_turns, m_turns), (dt_rot, m_rot), (dt_thr, m_thr)))
        every = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[round(float(v), 3) for v in x.tolist()] for x in every]
    # ---- the in-flight loop once more on a LONGER sequence: a GOP step is one launch over the step's frames of ALL GOPs, so G GOPs make every launch
    #      G / gops times as large and the ramp and drain of the nine launches weigh that much less (profiles/r06_experiments.txt item 20).  Last, and on its
    #      own feet: everything else of the line has been measured; a rank that cannot allocate the long clip makes every rank skip (a vote), and a failure
    #      inside the loop - the same call fails on every rank, that is the strip protocol - becomes an entry in the line, not the end of the leg ----
    long_res = long_err = None
    if loop == "native" and K > 1 and not threads_form and args.long_gops > 0 and args.long_gops != args.gops:
        n_long = args.long_gops * gop
        clip_s, outs_s = clip, d_outs
        ok = 1
        try:
            clip = M.synth.clip_torch(Ws, Hs, n_long, clip_index=0, device=dev)
            d_outs = [torch.empty(n_long * Ws * Hs * 3 // 2, dtype=torch.uint8, device=dev) if (rank == 0 or rotate) else None for _ in range(K)]
            torch.cuda.synchronize()
        except Exception as ex:  # noqa: BLE001
            ok, long_err = 0, "rank %d: %s" % (rank, str(ex)[:200])
        if dist is not None:
            t = torch.tensor([ok], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0:
                ok, long_err = 0, long_err or "another rank could not allocate the long sequence"
        if ok:
            try:
                for e in encs:
                    e.set_option("split_streams", 1)
                steps_long = max(2 * K, args.steps * args.gops // args.long_gops)
                run_turns(2 * K, rotate)
                barrier()
                t0 = time.perf_counter()
                run_turns(steps_long, rotate)
                barrier()
                t_long = time.perf_counter() - t0
                if dist is not None:
                    t = torch.tensor([t_long], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    t_long = float(t.item())
                long_res = (t_long, steps_long, n_long, enc.strip_last_form())
            except Exception as ex:  # noqa: BLE001
                long_err = "rank %d: %s" % (rank, str(ex)[:300])
                sys.stderr.write("bench.py --mode strips: the long-sequence loop failed on rank %d: %s\n  state: %s\n" % (rank, ex, json.dumps(describe())))
                for e in encs:                 # sequences begun and never collected: dropped (every rank does the same)
                    try:
                        e.reset()
                    except Exception:  # noqa: BLE001
                        pass
        clip, d_outs = clip_s, outs_s
        torch.cuda.empty_cache()
        for e in encs:
            try:
                e.set_option("split_streams", cfg.LIB_DEFAULT_SPLIT_STREAMS)
            except Exception:  # noqa: BLE001
                pass
    dt_fly = dt_thr if threads_form else (dt_rot if rotate else dt_turns)
    if rank == 0:
        px = nframes * Ws * Hs
        rows = M.parallel.partition_rows(128, world)[0]
        strip_px = (rows[1] - rows[0]) * 16 * Ws
        alg_bytes = args.gops * ((PFRAMES - 1) * 6.0 + 4.5) * strip_px          # this rank's P-frame launches of one step
        achieved = alg_bytes / (ms_p * 1e-3) * 1e-9 if ms_p > 0 else 0.0
        rate = lambda d: round(args.steps * px / d * 1e-6, 2) if d else None      # noqa: E731
        per = lambda d: round(d / args.steps * 1e3, 3) if d else None             # noqa: E731
        line = {
            "metric": "MPixels/s encoded, 2048x2048 I+P, macroblock-row strips", "value": rate(dt_fly or dt),
            "unit": "MPixels/s", "n_gpus": dist.get_world_size() if dist is not None else 1,
            "ranks_seen": dist.get_world_size() if dist is not None else 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per(dt_fly or dt), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": round(args.steps * px / (dt_fly or dt) * 1e-6 / FPGA_MPIXELS, 3), "dtype": "u8", "data": "synthetic",
            # `value`: the K steps with `sequences_in_flight` of them under way on every rank; one sequence at a time - what earlier rounds
            # reported - beside it
            "sequences_in_flight": K if dt_fly is not None else 1,
            "in_flight_form": None if dt_fly is None else ("threads" if threads_form else "one thread, m2v_strip_encode_begin / _end on %d handles%s" %
                                                           (K, ", output rank = sequence mod world" if rotate else ", output rank 0")),
            "one_sequence_at_a_time": {"value": rate(dt), "ms_per_step": per(dt)},
            "in_flight_output_rank_0": {"value": rate(dt_turns), "ms_per_step": per(dt_turns), "identical_to_the_blocking_call": turns_identical} if dt_turns else None,
            "in_flight_output_rank_rotating": {"value": rate(dt_rot), "ms_per_step": per(dt_rot)} if dt_rot else None,
            "in_flight_long_sequence": {"gops": args.long_gops, "frames": long_res[2], "sequences": long_res[1],
                                        "value": round(long_res[1] * long_res[2] * Ws * Hs / long_res[0] * 1e-6, 2),
                                        "ms_per_90_frames": round(long_res[0] / long_res[1] * 1e3 * 90 / long_res[2], 4),
                                        "output_rank": "rotating" if rotate else "rank 0", "gop_steps_ran_as": long_res[3],
                                        "note": "the same loop on a sequence of %d GOPs: a GOP step is one launch over all GOPs' frames" % args.long_gops} if long_res else
                                       ({"error": long_err, "gops": args.long_gops} if long_err else None),
            "config": {"workload": "c5: ONE 2048x2048 yuv444p sequence, %d GOPs of 1 I + %d P, VECTOR_LEVEL=3 Q_LEVEL=2, "
                                   "%d strips of macroblock rows, halo = 9 rows x 2048 B per frame per direction"
                                   % (args.gops, PFRAMES, world), "frames": nframes,
                       "stream_bytes": int(out.numel()) if out is not None else None,
                       "baseline": "FPGA Kintex-7 268 MPixels/s (README.md:22)",
                       "strip_loop": loop, "strip_loop_why": why, "dist_backend": backend,
                       "transport": comm.kind if comm is not None else None,
                       "transport_asked_for": args.transport if world > 1 else None,
                       "peer": [c.peer_stats() for c in comms] if peer_on else None,
                       "gop_steps_ran_as": enc.strip_last_form() if loop == "native" else "python loop",
                       "strip_graph": graph_stats,
                       "launched_by": os.environ.get("M2V_BENCH_LAUNCHED_BY", "caller")},
            "roofline": {"bound": "hbm", "kernel": "k_mb<3,true> on rank 0's strip (%d macroblock rows), P-frame launches of one step" % (rows[1] - rows[0]),
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": None, "launches_per_step": launches, "kernel_ms_per_step": round(ms_p, 3),
                         "algorithmic_bytes_per_step": round(alg_bytes),
                         "timed_in": "extra pass with option profile (HIP events around every launch on the engine's stream)"},
            "exchange_ms_per_step": {"halo_exposed": round(timings.get("halo_exposed", 0.0), 3), "halo_total": round(timings.get("halo_total", 0.0), 3),
                                     "gather_and_assembly": round(timings.get("gather", 0.0), 3),
                                     "host_us_per_gop_step": round(host_us, 1) if host_us is not None else None,
                                     "host_us_per_gop_step_in_the_timed_steps": round(host_us_timed, 1) if host_us_timed is not None else None,
                                     "host_us_per_gop_step_outside_the_communicator": round(host_out, 1) if host_out is not None else None,
                                     "note": "max over ranks; halo_exposed = stream time spent waiting for neighbour rows after the "
                                             "interior rows were done, halo_total = from edge rows packed to neighbour rows there"},
            "kernel_ms_per_step": {"k_mb_P": round(ms_p, 3), "k_mb_I": round(ms_i, 3), "scans": round(ms_scan, 3),
                                   "k_assemble": round(ms_asm, 3), "k_strip_layout + k_strip_assemble": round(ms_fin, 3)},
            "per_rank_ms_per_step": {"columns": ["k_mb_P", "k_mb_I", "scans + k_assemble", "halo_exposed"], "ranks": per_rank} if per_rank else None,
        }
        if not args.no_cpu_baseline:
            from concurrent.futures import ThreadPoolExecutor
            from oracle import m2v_oracle_ctypes as orc
            orc.build()
            clip_np = clip.cpu().numpy()
            t1 = time.perf_counter()
            first = orc.encode(clip_np[:gop], 128, 128, PFRAMES, 7, 7, VL, Q)
            d1 = time.perf_counter() - t1
            line["cpu_baseline"] = dict(value=round(gop * Ws * Hs / d1 * 1e-6, 4), unit="MPixels/s", cores=1, kind="port",
                                        sample="first GOP (%d frames) of the 2048x2048 clip, oracle/m2v_oracle.c, %.1f s" % (gop, d1))
            with ThreadPoolExecutor(args.gops) as ex:
                refs = [first] + list(ex.map(lambda k: orc.encode(clip_np[k * gop:(k + 1) * gop], 128, 128, PFRAMES, 7, 7, VL, Q),
                                             range(1, args.gops)))
            bad = compare_with_per_gop_oracle(out.cpu().numpy().tobytes(), refs, gop)
            line["parity_check"] = {"gops_compared": args.gops, "stream_bytes_compared": int(out.numel()), "identical_to_oracle": not bad,
                                    "problems": bad[:5]}
            line["rtl_sim"] = rtl_sim_probe()
        print(json.dumps(line))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()                               # nobody frees a landing block a neighbour may still be storing into
    for e in encs:
        e.close()
    if peer_on:
        for c in comms:
            c.close()
    for b in bases:
        b.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
