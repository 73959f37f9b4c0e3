#!/usr/bin/env python3
"""bench.py — MPixels/s of the MI355X MPEG-2 I/P encoder on BASELINE.json's config c3.

One "step" = one pass of the hot path over one batch of synthetic input: a 1920x1152 yuv444p clip
of 10 closed GOPs (1 I + 8 P frames each, VECTOR_LEVEL=3, Q_LEVEL=2, XL=YL=7) resident in HBM,
encoded to the final MPEG-2 elementary stream in HBM through the C-ABI.  The timed loop keeps two such
sequences in flight on two encoder handles (m2v_encode_resident_begin / _end): the stream assembly of one
runs beside the first macroblock kernels of the next.  `one_synchronous_call_per_step` in the line is
the same K steps as one blocking m2v_encode_resident call after the other (what rounds 1 and 2 timed).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode sequences|strips] [--config c3|c2]

N > 1: one rank per GPU.  Either the caller starts the ranks (torch.distributed.run: RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_* in the environment), or - WORLD_SIZE unset - this process starts them itself: `python bench.py --gpus 8` alone is an
8-rank job (bench_launch.launch_ranks(): N children before anything touches the GPU, rank 0's JSON line relayed, worst exit code).
Every rank encodes its own clip (BASELINE config c4: independent sequences, no data-path collective) -> "scaling": "weak";
--mode strips is config c5 (ONE sequence, macroblock-row strips, RCCL halo exchange or peer stores) -> "scaling": "strong".
Rank 0 prints ONE JSON line; "n_gpus" is the world size the process group reports ("ranks_seen").

What an unattended `bench.py --gpus N` (N > 1) runs, in order (bench_launch.py): the c4 measurement on N ranks; then, its line in hand
and not yet printed, config c5 twice as N FRESH child processes each under a hard wall-clock bound - halo through RCCL, then the peer
transport - whose results (or a stated error: timeout, exit code) become `strips: {rccl, peer}` of the c4 line; then the line, once.

The legs live in modules of their own: bench_util.py (parity comparison, sensors, provenance of the counter passes, queue placement),
bench_launch.py (ranks, strips legs, --dry-launch), bench_strips.py (config c5), bench_e2e.py (the port contract, host to host)."""
import argparse
import json
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_launch  # noqa: E402  (no torch, no HIP: the launcher imports it before it starts its children)
import bench_util  # noqa: E402
from bench_util import (FPGA_MPIXELS, HBM_PEAK_GBS, compare_with_per_gop_oracle, gop_time_code, gpu_sensors, hbm_copy_rate,  # noqa: E402,F401
                        kernel_source_sha, pmc_traffic, pmc_valu_busy, rtl_sim_probe, sensors_verdict, settle_queue_placement, source_shas,
                        split_gops, sysfs_card_of, visible_gpus)

W, H, XS16, YS16 = 1920, 1152, 120, 72
PFRAMES, GOPS = 8, 10
XL = YL = 7
VL, Q = 3, 2
LIB_DEFAULT_SPLIT_STREAMS = 2  # option split_streams as m2v_create leaves it (csrc/m2v_host.hpp)


def config():
    """the workload's parameters as the legs in other modules take them (main() rewrites the globals for --config c2)"""
    return types.SimpleNamespace(W=W, H=H, XS16=XS16, YS16=YS16, PFRAMES=PFRAMES, GOPS=GOPS, XL=XL, YL=YL, VL=VL, Q=Q,
                                 LIB_DEFAULT_SPLIT_STREAMS=LIB_DEFAULT_SPLIT_STREAMS)


def cpu_baseline(frames_np):
    """The CPU oracle (oracle/, a C restatement of the RTL: kind 'port') timed on the first frames of the same clip, 1 core:
    one GOP of config c3, 96 I frames of config c2."""
    from oracle import m2v_oracle_ctypes as orc
    orc.build()
    n = frames_np.shape[0]
    t0 = time.perf_counter()
    orc.encode(frames_np, XS16, YS16, PFRAMES, XL, YL, VL, Q)
    dt = time.perf_counter() - t0
    what = "first GOP (%d frames, 1 I + %d P)" % (n, n - 1) if PFRAMES else "first %d I frames" % n
    return dict(value=round(n * W * H / dt * 1e-6, 4), unit="MPixels/s", cores=1, kind="port",
                sample="%s of the benchmark clip, oracle/m2v_oracle.c, %.1f s" % (what, dt))


def cpu_baseline_all_cores(clip_np, gpu_stream_bytes):
    """SURVEY.md 8(d) baseline (2), 'all cores, one GOP per thread': closed GOPs are independent, so a CPU encoder scales
    by giving every core its own GOP (ctypes drops the GIL).  Thread t encodes GOP t mod 10 of the benchmark clip as a
    sequence of its own, so the same work also checks the WHOLE GPU stream of the timed workload."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import m2v_oracle_ctypes as orc
    gop = PFRAMES + 1
    ngops = clip_np.shape[0] // gop
    workers = min(os.cpu_count() or 1, 64)
    threads = max(ngops, workers)                 # jobs: every GOP at least once, and at least one per core
    t0 = time.perf_counter()
    with ThreadPoolExecutor(workers) as ex:
        outs = list(ex.map(lambda t: orc.encode(clip_np[(t % ngops) * gop:(t % ngops + 1) * gop], XS16, YS16, PFRAMES, XL, YL, VL, Q),
                           range(threads)))
    dt = time.perf_counter() - t0
    bad = ["oracle not deterministic on GOP %d" % (t % ngops) for t in range(ngops, threads) if outs[t] != outs[t % ngops]]
    bad += compare_with_per_gop_oracle(gpu_stream_bytes, outs[:ngops], gop)
    base = dict(value=round(threads * gop * W * H / dt * 1e-6, 3), unit="MPixels/s", cores=workers, kind="port",
                sample="%d jobs on %d threads, each one GOP (%d frames) of the benchmark clip (GOP t mod %d), %.1f s" % (threads, workers, gop, ngops, dt))
    parity = {"gops_compared": ngops, "stream_bytes_compared": len(gpu_stream_bytes),
              "identical_to_oracle": not bad, "problems": bad[:5],
              "how": "every GOP of the timed clip encoded by the oracle as its own sequence (closed GOPs) and compared byte for "
                     "byte; sequence headers, GOP time codes and the end code / padding checked against RTL:2598-2698, 2932-2937"}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-launch", action="store_true",
                    help="start / join the N ranks, all-reduce a 1 and print what the process group saw; no GPU work")
    ap.add_argument("--config", choices=["c3", "c2"], default="c3",
                    help="c3 (default, the metric's configuration): 1920x1152 I+P; c2: 640x480 I frames only (i_pframes_count = 0)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--gops", type=int, default=GOPS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--prewarm", type=float, default=1.5,
                    help="seconds of untimed encoder steps BEFORE the W warmup steps: a step is ~2 ms, far shorter than the "
                         "GPU's clock ramp out of its idle state (sclk 312 MHz), so a cold start would time the ramp")
    ap.add_argument("--sustain", type=float, default=5.0,
                    help="seconds of the SAME in-flight loop run once more after the K timed steps, in 0.5 s windows (the line's `sustained` "
                         "object: clocks, power cap, thermal state over thousands of sequences instead of K); 0 = skip")
    ap.add_argument("--split", type=int, default=-1,
                    help="option split_streams of the encoder (GOP groups on that many HIP streams); -1 = the library's default (2)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-to-host end_to_end leg")
    ap.add_argument("--inflight", type=int, default=2,
                    help="sequences kept in flight by the timed loop: that many encoder handles (own work buffers, own streams), each step "
                         "enqueued with m2v_encode_resident_begin and collected with _end when its handle comes round again; 1 = one "
                         "handle, every step a synchronous m2v_encode_resident call (what rounds 1 and 2 timed)")
    ap.add_argument("--ablate", type=int, default=0, help="profiling aid: skip kernel phases (output invalid), see Geom::ablate")
    ap.add_argument("--strip-inflight", type=int, default=-1,
                    help="--mode strips: sequences in flight per rank from ONE thread: that many handles taking turns through m2v_strip_encode_begin / "
                         "_end (with --transport peer a landing block each) over one base communicator.  1: blocking calls only.  -1 (default): 3 with "
                         "the peer transport, 2 with the RCCL form of the step (one rank of 8 alone on a GPU: 0.250 / 0.264 ms per sequence with 3 / 2 "
                         "in flight in the peer form, 0.390 / 0.363 in the RCCL form - profiles/r06_experiments.txt item 20)")
    ap.add_argument("--strip-threads", type=int, default=0,
                    help="--mode strips, opt-in (the round-5 form): K host threads per rank, each with a handle and a communicator stack of its own")
    ap.add_argument("--rotate-dst", action="store_true",
                    help="--mode strips: with sequences in flight, sequence i is assembled on rank i mod N (no single rank carries every gather)")
    ap.add_argument("--strips-legs", choices=["auto", "off"], default="auto",
                    help="N > 1, default mode: after the c4 measurement run config c5 (--mode strips; RCCL halo, then --transport peer) as fresh child "
                         "processes under a time bound and attach the results to the c4 line as `strips` (bench_launch.py).  off: the c4 line alone")
    ap.add_argument("--strips-steps", type=int, default=40, help="steps of each strips leg")
    ap.add_argument("--long-gops", type=int, default=-1,
                    help="--mode strips: also time the in-flight loop on a sequence of this many GOPs (a GOP step is ONE launch over the step's frames of all "
                         "GOPs: longer sequences make larger launches; reported per 90 frames as `in_flight_long_sequence`).  -1 (default): 40 in the strips "
                         "legs of the default N > 1 job, off in a plain --mode strips run; 0: off")
    ap.add_argument("--transport", choices=["rccl", "peer"], default=os.environ.get("M2V_STRIP_TRANSPORT", "rccl"),
                    help="--mode strips, N > 1: how the halo rows travel.  rccl (default): ncclSend / ncclRecv per GOP step.  peer: the edge-row "
                         "kernel stores them straight into the neighbour's landing block (hipIpc-mapped) and counts their arrival - one launch "
                         "per GOP step, no exchange step; sizes and strips still through RCCL.  Never run between two GPUs yet: opt-in")
    ap.add_argument("--mode", choices=["sequences", "strips"], default="sequences",
                    help="sequences (default): config c3 / c4, one 1920x1152 sequence per GPU, no collective; "
                         "strips: config c5, ONE 2048x2048 sequence cut into macroblock-row strips, RCCL halo exchange")
    args = ap.parse_args()

    # ---- N ranks: started by the caller (WORLD_SIZE set) or, failing that, by this process before it touches the GPU ----
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        have = visible_gpus()
        if not args.dry_launch and os.environ.get("M2V_BENCH_SHARE_GPU") != "1" and have is not None and have < args.gpus:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible on this node (one rank per GPU): nothing was started" % (args.gpus, have))
        if not bench_launch.legs_wanted(args, args.gpus):
            sys.exit(bench_launch.launch_ranks(args.gpus, sys.argv[1:])[0])
        # the default N > 1 job: the c4 ranks (their line kept, not relayed), then config c5 twice as fresh ranks under a time bound,
        # then the c4 line - once, last - with `strips` added.  A strips leg never costs the line or changes the exit code.
        rc, lines, _ = bench_launch.launch_ranks(args.gpus, sys.argv[1:], relay=False)
        strips = bench_launch.legs_from_launcher(args) if rc == 0 else {"skipped": "the c4 ranks failed (exit code %d)" % rc}
        bench_launch.attach_and_print(lines, strips)
        sys.exit(rc)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_launch:
        sys.exit(bench_launch.dry_launch(args, rank, world))

    global W, H, XS16, YS16, PFRAMES
    if args.config == "c2":                    # BASELINE configs[1]: 640x480, I frames only, Q_LEVEL 2 - DCT + quantiser + VLC, no search
        W, H, XS16, YS16, PFRAMES = 640, 480, 40, 30, 0
        if args.gops == GOPS:
            args.gops = 256                    # 256 frames = 307 200 macroblocks per launch (a frame is its own GOP)
        if args.mode == "strips":
            raise SystemExit("--config c2 is a --mode sequences workload")

    import torch
    import m2v_load
    M = m2v_load.load()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # Test hook for 1-GPU boxes only: M2V_BENCH_SHARE_GPU=1 puts every rank on GPU 0 and M2V_DIST_BACKEND=gloo replaces
    # RCCL (which refuses two ranks on one device) for the barrier / max-over-ranks.  Never set by the driver.
    if os.environ.get("M2V_BENCH_SHARE_GPU") == "1":
        local_rank = 0
    elif world > torch.cuda.device_count():
        raise SystemExit("bench.py: %d ranks but %d visible GPUs (one rank per GPU)" % (world, torch.cuda.device_count()))
    backend = os.environ.get("M2V_DIST_BACKEND", "nccl")
    dist = None
    torch.cuda.set_device(local_rank)
    ranks_seen = 1
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
        ranks_seen = dist.get_world_size()
    dev = "cuda:%d" % local_rank

    if rank == 0 or world == 1:
        M.build()                              # one rank compiles (if the library is stale at all); the others wait
    if dist is not None:
        dist.barrier()
    if args.mode == "strips":
        import bench_strips
        return bench_strips.bench_strips(args, config(), M, torch, dist, rank, local_rank, world, dev)
    gop = PFRAMES + 1
    nframes = args.gops * gop
    clip = M.synth.clip_torch(W, H, nframes, clip_index=rank, device=dev)       # resident in HBM
    cap = nframes * W * H * 3 // 2
    nh = max(1, args.inflight)
    d_outs = [torch.empty(cap, dtype=torch.uint8, device=dev) for _ in range(nh)]
    d_out = d_outs[0]
    # `enc`: the handle of the synchronous calls (warm-up, the comparison figure, the profiled pass), library defaults: the GOPs of a
    # sequence as two groups on two streams.  `encs`: the handles that take turns in the timed loop; with two sequences in flight the
    # sequences themselves are what overlaps, so each runs on ONE stream (measured: 2 in flight x 1 stream 191.0, 2 x 2 181.5,
    # 3 x 1 189.3, 1 x 2 184.4 GPixel/s on one box)
    enc = M.Mpeg2Encoder(XL, YL, VL, Q, device=local_rank, debug=bool(args.ablate))       # --ablate needs the -DM2V_DEBUG library
    encs = [enc] if nh == 1 else [M.Mpeg2Encoder(XL, YL, VL, Q, device=local_rank, debug=bool(args.ablate)) for _ in range(nh)]
    for h in set(encs + [enc]):
        h.set_option("batch_frames", nframes)
        if args.ablate:
            h.set_option("ablate", args.ablate)
        if args.split >= 0:
            h.set_option("split_streams", args.split)
        elif h is not enc:
            h.set_option("split_streams", 1)
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()                   # the clip is there: the handles' own streams do not wait for torch's

    def step():                                # one synchronous call on one handle (warm-up, the profiled pass, --inflight 1)
        return enc.encode_resident(clip.data_ptr(), nframes, d_out.data_ptr(), cap, XS16, YS16, PFRAMES, stream)

    def run_steps(steps):
        """exactly `steps` sequences, --inflight of them under way at any time: step i goes to handle i mod nh (on the handle's own
        streams) as soon as that handle's previous step has been collected"""
        if nh == 1:
            nb = 0
            for _ in range(steps):
                nb = step()
            return nb
        busy, nb = [False] * nh, 0
        for i in range(steps):
            h = i % nh
            if busy[h]:
                nb = encs[h].encode_resident_end()
            encs[h].encode_resident_begin(clip.data_ptr(), nframes, d_outs[h].data_ptr(), cap, XS16, YS16, PFRAMES, 0)
            busy[h] = True
        for h in range(nh):
            if busy[h]:
                nb = encs[h].encode_resident_end()
        return nb

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:      # wake the device: not counted, not timed
        run_steps(nh)
    nbytes = run_steps(max(args.warmup, nh))

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(steps, fn=None):
        barrier()
        t0 = time.perf_counter()
        nb = (fn or run_steps)(steps)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, nb

    def sync_steps(steps):                     # one blocking call after the other on one handle
        return [step() for _ in range(steps)][-1]

    # How the K sequences are submitted - two handles taking turns, or one blocking call after the other - is settled in the warm-up by an
    # untimed probe of both forms, and a bad placement of the two handles' streams on the runtime's hardware queues is repaired
    # (bench_util.settle_queue_placement: what is probed, the decision table and its thresholds).  The line says what happened.
    submission = "in_flight" if nh > 1 else "blocking"
    placement = None
    if nh > 1:
        for _ in range(max(3, args.warmup)):   # this handle has not run yet: its work buffers are allocated by its first call
            step()
        probe = max(2 * nh, min(10, args.steps))

        def probe_fly():
            run_steps(2 * nh)
            return min(timed(probe)[0], timed(probe)[0])

        def probe_blocking():
            return min(timed(probe, sync_steps)[0], timed(probe, sync_steps)[0])

        def probe_one_stream():                # the yardstick for "no overlap": the same sequences strictly one after the other on ONE stream
            enc.set_option("split_streams", 1)
            t = min(timed(probe, sync_steps)[0], timed(probe, sync_steps)[0])
            enc.set_option("split_streams", args.split if args.split >= 0 else LIB_DEFAULT_SPLIT_STREAMS)
            return t

        def restream(kind):
            # ONE handle is re-streamed, the last one: with the default two handles that is the pair; with --inflight > 2 the others keep
            # the streams they were created with (more than two in flight buys nothing, DESIGN.md section 2, and is not repaired).
            # new_stream: option stream_priority = 0 makes a fresh stream - the next hardware queue in the runtime's rotation;
            # priority: a stream of another priority never shares a queue with default-priority ones
            encs[-1].set_option("stream_priority", 0 if kind == "new_stream" else 1)
        submission, placement = settle_queue_placement(probe_fly, probe_blocking, probe_one_stream, restream, repair_allowed=args.split < 0,
                                                       per_step=1.0 / probe)
    # THE timed region: K steps of the encoder as shipped (no in-band timers)
    dt, nbytes = timed(args.steps, run_steps if submission == "in_flight" else sync_steps)
    # the same K steps the other way (rounds 1 and 2 reported the blocking form as `value`)
    dt_other, nbytes_other = timed(args.steps, sync_steps if submission == "in_flight" else run_steps) if nh > 1 else (dt, nbytes)
    assert nbytes_other == nbytes
    dt_sync, dt_fly = (dt_other, dt) if submission == "in_flight" else (dt, dt_other)
    # The K steps above are ~0.1 s of GPU time.  The same loop again for --sustain seconds of wall clock, in windows of ~0.5 s (a
    # window ends by collecting every handle, a bubble of one sequence in ~500): what the GPU holds once clocks, power and
    # temperature have settled.  Reported beside `value`, never as `value`.
    sustained = None
    if args.sustain > 0:
        fn = run_steps if submission == "in_flight" else sync_steps
        per = max(2 * nh, int(0.5 / max(dt / args.steps, 1e-6)))
        # the card this rank's HIP device IS, by PCI address (a lease that shows HIP one GPU of eight still lists all eight in sysfs)
        try:
            bus_id = M.device_pci_bus_id(local_rank)
        except Exception:  # noqa: BLE001
            bus_id = None
        card = sysfs_card_of(bus_id)
        s0 = gpu_sensors(card)
        # the sensors again every 0.1 s WHILE the loop runs, from a thread of their own (read between two windows the GPU has just gone
        # idle and shows its idle clock)
        import threading
        samples, stop_sampling = [], threading.Event()

        def sampler():
            while not stop_sampling.wait(0.1):
                x = gpu_sensors(card)
                if x:
                    samples.append(x)
        th = threading.Thread(target=sampler, daemon=True)
        barrier()
        th.start()
        wins, t_begin = [], time.perf_counter()
        while time.perf_counter() - t_begin < args.sustain:
            t0 = time.perf_counter()
            nb = fn(per)
            torch.cuda.synchronize()
            wins.append(time.perf_counter() - t0)
            assert nb == nbytes
        t_all = time.perf_counter() - t_begin
        stop_sampling.set()
        th.join(timeout=2.0)
        s1 = gpu_sensors(card)
        plausible, why_not = sensors_verdict(samples)

        def spread(key):
            v = [x[key] for x in samples if x.get(key) is not None]
            return {"min": min(v), "max": max(v), "mean": round(sum(v) / len(v), 1), "samples": len(v)} if v else None
        rate = [per * nframes * W * H / w * 1e-6 for w in wins]
        sustained = {"seconds": round(t_all, 2), "sequences": per * len(wins), "sequences_per_window": per, "windows": len(wins),
                     "value": round(per * len(wins) * nframes * W * H / sum(wins) * 1e-6, 2), "unit": "MPixels/s (this rank)",
                     "window_min": round(min(rate), 2), "window_max": round(max(rate), 2),
                     "first_window": round(rate[0], 2), "last_window": round(rate[-1], 2),
                     "submission": submission, "sensors_card": {"pci_bus_id": bus_id, "sysfs": os.path.dirname(card) if card else None},
                     "sensors_plausible": plausible, "sensors_not_plausible_because": why_not,
                     "sensors_start": s0 if plausible else None, "sensors_end": s1 if plausible else None,
                     "sensors_during": {"sclk_mhz": spread("sclk_mhz"), "power_w": spread("power_w"), "temp_c": spread("temp_c"), "every_s": 0.1} if plausible else None,
                     "sensors_source": "amdgpu sysfs of the card at the HIP device's PCI address (pp_dpm_sclk / pp_dpm_mclk / hwmon)" if card else
                                       "not readable on this host (no /sys/class/drm card at PCI address %s)" % bus_id}
    # second pass, same K steps, for the per-kernel numbers: option "profile" brackets every launch with HIP events on the
    # launch stream and keeps the whole chunk on ONE stream, so that a launch's duration is the kernel alone on the GPU
    enc.set_option("profile", 1)
    step()
    dt_prof, nbytes_prof = timed(args.steps, lambda k: [step() for _ in range(k)][-1])
    assert nbytes_prof == nbytes

    pixels_per_step = nframes * W * H
    value = world * args.steps * pixels_per_step / dt * 1e-6

    rank_parity = None
    if world > 1:
        # N > 1: every rank checks the first GOP of ITS sequence (its own seed) against the oracle - a few seconds of CPU per rank,
        # after the timed regions - and the ranks agree on the verdict.  (The whole-stream check of all GOPs is the N = 1 line's.)
        from oracle import m2v_oracle_ctypes as orc
        if rank == 0:
            orc.build()
        dist.barrier()
        mine = d_out[:nbytes].cpu().numpy().tobytes()
        ref = orc.encode(clip[:gop].cpu().numpy(), XS16, YS16, PFRAMES, XL, YL, VL, Q)
        head, gops, _ = split_gops(mine)
        rhead, rgops, _ = split_gops(ref)
        ok = int(len(gops) == args.gops and head == rhead and len(rgops) == 1 and gops[0] == rgops[0])
        t = torch.tensor([ok], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        rank_parity = {"ranks_checked": world, "gops_compared_per_rank": 1, "identical_to_oracle": bool(int(t.item())),
                       "note": "every rank: sequence headers + first GOP of its own sequence against oracle/m2v_oracle.c"}

    if rank == 0:
        # dominant kernel: k_mb<3,true> (P-frame macroblock kernel); for the I-only config c2 it is k_mb<1,false>.
        # Statistics of the LAST step.
        launches, ms, px = enc.kernel_stats(0)
        li, msi, pxi = enc.kernel_stats(1)
        l3, ms3, _ = enc.kernel_stats(3)
        l4, ms4, _ = enc.kernel_stats(4)
        if PFRAMES > 0:
            # algorithmic HBM bytes per luma pixel of a P frame: 3.0 input 4:4:4 + 1.5 reference
            # load + 1.5 reconstruction store (frames that are referenced later); the bitstream leaves through k_assemble
            dom, dom_name, dom_launches, dom_ms, tkey = "P", "k_mb<3,true> (P-frame macroblock kernel)", launches, ms, "k_mb_p_bytes_per_launch"
            alg_bytes = (args.gops * (PFRAMES - 1) * 6.0 + args.gops * 4.5) * W * H
        else:
            # I frames that nothing references: 3.0 B/px of input, no reconstruction store
            dom, dom_name, dom_launches, dom_ms, tkey = "I", "k_mb<1,false> (I-frame macroblock kernel, no search)", li, msi, "k_mb_i_c2_bytes_per_launch"
            alg_bytes = nframes * 3.0 * W * H
        achieved = alg_bytes / (dom_ms * 1e-3) * 1e-9 if dom_ms > 0 else 0.0
        traffic = pmc_traffic(tkey) if PFRAMES > 0 and args.gops == GOPS or PFRAMES == 0 and nframes == 256 else {"traffic": None}   # measured for these launches
        workload = ("c3: 1920x1152 yuv444p, %d closed GOPs of 1 I + %d P frames (%d frames), VECTOR_LEVEL=3 Q_LEVEL=2 XL=YL=7, "
                    "one independent sequence per GPU (c4 for N > 1), no data-path collective" % (args.gops, PFRAMES, nframes)) if PFRAMES else \
                   ("c2: 640x480 yuv444p, %d I frames (i_pframes_count = 0), Q_LEVEL=2 XL=YL=7, one independent sequence per GPU, "
                    "no data-path collective" % nframes)
        out = {
            "metric": "MPixels/s encoded, %dx%d %s" % (W, H, "I+P" if PFRAMES else "I only"),
            "value": round(value, 2), "unit": "MPixels/s", "n_gpus": ranks_seen, "ranks_seen": ranks_seen, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": round(value / FPGA_MPIXELS, 3), "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": workload, "frames": nframes, "stream_bytes": int(nbytes),
                       "bits_per_pixel": round(nbytes * 8 / pixels_per_step, 4),
                       "baseline": "FPGA Kintex-7 268 MPixels/s (README.md:22)",
                       "launched_by": os.environ.get("M2V_BENCH_LAUNCHED_BY", "caller"),
                       "dist_backend": backend if world > 1 else None,
                       "sequences_in_flight": nh if submission == "in_flight" else 1,
                       "submission": submission + (" (checked by an untimed probe of both forms in the warm-up)" if nh > 1 else ""),
                       "queue_placement": placement},
            # `value`: K sequences, `sequences_in_flight` encoder handles taking turns (m2v_encode_resident_begin / _end): the stream
            # assembly of one sequence runs beside the first macroblock kernels of the next.  One handle, one blocking call per
            # sequence - what rounds 1 and 2 reported as `value` - is the entry below.
            "one_synchronous_call_per_step": {"value": round(world * args.steps * pixels_per_step / dt_sync * 1e-6, 2),
                                              "ms_per_step": round(dt_sync / args.steps * 1e3, 3)},
            "sequences_in_flight_loop": {"value": round(world * args.steps * pixels_per_step / dt_fly * 1e-6, 2),
                                         "ms_per_step": round(dt_fly / args.steps * 1e3, 3), "handles": nh},
            "roofline": dict({"bound": "hbm", "kernel": dom_name,
                              "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 5)}, **traffic, **pmc_valu_busy(tkey),
                             **{"launches_per_step": dom_launches, "avg_launch_ms": round(dom_ms / max(dom_launches, 1), 4),
                                "algorithmic_bytes_per_launch": round(alg_bytes / max(dom_launches, 1))}),
            "kernel_ms_per_step": {"k_mb_P": round(ms, 3), "k_mb_I": round(msi, 3), "k_assemble": round(ms3, 3),
                                   "scans_headers": round(ms4, 3)},
            # `value` is the encoder as shipped: the closed GOPs of the chunk run as two groups on two HIP streams, so the
            # partially filled tail of one group's launch overlaps with the other group's next launch.  The per-kernel
            # durations above (and the roofline) come from the second pass of the same K steps with in-band HIP events,
            # which runs everything on one stream: a launch's duration there is the kernel alone on the whole GPU.
            "profiled_pass": {"value": round(world * args.steps * pixels_per_step / dt_prof * 1e-6, 2),
                              "ms_per_step": round(dt_prof / args.steps * 1e3, 3), "steps": args.steps,
                              "streams": 1, "in_band_event_timers": True},
            "streams": {"per_sequence_in_the_timed_loop": args.split if args.split >= 0 else (1 if submission == "in_flight" else 2),
                        "per_sequence_in_a_synchronous_call": args.split if args.split >= 0 else 2},
        }
        if sustained is not None:
            sustained["vs_value"] = round(sustained["value"] * world / value, 4) if value > 0 else None
            out["sustained"] = sustained
        if dom == "P":
            # secondary "operation roofline" of SURVEY.md 8(d): the search alone is (2*6+1)^2 + 9 = 178 byte absolute
            # differences per luma pixel of a P frame; v_qsad / v_sad retire one per lane per clock
            out["op_roofline"] = {"bound": "valu-sad", "unit": "T byte-absdiff/s",
                                  "achieved": round(178.0 * px / (ms * 1e-3) * 1e-12, 2) if ms > 0 else 0.0,
                                  "peak": round(256 * 4 * 64 * 2.4e9 * 1e-12, 1),
                                  "frac": round(178.0 * px / (ms * 1e-3) / (256 * 4 * 64 * 2.4e9), 4) if ms > 0 else 0.0,
                                  "note": "256 CUs x 4 SIMDs x 64 lanes x 2.4 GHz; every unit the macroblock kernel touches is 65-90 % busy "
                                          "(vector ALU 85-89 %, SQ_ACTIVE_INST_VALU; LDS ~70 %, scalar unit ~69 %): the full-pel search is ~30 % of its "
                                          "vector cycles (52 v_qsad_pk_u16_u8 = 832 of ~2 800) at 81 % of the instruction's candidate slots "
                                          "(169 of 208 live) - DESIGN.md section 4, profiles/r05_final_pmc_sq.json"}
        out["roofline"]["timed_in"] = ("profiled_pass (one stream; HIP events on the launch stream around every RUN of consecutive launches of one kernel - the eight P launches of "
                                       "a sequence are one interval, duration / launches: an event in every gap costs the next launch ~3 us)")
        if rank_parity is not None:
            out["parity_check"] = rank_parity
        if world == 1 and not args.no_cpu_baseline:
            clip_np = clip.cpu().numpy()
            gpu_bytes = d_out[:nbytes].cpu().numpy().tobytes()
            out["handles_agree"] = all(torch.equal(d_outs[0][:nbytes], d[:nbytes]) for d in d_outs[1:])
            out["cpu_baseline"] = cpu_baseline(clip_np[:gop if PFRAMES else min(nframes, 96)])
            out["cpu_baseline_all_cores"], out["parity_check"] = cpu_baseline_all_cores(clip_np, gpu_bytes)
            out["rtl_sim"] = rtl_sim_probe()
            copy = hbm_copy_rate(torch, dev)
            out["roofline"]["hbm_copy_measured"] = round(copy, 1)
            out["roofline"]["frac_of_measured_copy"] = round(achieved / copy, 5) if copy > 0 else None
            if not args.no_e2e:
                import bench_e2e
                out["end_to_end"] = bench_e2e.end_to_end(M, config(), clip_np, gpu_bytes)
    for h in set(encs + [enc]):
        h.close()
    if bench_launch.legs_wanted(args, world) and os.environ.get("M2V_BENCH_LAUNCHED_BY", "caller") == "caller":
        # The driver's form of the N > 1 job (torch.distributed.run started this rank): config c5 now, as ONE fresh child process per
        # rank and leg on a rendezvous port of its own, under a wall-clock bound - a child, never an exec: this process has touched the
        # GPU.  Whatever the legs do, the c4 line below is printed, once, last (bench_launch.py).
        strips = None
        try:
            del clip, d_outs, d_out
            torch.cuda.empty_cache()
            ports = torch.tensor([bench_launch.free_port() for _ in bench_launch.LEGS] if rank == 0 else [0] * len(bench_launch.LEGS),
                                 dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
            dist.broadcast(ports, src=0)
            strips = bench_launch.legs_from_rank(args, rank, world, [int(x) for x in ports.tolist()])
        except Exception as ex:  # noqa: BLE001
            strips = {"error": "could not start the strips legs: %r" % (ex,)}
        if rank == 0:
            out["strips"] = strips
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
