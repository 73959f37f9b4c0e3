#!/usr/bin/env python3
"""Strip sequences in flight on ONE rank from ONE thread: tools/strip_solo.py's set-up (rank r of 8 alone on a GPU, peer transport on a
solo base) with K handles taking turns through m2v_strip_encode_begin / _end - a peer communicator (landing block) per handle over ONE
shared base communicator.  K = 1 is the blocking call's timing with the call split in two.  ms per sequence = wall time / sequences.
    usage (GPU box): python tools/strip_solo_turns.py [--handles 1 2 3] [--rccl 0|1] [--peer 1|0] [--seconds 1.5]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m2v_load

ap = argparse.ArgumentParser()
ap.add_argument("--handles", type=int, nargs="+", default=[1, 2, 3])
ap.add_argument("--rccl", type=int, default=0, help="1: the base is a 1-rank RCCL communicator (sizes all-gather and halo as RCCL kernels)")
ap.add_argument("--peer", type=int, default=1)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--seconds", type=float, default=1.5)
ap.add_argument("--ranks", type=int, nargs="+", default=None)
ap.add_argument("--gops", type=int, default=10, help="GOPs (of 1 I + 8 P) in the sequence: a GOP step is ONE launch over the step's frames of all GOPs")
ap.add_argument("--split", type=int, nargs="+", default=[-1], help="option split_streams of the handles (GOP groups on a stream each); -1 = the library's default")
args = ap.parse_args()
M = m2v_load.load()
W = H = 2048
pf, n, world = 8, 9 * args.gops, args.world
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
torch.cuda.synchronize()
for K, split in [(k, sp) for k in args.handles for sp in args.split]:
    for rank in (args.ranks if args.ranks is not None else sorted({0, world // 2})):
        base = M.StripComm.solo(world, rccl=bool(args.rccl))
        encs = [M.Mpeg2Encoder(7, 7, 3, 2) for _ in range(K)]
        if split >= 0:
            for e in encs:
                e.set_option("split_streams", split)
        comms = [M.StripComm.peer(base, rank, 0) if args.peer else base for _ in range(K)]
        outs = [torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0") if rank == 0 else None for _ in range(K)]
        torch.cuda.synchronize()

        def run(limit_s=None, count=None):
            busy, done, t0 = [False] * K, 0, time.perf_counter()
            i = 0
            while (count is not None and i < count) or (limit_s is not None and time.perf_counter() - t0 < limit_s):
                h = i % K
                if busy[h]:
                    M.parallel.encode_strips_native_end(encs[h], outs[h], rank)
                    done += 1
                M.parallel.encode_strips_native_begin(encs[h], comms[h], rank, world, clip, 128, 128, pf, outs[h])
                busy[h] = True
                i += 1
            for k in range(K):
                h = (i + k) % K
                if busy[h]:
                    M.parallel.encode_strips_native_end(encs[h], outs[h], rank)
                    done += 1
            return done
        try:
            run(count=max(30, 3000 // args.gops))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            seqs = run(limit_s=args.seconds)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            print(json.dumps({"handles_in_flight": K, "split_streams": split, "rank": rank, "world": world, "base": base.kind, "peer": comms[0].peer_stats() if args.peer else None,
                              "form": encs[0].strip_last_form(), "gops": args.gops, "sequences": seqs, "ms_per_sequence": round(dt / seqs * 1e3, 4),
                              "ms_per_90_frames": round(dt / seqs * 1e3 * 10 / args.gops, 4)}), flush=True)
        finally:
            for c in comms:
                if c is not base:
                    c.close()
            for e in encs:
                e.close()
            base.close()
