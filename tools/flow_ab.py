#!/usr/bin/env python3
"""Same-box comparison of the resident entry with and without option "flow" (all P frames of a chunk in one launch) on the benchmark
clip (config c3), plus - in the -DM2V_DEBUG library - the timing experiments that take the hand-off apart (results invalid there):
    ablate 64   FLOW without the poll of the reference rows
    ablate 128  FLOW with plain 4-byte stores instead of write-through 16-byte ones
usage: python tools/flow_ab.py [--gops 10] [--steps 60] [--inflight 1|2]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--gops", type=int, default=10)
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1152)
ap.add_argument("--variants", nargs="+", default=["rel:0:0", "rel:1:0", "dbg:0:0", "dbg:1:0", "dbg:1:64", "dbg:1:128", "dbg:1:192", "rel:0:0", "rel:1:0"])
ap.add_argument("--inflight", type=int, default=1)
args = ap.parse_args()
import torch
import m2v_load

M = m2v_load.load()
W, H, pf = args.w, args.h, 8
n = args.gops * (pf + 1)
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
cap = n * W * H * 3 // 2
outs = [torch.empty(cap, dtype=torch.uint8, device="cuda:0") for _ in range(args.inflight)]
torch.cuda.synchronize()
for v in args.variants:
    lib, flow, abl = v.split(":")
    encs = [M.Mpeg2Encoder(7, 7, 3, 2, device=0, debug=(lib == "dbg")) for _ in range(args.inflight)]
    for e in encs:
        e.set_option("batch_frames", n)
        e.set_option("flow", int(flow))
        if int(abl):
            e.set_option("ablate", int(abl))
        if args.inflight > 1:
            e.set_option("split_streams", 1)

    def run(k):
        if args.inflight == 1:
            for _ in range(k):
                encs[0].encode_resident(clip.data_ptr(), n, outs[0].data_ptr(), cap, W // 16, H // 16, pf)
            return
        busy = [False] * args.inflight
        for i in range(k):
            h = i % args.inflight
            if busy[h]:
                encs[h].encode_resident_end()
            encs[h].encode_resident_begin(clip.data_ptr(), n, outs[h].data_ptr(), cap, W // 16, H // 16, pf)
            busy[h] = True
        for h in range(args.inflight):
            if busy[h]:
                encs[h].encode_resident_end()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        run(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    encs[0].set_option("profile", 1)
    encs[0].encode_resident(clip.data_ptr(), n, outs[0].data_ptr(), cap, W // 16, H // 16, pf)
    ks = [round(encs[0].kernel_stats(k)[1], 3) for k in (0, 1, 3, 4)]
    print("%-12s inflight %d  %.3f ms/sequence  %.1f GPixel/s   profiled kernels [P, I, assemble, scans] ms %s  flow_state %s"
          % (v, args.inflight, dt * 1e3, n * W * H / dt * 1e-9, ks, encs[0].flow_state()))
    sys.stdout.flush()
    for e in encs:
        e.close()
