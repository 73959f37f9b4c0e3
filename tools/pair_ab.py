#!/usr/bin/env python3
"""Same-box A/B of option "pair_mb" (two macroblocks per wavefront, the second one's loads in flight during the first) on configs c3 and c2:
blocking single-handle calls and two handles in flight, bytes compared between the two settings.   python tools/pair_ab.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m2v_load

M = m2v_load.load()


def run(W, H, n, pf, pair, inflight, steps):
    clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
    cap = n * W * H * 3 // 2
    outs = [torch.empty(cap, dtype=torch.uint8, device="cuda:0") for _ in range(inflight)]
    encs = [M.Mpeg2Encoder(7, 7, 3, 2, device=0) for _ in range(inflight)]
    for e in encs:
        e.set_option("batch_frames", n)
        e.set_option("pair_mb", pair)
        if inflight > 1:
            e.set_option("split_streams", 1)
    torch.cuda.synchronize()
    nb = [0]

    def go(k):
        if inflight == 1:
            for _ in range(k):
                nb[0] = encs[0].encode_resident(clip.data_ptr(), n, outs[0].data_ptr(), cap, W // 16, H // 16, pf)
            return
        busy = [False] * inflight
        for i in range(k):
            h = i % inflight
            if busy[h]:
                nb[0] = encs[h].encode_resident_end()
            encs[h].encode_resident_begin(clip.data_ptr(), n, outs[h].data_ptr(), cap, W // 16, H // 16, pf)
            busy[h] = True
        for h in range(inflight):
            if busy[h]:
                nb[0] = encs[h].encode_resident_end()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        go(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    go(steps)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    encs[0].set_option("profile", 1)
    encs[0].encode_resident(clip.data_ptr(), n, outs[0].data_ptr(), cap, W // 16, H // 16, pf)
    ks = [round(encs[0].kernel_stats(k)[1], 3) for k in (0, 1, 3, 4)]
    data = outs[0][:nb[0]].clone()
    for e in encs:
        e.close()
    return dt, ks, data


for name, (W, H, n, pf, steps) in {"c3": (1920, 1152, 90, 8, 60)}.items():
    ref = None
    for rep in range(2):
        for inflight in (1, 2):
            for pair in (0, 1):
                dt, ks, data = run(W, H, n, pf, pair, inflight, steps)
                if ref is None:
                    ref = data
                same = data.numel() == ref.numel() and bool(torch.equal(data, ref))
                print("%s pair_mb %d inflight %d  %.3f ms/sequence  %.1f GPixel/s  kernels [P, I, assemble, scans] %s  bytes %s"
                      % (name, pair, inflight, dt * 1e3, n * W * H / dt * 1e-9, ks, "same" if same else "DIFFERENT"))
                sys.stdout.flush()
