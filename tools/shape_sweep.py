#!/usr/bin/env python3
"""Throughput of the resident path across SHAPES: GOP lengths (i_pframes_count 0 .. 89) at 1920x1152 and frame sizes at i_pframes_count 8, always about
the same number of pixels per sequence, two handles in flight (what bench.py times for config c3); VECTOR_LEVEL 3, Q_LEVEL 2.  Looks for anomalies the
benchmark's one shape cannot show (tools/param_sweep.py found one across VECTOR_LEVEL).
    usage (GPU box): python tools/shape_sweep.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m2v_load

M = m2v_load.load()
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
PIX = 90 * 1920 * 1152


def run_case(W, H, n, pf, tag):
    clip = M.synth.clip_torch(W, H, n, clip_index=1, device="cuda:0")
    cap = n * W * H * 3 // 2 + 65536
    outs = [torch.empty(cap, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    encs = [M.Mpeg2Encoder(7, 7, 3, 2) for _ in range(2)]
    torch.cuda.synchronize()
    try:
        for e in encs:
            e.set_option("batch_frames", n)
            e.set_option("split_streams", 1)

        def run(k):
            busy, nb = [False, False], 0
            for i in range(k):
                h = i % 2
                if busy[h]:
                    nb = encs[h].encode_resident_end()
                encs[h].encode_resident_begin(clip.data_ptr(), n, outs[h].data_ptr(), cap, W // 16, H // 16, pf, 0)
                busy[h] = True
            for h in range(2):
                if busy[h]:
                    nb = encs[h].encode_resident_end()
            return nb
        run(10)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nb = run(steps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        same = bool(torch.equal(outs[0][:nb], outs[1][:nb]))
        print(json.dumps({"case": tag, "W": W, "H": H, "frames": n, "pframes_count": pf, "gops": (n + pf) // (pf + 1), "MPixels_per_s": round(steps * n * W * H / dt * 1e-6, 1),
                          "ms_per_sequence": round(dt / steps * 1e3, 4), "bits_per_pixel": round(nb * 8 / (n * W * H), 4), "handles_agree": same}), flush=True)
    finally:
        for e in encs:
            e.close()
        del clip, outs
        torch.cuda.empty_cache()


for pf in (0, 1, 2, 4, 8, 14, 29, 44, 89):
    run_case(1920, 1152, 90, pf, "gop length")
for W, H in ((288, 208), (640, 480), (1440, 704), (1920, 1152), (2048, 2048), (2048, 1024), (1024, 2048)):
    n = max(9, (PIX // (W * H)) // 9 * 9)
    run_case(W, H, n, 8, "frame size")
