#!/usr/bin/env python3
"""Ad-hoc wide-frame differential run (HIP path vs oracle): widths up to the XL=7 maximum (128 macroblocks per slice: the
largest slice tables of k_slice_scan / k_assemble), every content kind of tests/test_gpu_fuzz.py, both interfaces.
usage (GPU box): python tools/fuzz_wide.py [cases] [seed]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np

import gpu_util as G
from oracle import m2v_oracle_ctypes as orc
from test_gpu_fuzz import make_content

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
bad = 0
for case in range(n_cases):
    W = 16 * int(rng.choice([128, 127, 120, 97, 64]))
    H = 16 * int(rng.integers(4, 9))
    VL, Q = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    pf = int(rng.choice([0, 2, 8, 255]))
    n = int(rng.integers(2, 6))
    bf = int(rng.choice([2, 96]))
    clip = make_content(rng, G.M, W, H, n)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q)
    got = G.resident_encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, batch_frames=bf)
    enc = G.M.Mpeg2Encoder(7, 7, VL, Q)
    try:
        enc.set_option("batch_frames", bf)
        got2 = enc.encode(clip, W // 16, H // 16, pf)
        enc.set_option("split_streams", 1)
        got3 = enc.encode(clip, W // 16, H // 16, pf)
        # the beats as packed samples, a random layout, pushes that line up with nothing (m2v_push_packed: interleaved until in HBM)
        layout = ["yuv24", "uyv24", "yuvx32", "ayuv32"][int(rng.integers(0, 4))]
        y, u, v = clip[:, 0], clip[:, 1], clip[:, 2]
        pad = np.full_like(y, 0x3C)
        order = {"yuv24": (y, u, v), "uyv24": (u, y, v), "yuvx32": (y, u, v, pad), "ayuv32": (pad, y, u, v)}[layout]
        packed = np.ascontiguousarray(np.stack(order, axis=-1)).reshape(-1)
        bpp, beats, b = len(order), n * W * H // 4, 0
        while b < beats:
            take = int(min(beats - b, rng.choice([1, 333, W * H // 4, 2 * (W * H // 4) + 7])))
            enc.push_packed(W // 16, H // 16, pf, packed[4 * bpp * b:4 * bpp * (b + take)], layout, stop_with_last=(b + take == beats))
            b += take
        got4 = enc.pull_all()
    finally:
        enc.close()
    ok = got == want and got2 == want and got3 == want and got4 == want
    bad += not ok
    print("case %2d %4dx%-4d n=%d pf=%3d VL=%d Q=%d batch=%2d  %7d bytes  %s" % (case, W, H, n, pf, VL, Q, bf, len(want), "ok" if ok else "MISMATCH"))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
