#!/usr/bin/env python3
"""PCIe-inclusive throughput: host yuv444p frames in (m2v_push_frames), stream bytes out (m2v_pull) on 1920x1152 I+P.
This is the end-to-end rate of the port-level interface; it is never bench.py's `value` (inputs resident in HBM)."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import m2v_load

M = m2v_load.load()
W, H, pf, n = 1920, 1152, 8, 90
import torch
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0").cpu().numpy()
enc = M.Mpeg2Encoder(7, 7, 3, 2)
enc.set_option("batch_frames", 90)
for it in range(3):
    t0 = time.perf_counter()
    data = enc.encode(clip, W // 16, H // 16, pf)
    dt = time.perf_counter() - t0
    print("pass %d: %d frames %dx%d from host memory -> %d bytes in %.1f ms = %.0f MPixels/s (input %.1f GB/s over PCIe)"
          % (it, n, W, H, len(data), dt * 1e3, n * W * H / dt * 1e-6, n * W * H * 3 / dt * 1e-9))
enc.close()
