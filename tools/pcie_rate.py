#!/usr/bin/env python3
"""PCIe-inclusive throughput: host yuv444p frames in (m2v_push_frames), stream bytes out (m2v_pull) on 1920x1152 I+P.
This is the end-to-end rate of the port-level interface; it is never bench.py's `value` (inputs resident in HBM).
Compares the double-buffered port path (option async=1: the host fills one pinned stage while the previous chunk is
uploaded, encoded and read back) with the synchronous one."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import m2v_load

M = m2v_load.load()
W, H, pf, n = 1920, 1152, 8, 270
import torch
base = M.synth.clip_torch(W, H, 90, clip_index=0, device="cuda:0").cpu().numpy()
clip = np.concatenate([base, base, base])            # 270 frames = 30 GOPs, 1.8 GB of host memory
want = None
for use_async, bf, threads in ((0, 90, 1), (1, 90, 1), (1, 90, 2), (1, 90, 4), (1, 90, 8), (1, 27, 4)):
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    enc.set_option("batch_frames", bf)
    enc.set_option("async", use_async)
    enc.set_option("copy_threads", threads)
    best = 1e9
    for it in range(3):
        t0 = time.perf_counter()
        out = []
        for k in range(0, n, 9):                       # the caller hands over one GOP at a time and drains as it goes
            enc.push_frames(W // 16, H // 16, pf, clip[k:k + 9])
            b, _ = enc.pull(1 << 24)
            out.append(b)
        enc.sequence_stop()
        out.append(enc.pull_all())
        dt = time.perf_counter() - t0
        best = min(best, dt)
        data = b"".join(out)
        if want is None:
            want = data
        assert data == want, "stream differs between modes"
    print("async=%d batch_frames=%d copy_threads=%d: %d frames %dx%d host -> %d bytes, best of 3 %.1f ms = %.0f MPixels/s (input %.1f GB/s)"
          % (use_async, bf, threads, n, W, H, len(data), best * 1e3, n * W * H / best * 1e-6, n * W * H * 3 / best * 1e-9))
    enc.close()
