#!/usr/bin/env python3
"""Timeline of bench.py's step on this box: wall time per step in blocks of 50 steps, with and without the per-kernel
HIP-event timers (option "profile").  Diagnoses clock ramps / throttling / host launch overhead."""
import sys
import time

sys.path.insert(0, ".")
import numpy as np
import torch
import m2v_load

M = m2v_load.load()
W, H, pf, n = 1920, 1152, 8, 90
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
cap = n * W * H * 3 // 2
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
enc = M.Mpeg2Encoder(7, 7, 3, 2)
enc.set_option("batch_frames", n)
stream = torch.cuda.current_stream().cuda_stream
# host-side yardsticks: a pure-Python loop, a HIP API call that does no work, a tiny launch + sync round trip
t0 = time.perf_counter(); x = 0
for i in range(2000000):
    x += i
t_py = time.perf_counter() - t0
t0 = time.perf_counter()
for _ in range(2000):
    enc.busy
t_api = (time.perf_counter() - t0) / 2000
z = torch.zeros(64, device="cuda:0")
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    z.add_(1)
    torch.cuda.synchronize()
t_rt = (time.perf_counter() - t0) / 500
t0 = time.perf_counter()
for _ in range(2000):
    z.add_(1)
t_launch = (time.perf_counter() - t0) / 2000
torch.cuda.synchronize()
print("host: python 2M-iteration loop %.0f ms, ctypes call %.2f us, launch %.1f us, launch+sync round trip %.1f us, load %s"
      % (t_py * 1e3, t_api * 1e6, t_launch * 1e6, t_rt * 1e6, open("/proc/loadavg").read().split()[:3]))
import os
for profile in (0, 1, 0, 2, 0, 2):
    enc.set_option("split_streams", 1 if profile == 2 else 0)
    enc.set_option("profile", 1 if profile == 1 else 0)
    torch.cuda.synchronize()
    rows = []
    t_start = time.perf_counter()
    for blk in range(6):
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            enc.encode_resident(clip.data_ptr(), n, d_out.data_ptr(), cap, W // 16, H // 16, pf, stream)
            ts.append(time.perf_counter() - t0)
        ts = np.array(ts) * 1e3
        rows.append("%.2f/%.2f/%.2f" % (ts.min(), np.median(ts), ts.max()))
    print("profile=%d  %.2f s   min/median/max ms per 50 steps: %s" % (profile, time.perf_counter() - t_start, "  ".join(rows)))
    sys.stdout.flush()
enc.close()
