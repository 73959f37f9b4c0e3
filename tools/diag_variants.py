#!/usr/bin/env python3
"""Bisect what makes bench.py's timed loop slower than tools/diag_steps.py's on some boxes: same step, one
ingredient of bench.py's set-up at a time.  usage: diag_variants.py <variant>"""
import os
import sys
import time

sys.path.insert(0, ".")
variant = sys.argv[1]
import torch
import m2v_load

M = m2v_load.load()
W, H, pf, n = 1920, 1152, 8, 90
if "setdev" in variant:
    torch.cuda.set_device(0)
if "build" in variant:
    M.build()
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
cap = n * W * H * 3 // 2
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
enc = M.Mpeg2Encoder(7, 7, 3, 2, device=0)
enc.set_option("batch_frames", n)
if "cupack0" in variant:          # (tools/variant_run.sh passes the library's name as part of the variant)
    enc.set_option("cu_pack", 0)
stream = torch.cuda.current_stream().cuda_stream


def step():
    return enc.encode_resident(clip.data_ptr(), n, d_out.data_ptr(), cap, W // 16, H // 16, pf, stream)


if "prewarm" in variant:
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.5:
        step()
for _ in range(20):
    step()
if "profile" in variant:
    enc.set_option("profile", 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    nb = step()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
extra = ""
if "stats" in variant:
    extra = " kernel_ms %s" % [round(enc.kernel_stats(k)[1], 3) for k in (0, 1, 3, 4)]
print("%-40s %.3f ms/step%s" % (variant, dt / 100 * 1e3, extra))
enc.close()
