#!/usr/bin/env python3
"""Where the host-memory-to-host-memory time of bench.py's end_to_end leg goes: the same calls (one GOP per m2v_push_frames from
page-locked memory, m2v_pull after each, m2v_sequence_stop, drain), timed per kind of call.  usage (GPU box): python tools/e2e_breakdown.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import m2v_load

M = m2v_load.load()
W, H, PF, n = 1920, 1152, 8, 90
gop = PF + 1
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0").cpu()
pinned_t = clip.pin_memory()
frames = pinned_t.numpy()
import hashlib
for batch, into, deferred in ((2 * gop, False, 0), (2 * gop, True, 0), (2 * gop, True, 1), (gop, True, 1), (2 * gop, True, 0), (2 * gop, True, 1)):
  enc = M.Mpeg2Encoder(7, 7, 3, 2)
  enc.set_option("batch_frames", batch)
  enc.set_option("direct_upload", 2 if deferred else 1)
  outbuf = np.empty(n * W * H * 3 // 2, np.uint8)
  for rep in range(4):
    t = {"push": 0.0, "pull": 0.0, "stop": 0.0, "drain": 0.0}
    t0 = time.perf_counter()
    out, pos = [], 0
    for k in range(0, n, gop):
        a = time.perf_counter()
        enc.push_frames(W // 16, H // 16, PF, frames[k:k + gop])
        b = time.perf_counter()
        if into:
            pos += enc.pull_into(outbuf, pos)[0]
        else:
            out.append(enc.pull(1 << 24)[0])
        c = time.perf_counter()
        t["push"] += b - a
        t["pull"] += c - b
    a = time.perf_counter()
    enc.sequence_stop()
    b = time.perf_counter()
    if into:
        last = False
        while not last:
            m, last = enc.pull_into(outbuf, pos)
            pos += m
    else:
        out.append(enc.pull_all())
    c = time.perf_counter()
    t["stop"], t["drain"] = b - a, c - b
    total = c - t0
    data = outbuf[:pos].tobytes() if into else b"".join(out)
    print("%s batch_frames %2d %s total %.2f ms (%.1f GB/s of input)  " % ("deferred" if deferred else "blocking", batch, "pull_into" if into else "pull     ", total * 1e3, n * W * H * 3 / total * 1e-9) + "  ".join("%s %.2f" % (k, v * 1e3) for k, v in t.items()),
          " bytes", len(data), hashlib.sha1(data).hexdigest()[:10])
  enc.close()
enc = M.Mpeg2Encoder(7, 7, 3, 2)
dev = torch.empty_like(pinned_t, device="cuda")
dev.copy_(pinned_t, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
dev.copy_(pinned_t, non_blocking=True)
torch.cuda.synchronize()
print("plain pinned copy of the clip: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
enc.close()

# two callers at once (two threads, a handle and a page-locked clip each): what the link gives when one caller's turn-around
# is covered by the other's upload
import threading

clips = [pinned_t, clip.clone().pin_memory()]
encs = [M.Mpeg2Encoder(7, 7, 3, 2) for _ in range(2)]
for e in encs:
    e.set_option("batch_frames", 2 * gop)


def one(e, fr, res, i):
    out = []
    for k in range(0, n, gop):
        e.push_frames(W // 16, H // 16, PF, fr[k:k + gop])
        out.append(e.pull(1 << 24)[0])
    e.sequence_stop()
    out.append(e.pull_all())
    res[i] = sum(len(o) for o in out)


for rep in range(4):
    res = [0, 0]
    th = [threading.Thread(target=one, args=(encs[i], clips[i].numpy(), res, i)) for i in range(2)]
    t0 = time.perf_counter()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.perf_counter() - t0
    print("two callers: %.2f ms for 2 x %d frames = %.1f GB/s of input, %.1f GPixel/s  bytes %s" % (dt * 1e3, n, 2 * n * W * H * 3 / dt * 1e-9, 2 * n * W * H / dt * 1e-9, res))
for e in encs:
    e.close()
