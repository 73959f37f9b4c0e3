"""The port path from one thread, three ways (page-locked push + pull, page-locked m2v_push_frames_pull, pageable push + pull): best of five,
ms per 90-frame 1920x1152 clip.  With M2V_LIB=<other build> it is the same-box A/B of two libraries (profiles/r05_experiments.txt item 16).
    python tools/e2e_ab.py"""
import os, sys, time
sys.path.insert(0, ".")
import numpy as np, torch, m2v_load
M = m2v_load.load()
W, H, PF, n = 1920, 1152, 8, 90
gop = PF + 1
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0").cpu()
pinned = clip.pin_memory().numpy(); pageable = clip.numpy().copy()
for name, src, one in (("pinned two calls", pinned, 0), ("pinned one call", pinned, 1), ("pageable two calls", pageable, 0)):
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    enc.set_option("batch_frames", gop)
    out = np.empty(n * W * H * 3 // 2, np.uint8)
    best = 1e9
    for rep in range(5):
        t0 = time.perf_counter(); pos = 0
        for k in range(0, n, gop):
            if one: pos += enc.push_frames_pull(W // 16, H // 16, PF, src[k:k + gop], out, pos)[0]
            else:
                enc.push_frames(W // 16, H // 16, PF, src[k:k + gop]); pos += enc.pull_into(out, pos)[0]
        enc.sequence_stop(); last = False
        while not last:
            m, last = enc.pull_into(out, pos); pos += m
        best = min(best, time.perf_counter() - t0)
    print("%-20s %.2f ms  %.1f GB/s" % (name, best * 1e3, n * W * H * 3 / best * 1e-9))
    enc.close()
