#!/usr/bin/env python3
"""Experiment: option cu_pack (xcd_remap) - which macroblocks share a CU in time - on config c3, blocking calls; bytes compared.
usage: python tools/cu_pack_ab.py [values...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m2v_load

M = m2v_load.load()
W, H, n, pf = 1920, 1152, 90, 8
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
cap = n * W * H * 3 // 2
out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
vals = [int(v) for v in sys.argv[1:]] or [0, 5, 4, 3, 6, 0]
ref = None
for rep in range(2):
    for v in vals:
        enc = M.Mpeg2Encoder(7, 7, 3, 2, device=0)
        enc.set_option("batch_frames", n)
        enc.set_option("cu_pack", v)
        enc.set_option("split_streams", 1)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            nb = enc.encode_resident(clip.data_ptr(), n, out.data_ptr(), cap, W // 16, H // 16, pf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            nb = enc.encode_resident(clip.data_ptr(), n, out.data_ptr(), cap, W // 16, H // 16, pf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        data = out[:nb].clone()
        if ref is None:
            ref = data
        enc.set_option("profile", 1)
        enc.encode_resident(clip.data_ptr(), n, out.data_ptr(), cap, W // 16, H // 16, pf)
        ks = [round(enc.kernel_stats(k)[1], 3) for k in (0, 1, 3, 4)]
        print("cu_pack %d  %.3f ms/sequence  kernels [P, I, assemble, scans] %s  bytes %s" % (v, dt * 1e3, ks, "same" if torch.equal(data, ref) else "DIFFERENT"))
        sys.stdout.flush()
        enc.close()
