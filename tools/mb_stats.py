#!/usr/bin/env python3
"""Macroblock statistics of the benchmark clip (what the data-dependent paths of k_mb see): share of intra / inter /
inter-with-cbp-0 macroblocks, coded tiles per macroblock, stored bits per macroblock.  GPU box only.
    python tools/mb_stats.py"""
import sys

sys.path.insert(0, ".")
import numpy as np
import torch
import m2v_load

M = m2v_load.load()
W, H, pf, n = 1920, 1152, 8, 90
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
cap = n * W * H * 3 // 2
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda:0")
enc = M.Mpeg2Encoder(7, 7, 3, 2, device=0)
enc.set_option("batch_frames", n)
nb = enc.encode_resident(clip.data_ptr(), n, d_out.data_ptr(), cap, W // 16, H // 16, pf)
mbs = (W // 16) * (H // 16)
info = enc.debug_read(0, n * mbs * 4, np.uint32).reshape(n, mbs)
bits = enc.debug_read(2, n * mbs * 4, np.uint32).reshape(n, mbs)
enc.close()
pmask = (np.arange(n) % (pf + 1)) != 0
pi, pb = info[pmask], bits[pmask]
inter = (pi & 1) == 1
cbp = (pi >> 1) & 63
ntile = np.array([bin(c).count("1") for c in range(64)])[cbp]
print("P-frame macroblocks: %d; intra %.2f %%, inter %.2f %%, inter with cbp == 0: %.3f %%" % (
    pi.size, 100 * (~inter).mean(), 100 * inter.mean(), 100 * (inter & (cbp == 0)).mean()))
print("coded tiles per inter macroblock: " + " ".join("%d:%.1f%%" % (k, 100 * (ntile[inter] == k).mean()) for k in range(7)))
print("bits per P macroblock: mean %.0f, median %.0f, p99 %.0f, max %d; > 1024 bits (overflow slot): %.3f %%" % (
    pb.mean(), np.median(pb), np.percentile(pb, 99), pb.max(), 100 * (pb > 1024).mean()))
mvx = ((pi >> 8) & 255).astype(np.uint8).view(np.int8)[inter]
mvy = ((pi >> 16) & 255).astype(np.uint8).view(np.int8)[inter]
print("vectors: zero %.1f %%, half-pel in x %.1f %%, in y %.1f %%" % (100 * ((mvx == 0) & (mvy == 0)).mean(), 100 * (mvx & 1).mean(), 100 * (mvy & 1).mean()))
print("stream: %d bytes, %.4f bit/pixel" % (nb, nb * 8 / (n * W * H)))
