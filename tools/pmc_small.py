#!/usr/bin/env python3
"""A small resident workload for counter passes (rocprofv3 --pmc collects per dispatch: bench.py's torch clip generator alone is thousands of
dispatches): 2 GOPs of 1920x1152 built on the CPU, uploaded once, encoded three times.  usage: python tools/pmc_small.py [cu_pack]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import m2v_load

M = m2v_load.load()
W, H, n, pf = 1920, 1152, 18, 8
clip = M.synth.clip(W, H, n, clip_index=0)
d_in = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
out = torch.empty(n * W * H * 3 // 2, dtype=torch.uint8, device="cuda:0")
torch.cuda.synchronize()
enc = M.Mpeg2Encoder(7, 7, 3, 2, device=0)
enc.set_option("split_streams", 1)
if len(sys.argv) > 1:
    enc.set_option("cu_pack", int(sys.argv[1]))
for _ in range(3):
    nb = enc.encode_resident(d_in.data_ptr(), n, out.data_ptr(), out.numel(), W // 16, H // 16, pf)
print("bytes", nb)
enc.close()
