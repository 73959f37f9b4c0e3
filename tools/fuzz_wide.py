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
    finally:
        enc.close()
    ok = got == want and got2 == want and got3 == want
    bad += not ok
    print("case %2d %4dx%-4d n=%d pf=%3d VL=%d Q=%d batch=%2d  %7d bytes  %s" % (case, W, H, n, pf, VL, Q, bf, len(want), "ok" if ok else "MISMATCH"))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
