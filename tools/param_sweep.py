#!/usr/bin/env python3
"""Throughput of the resident path across the module's static parameters (RTL:11-14): VECTOR_LEVEL 1 .. 3 (search range +-2 / 4 / 6) x Q_LEVEL 1 .. 4 on the
benchmark's clip shape (1920x1152, 10 GOPs of 1 I + 8 P), two handles in flight as bench.py times config c3; every stream's first GOP against the oracle.
    usage (GPU box): python tools/param_sweep.py [steps]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()
W, H, pf, gops = 1920, 1152, 8, 10
n = gops * (pf + 1)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
first_gop = clip[:pf + 1].cpu().numpy()
cap = n * W * H * 3 // 2
outs = [torch.empty(cap, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
torch.cuda.synchronize()
for VL in (1, 2, 3):
    for Q in (1, 2, 3, 4):
        encs = [M.Mpeg2Encoder(7, 7, VL, Q) for _ in range(2)]
        try:
            for e in encs:
                e.set_option("batch_frames", n)
                e.set_option("split_streams", 1)

            def run(k):
                busy, nb = [False, False], 0
                for i in range(k):
                    h = i % 2
                    if busy[h]:
                        nb = encs[h].encode_resident_end()
                    encs[h].encode_resident_begin(clip.data_ptr(), n, outs[h].data_ptr(), cap, W // 16, H // 16, pf, 0)
                    busy[h] = True
                for h in range(2):
                    if busy[h]:
                        nb = encs[h].encode_resident_end()
                return nb
            run(30)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            nb = run(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            got = outs[(steps - 1) % 2][:nb].cpu().numpy().tobytes()
            ref = orc.encode(first_gop, W // 16, H // 16, pf, 7, 7, VL, Q)
            cut = got.find(b"\x00\x00\x01\xb8", 8 + 1)          # the second GOP header: everything before it is the first GOP (+ sequence headers)
            second = got.find(b"\x00\x00\x01\xb8", got.find(b"\x00\x00\x01\xb8") + 4)
            ref_end = ref.rfind(b"\x00\x00\x01\xb7")
            ok = got[:second] == ref[:ref_end]
            print(json.dumps({"VECTOR_LEVEL": VL, "Q_LEVEL": Q, "MPixels_per_s": round(steps * n * W * H / dt * 1e-6, 1), "ms_per_sequence": round(dt / steps * 1e3, 4),
                              "bits_per_pixel": round(nb * 8 / (n * W * H), 4), "first_gop_identical_to_oracle": ok}), flush=True)
        finally:
            for e in encs:
                e.close()
