#!/usr/bin/env python3
"""Same-box, same-process A/B of the stream assembly behind the last macroblock kernel: option fused_tail = 0 (k_slice_scan + k_frame_scan +
k_assemble, three launches) against 1 (k_assemble<true>: the scans inside, decoupled look-back over the slices, one launch).  Blocking
m2v_encode_resident calls on config c3's clip, alternating blocks of 200 calls; then one rank of 8 of config c5 (tools/strip_solo.py's
set-up, peer transport).    usage (GPU box): python tools/tail_ab.py [rounds]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import m2v_load

M = m2v_load.load()
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
W, H, pf, n = 1920, 1152, 8, 90
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
cap = n * W * H * 3 // 2
outs = [torch.empty(cap, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
encs = []
for fused in (0, 1):
    e = M.Mpeg2Encoder(7, 7, 3, 2)
    e.set_option("batch_frames", n)
    e.set_option("fused_tail", fused)
    encs.append(e)
torch.cuda.synchronize()
nb = [0, 0]
for k in range(2):
    for _ in range(300):
        nb[k] = encs[k].encode_resident(clip.data_ptr(), n, outs[k].data_ptr(), cap, W // 16, H // 16, pf)
assert nb[0] == nb[1] and torch.equal(outs[0][:nb[0]], outs[1][:nb[1]]), "the two forms of the tail differ"
for r in range(rounds):
    line = []
    for k in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            encs[k].encode_resident(clip.data_ptr(), n, outs[k].data_ptr(), cap, W // 16, H // 16, pf)
        torch.cuda.synchronize()
        line.append((time.perf_counter() - t0) / 200 * 1e3)
    print("c3 blocking, ms per sequence: three launches %.4f   fused %.4f   (%+.2f %%)" % (line[0], line[1], (line[1] / line[0] - 1) * 100), flush=True)
for k in range(2):
    encs[k].set_option("profile", 1)
    for _ in range(3):
        encs[k].encode_resident(clip.data_ptr(), n, outs[k].data_ptr(), cap, W // 16, H // 16, pf)
    print("fused_tail %d: kernel ms per sequence (profiled, one stream): scans %.4f  assemble %.4f" % (k, encs[k].kernel_stats(4)[1], encs[k].kernel_stats(3)[1]))
    encs[k].close()
# one rank of 8 of config c5, peer transport
Ws = Hs = 2048
clip = M.synth.clip_torch(Ws, Hs, n, clip_index=0, device="cuda:0")
out = torch.empty(M.parallel.strip_output_bound(n, Ws, Hs), dtype=torch.uint8, device="cuda:0")
setups = []
for fused in (0, 1):
    e = M.Mpeg2Encoder(7, 7, 3, 2)
    e.set_option("fused_tail", fused)
    base = M.StripComm.solo(8)
    comm = M.StripComm.peer(base, 4, 0)
    setups.append((e, base, comm))
torch.cuda.synchronize()
for e, base, comm in setups:
    for _ in range(300):
        M.parallel.encode_strips_native(e, comm, 4, 8, clip, 128, 128, pf, None)
for r in range(rounds):
    line = []
    for e, base, comm in setups:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            M.parallel.encode_strips_native(e, comm, 4, 8, clip, 128, 128, pf, None)
        torch.cuda.synchronize()
        line.append((time.perf_counter() - t0) / 200 * 1e3)
    print("c5, inner rank of 8 (peer form), ms per sequence: three launches %.4f   fused %.4f   (%+.2f %%)" % (line[0], line[1], (line[1] / line[0] - 1) * 100), flush=True)
for e, base, comm in setups:
    comm.close(); base.close(); e.close()
