#!/usr/bin/env python3
"""Two sequences in flight on two handles: does the second handle's stream priority decide whether they overlap?  Run under different
GPU_MAX_HW_QUEUES to provoke the two streams sharing one hardware queue.   python tools/queue_ab.py <priority of handle 2: 0 | 1 | -1>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m2v_load

M = m2v_load.load()
prio = int(sys.argv[1]) if len(sys.argv) > 1 else 0
W, H, n, pf = 1920, 1152, 90, 8
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
cap = n * W * H * 3 // 2
outs = [torch.empty(cap, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
encs = [M.Mpeg2Encoder(7, 7, 3, 2, device=0) for _ in range(2)]
for e in encs:
    e.set_option("batch_frames", n)
    e.set_option("split_streams", 1)
if prio:
    encs[1].set_option("stream_priority", prio)


def fly(k):
    busy = [False, False]
    for i in range(k):
        h = i % 2
        if busy[h]:
            encs[h].encode_resident_end()
        encs[h].encode_resident_begin(clip.data_ptr(), n, outs[h].data_ptr(), cap, W // 16, H // 16, pf)
        busy[h] = True
    for h in range(2):
        if busy[h]:
            encs[h].encode_resident_end()


def block(k):
    for _ in range(k):
        encs[0].encode_resident(clip.data_ptr(), n, outs[0].data_ptr(), cap, W // 16, H // 16, pf)


res = {}
for name, fn in (("in flight", fly), ("blocking", block)):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 1.0:
        fn(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(80)
    torch.cuda.synchronize()
    res[name] = (time.perf_counter() - t0) / 80 * 1e3
print("GPU_MAX_HW_QUEUES=%s priority of handle 2 = %2d:  in flight %.3f ms  blocking (one stream) %.3f ms  -> overlap gain %.1f %%"
      % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), prio, res["in flight"], res["blocking"], 100 * (res["blocking"] / res["in flight"] - 1)))
