#!/usr/bin/env python3
"""How long does the host take to notice that a stream has finished?  hipStreamSynchronize against spinning on a word in pinned memory that the
stream's last copy delivers.  A dependent chain: kernel (~20 us) -> 64-byte device-to-host copy -> host sees it -> next kernel; ms per 1000 links.
usage (GPU box): python tools/ubench/sync_latency.py"""
import time
import torch

dev = torch.device("cuda:0")
x = torch.zeros(1 << 22, device=dev)
flag_d = torch.zeros(16, dtype=torch.int32, device=dev)
flag_h = torch.zeros(16, dtype=torch.int32).pin_memory()
s = torch.cuda.Stream()
fh = flag_h.numpy()


def link(k, spin):
    with torch.cuda.stream(s):
        x.add_(1.0)                       # ~20 us of GPU work
        flag_d.fill_(k)
        flag_h.copy_(flag_d, non_blocking=True)
    if spin:
        while fh[0] != k:
            pass
    else:
        s.synchronize()


for spin in (False, True, False, True):
    for k in range(1, 200):
        link(k, spin)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 1000
    for k in range(1000, 1000 + n):
        link(k, spin)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%-22s %.1f us per link" % ("spin on pinned word" if spin else "stream synchronize", dt / n * 1e6))
# the GPU work alone, back to back
torch.cuda.synchronize()
t0 = time.perf_counter()
with torch.cuda.stream(s):
    for k in range(1000):
        x.add_(1.0)
        flag_d.fill_(k)
        flag_h.copy_(flag_d, non_blocking=True)
s.synchronize()
print("no host in the chain   %.1f us per link" % ((time.perf_counter() - t0) / 1000 * 1e6))
