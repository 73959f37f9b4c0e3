#!/usr/bin/env python3
"""Ad-hoc concurrency run: several host threads at once on ONE GPU, each with handles of its own - blocking resident calls, two
handles taking turns with m2v_encode_resident_begin / _end, the port path (beats; frames from page-locked memory), and strip mode with an in-process
communicator (whose ranks are further threads) - created, used and closed while the others run.  Every stream is checked against
the oracle (computed beforehand, single-threaded).  usage (GPU box): python tools/stress_threads.py [threads] [rounds] [seed]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import gpu_util as G
from oracle import m2v_oracle_ctypes as orc
from test_gpu_strips import run_native_strips
from test_gpu_strip_peer import run_turns

M = G.M
T = int(sys.argv[1]) if len(sys.argv) > 1 else 6
R = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 5)

cases = []
for _ in range(10):
    W, H = 16 * int(rng.integers(4, 24)), 16 * int(rng.integers(4, 12))
    n, pf = int(rng.integers(2, 12)), int(rng.choice([0, 2, 8]))
    VL, Q = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    clip = M.synth.clip(W, H, n, clip_index=int(rng.integers(0, 1000)), scene_len=int(rng.integers(3, 9)))
    cases.append((W, H, n, pf, VL, Q, clip, orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q)))
torch.cuda.synchronize()
bad, done, lock = [], [0], threading.Lock()


def resident(encs, c):
    W, H, n, pf, VL, Q, clip, want = c
    d = torch.from_numpy(np.ascontiguousarray(clip)).cuda()
    outs = [torch.empty(n * W * H * 3 + 65536, dtype=torch.uint8, device="cuda") for _ in encs]
    torch.cuda.synchronize()
    got = []
    for h, (e, o) in enumerate(zip(encs, outs)):
        e.encode_resident_begin(d.data_ptr(), n, o.data_ptr(), o.numel(), W // 16, H // 16, pf, 0)
    for e, o in zip(encs, outs):
        nb = e.encode_resident_end()
        got.append(o[:nb].cpu().numpy().tobytes())
    return all(g == want for g in got)


def worker(t):
    r = np.random.default_rng(1000 + t)
    for it in range(R):
        c = cases[int(r.integers(0, len(cases)))]
        W, H, n, pf, VL, Q, clip, want = c
        kind = int(r.integers(0, 7))
        t_begin = time.perf_counter()
        try:
            if kind == 0:
                ok = G.resident_encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, batch_frames=int(r.choice([2, 96]))) == want
            elif kind == 1:
                encs = [M.Mpeg2Encoder(7, 7, VL, Q) for _ in range(2)]
                try:
                    for e in encs:
                        e.set_option("split_streams", int(r.integers(1, 4)))
                    ok = resident(encs, c) and resident(encs, c)
                finally:
                    for e in encs:
                        e.close()
            elif kind == 2:
                e = M.Mpeg2Encoder(7, 7, VL, Q)
                try:
                    e.set_option("batch_frames", int(r.choice([1, 3, 96])))
                    ok = e.encode(clip, W // 16, H // 16, pf) == want
                finally:
                    e.close()
            elif kind == 4:
                # the port path from page-locked memory: blocking m2v_push_frames or m2v_push_frames_pull, chunk and call sizes at random, pulls or none in between
                e = M.Mpeg2Encoder(7, 7, VL, Q)
                try:
                    src = torch.from_numpy(np.ascontiguousarray(clip)).pin_memory().numpy()
                    e.set_option("batch_frames", int(r.choice([1, 2, 5, 96])))
                    per, mode = int(r.choice([1, 2, 3, 7])), int(r.integers(0, 3))
                    out = np.zeros(len(want) + 4096, np.uint8)
                    pos, last = 0, False
                    for k in range(0, n, per):
                        if mode == 2:
                            pos += e.push_frames_pull(W // 16, H // 16, pf, src[k:k + per], out, pos)[0]
                        else:
                            e.push_frames(W // 16, H // 16, pf, src[k:k + per])
                            if mode == 1:
                                pos += e.pull_into(out, pos)[0]
                    e.sequence_stop()
                    while not last:
                        m, last = e.pull_into(out, pos)
                        pos += m
                    ok = out[:pos].tobytes() == want
                finally:
                    e.close()
            elif kind == 5:
                # the beat entry points: packed samples / three arrays, page-locked or not, calls that line up with nothing (round 6: packed bytes
                # are de-interleaved on the device, whole frames of beats upload as strided copies)
                e = M.Mpeg2Encoder(7, 7, VL, Q)
                try:
                    e.set_option("batch_frames", int(r.choice([1, 2, 5, 96])))
                    layout = ["yuv24", "uyv24", "yuvx32", "ayuv32", None][int(r.integers(0, 5))]
                    pin = bool(r.integers(0, 2))
                    y, u, v = clip[:, 0], clip[:, 1], clip[:, 2]
                    if layout is None:
                        arrs = [np.ascontiguousarray(a).reshape(-1) for a in (y, u, v)]
                        if pin:
                            arrs = [torch.from_numpy(a).pin_memory().numpy() for a in arrs]
                        bpp = 1
                    else:
                        pad = np.full_like(y, 0x11)
                        order = {"yuv24": (y, u, v), "uyv24": (u, y, v), "yuvx32": (y, u, v, pad), "ayuv32": (pad, y, u, v)}[layout]
                        pk = np.ascontiguousarray(np.stack(order, axis=-1)).reshape(-1)
                        pk = torch.from_numpy(pk).pin_memory().numpy() if pin else pk
                        bpp = len(order)
                    beats, b, bpf = n * W * H // 4, 0, W * H // 4
                    while b < beats:
                        take = int(min(beats - b, r.choice([7, bpf // 2 + 1, bpf, 2 * bpf, 3 * bpf + 5])))
                        stop = b + take == beats
                        if layout is None:
                            e.push_beats(W // 16, H // 16, pf, arrs[0][4 * b:4 * (b + take)], arrs[1][4 * b:4 * (b + take)], arrs[2][4 * b:4 * (b + take)], stop_with_last=stop)
                        else:
                            e.push_packed(W // 16, H // 16, pf, pk[4 * bpp * b:4 * bpp * (b + take)], layout, stop_with_last=stop)
                        b += take
                    ok = e.pull_all() == want
                finally:
                    e.close()
            elif kind == 6:
                # strip sequences in flight from one thread per rank (m2v_strip_encode_begin / _end), the ranks further threads, peer form or not
                world = int(r.integers(2, min(4, H // 16) + 1))
                d = torch.from_numpy(np.ascontiguousarray(clip)).cuda()
                torch.cuda.synchronize()
                got, _ = run_turns(M, [d], W, H, pf, VL, world, turns=4, use_peer=bool(r.integers(0, 2)), Q=Q)
                ok = len(got) == 4 and all(g == want for g in got.values())
            else:
                world = int(r.integers(1, min(5, H // 16) + 1))
                d = torch.from_numpy(np.ascontiguousarray(clip)).cuda()
                torch.cuda.synchronize()
                ok = run_native_strips(M, d, W, H, pf, VL, world, general=bool(r.integers(0, 2)), Q=Q)[0] == want
        except Exception as ex:  # noqa: BLE001
            ok = False
            print("thread %d round %d kind %d: %r" % (t, it, kind, ex), flush=True)
        if time.perf_counter() - t_begin > 3.0:
            print("thread %d round %d kind %d took %.1f s" % (t, it, kind, time.perf_counter() - t_begin), flush=True)
        with lock:
            done[0] += 1
            if not ok:
                bad.append((t, it, kind, W, H, n, pf, VL, Q))


th = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
for x in th:
    x.start()
for x in th:
    x.join(timeout=900)
stuck = [i for i, x in enumerate(th) if x.is_alive()]
print("sequences checked: %d, mismatches: %d %s, stuck threads: %s" % (done[0], len(bad), bad[:8], stuck), flush=True)
os._exit(1 if bad or stuck else 0)
