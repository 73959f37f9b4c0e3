#!/usr/bin/env python3
"""Two strip sequences in flight on ONE rank: tools/strip_solo.py's set-up (rank r of 8 alone on a GPU, peer transport on a solo base) driven by
two host threads, a handle and a communicator stack each - while one thread sits in its host waits (the sizes, the final sync) the other's
kernels run.  ms per sequence = wall time / sequences of both threads.    usage (GPU box): python tools/strip_solo_pair.py [threads ...]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import m2v_load

M = m2v_load.load()
W = H = 2048
pf, n, world = 8, 90, 8
clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
torch.cuda.synchronize()
for nthreads in [int(x) for x in sys.argv[1:]] or [1, 2, 3]:
    for rank in (0, 4):
        sets = []
        for _ in range(nthreads):
            enc = M.Mpeg2Encoder(7, 7, 3, 2)
            base = M.StripComm.solo(world)
            comm = M.StripComm.peer(base, rank, 0)
            out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0") if rank == 0 else None
            sets.append((enc, base, comm, out))
        torch.cuda.synchronize()
        stop = [False]
        counts = [0] * nthreads

        def work(k, limit):
            enc, base, comm, out = sets[k]
            c = 0
            while (limit is None and not stop[0]) or (limit is not None and c < limit):
                M.parallel.encode_strips_native(enc, comm, rank, world, clip, 128, 128, pf, out)
                c += 1
            counts[k] = c
        # warm up
        th = [threading.Thread(target=work, args=(k, 300)) for k in range(nthreads)]
        [t.start() for t in th]; [t.join() for t in th]
        torch.cuda.synchronize()
        th = [threading.Thread(target=work, args=(k, None)) for k in range(nthreads)]
        t0 = time.perf_counter()
        [t.start() for t in th]
        time.sleep(1.5)
        stop[0] = True
        [t.join() for t in th]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps({"threads": nthreads, "rank": rank, "sequences": sum(counts), "ms_per_sequence": round(dt / sum(counts) * 1e3, 4),
                          "peer": sets[0][2].peer_stats()}), flush=True)
        for enc, base, comm, out in sets:
            comm.close(); base.close(); enc.close()
