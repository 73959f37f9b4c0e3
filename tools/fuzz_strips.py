#!/usr/bin/env python3
"""Ad-hoc differential run of strip mode (config c5's path) against the oracle: random frame sizes, 1 - 8 ranks (threads on
one GPU, the in-process communicator) with strips down to ONE macroblock row, every VECTOR_LEVEL / Q_LEVEL, GOP lengths
from intra-only to 255, sequences that end inside a GOP, every content kind of tests/test_gpu_fuzz.py; the fused edge-row
kernel and the general form (pack / unpack kernels) of the step.
usage (GPU box): python tools/fuzz_strips.py [cases] [seed] [peer]
"peer": every case with two or more ranks also through the peer transport (landing blocks, arrival counters; a hardware queue per rank is
asked for so that the ranks' launches do not wait behind each other's waiting blocks) - the number of sequences that really ran in the
peer form is printed at the end."""
import os
import sys

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")        # read when the HIP runtime starts

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import torch

import gpu_util as G
from oracle import m2v_oracle_ctypes as orc
from test_gpu_fuzz import make_content
from test_gpu_strips import run_native_strips
from test_gpu_strip_peer import run_peer_threads

with_peer = len(sys.argv) > 3 and sys.argv[3] == "peer"
peer_runs = peer_form = peer_fallbacks = 0

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
bad = 0
for case in range(n_cases):
    world = int(rng.integers(1, 9))
    rows = int(rng.integers(max(4, world), max(4, world) + 12))          # at least one macroblock row per rank
    W, H = 16 * int(rng.integers(4, 41)), 16 * rows
    VL, Q = int(rng.integers(1, 4)), int(rng.integers(1, 5))
    pf = int(rng.choice([0, 1, 3, 8, 255]))
    n = int(rng.integers(1, 8))
    general = bool(rng.integers(0, 2))
    clip = make_content(rng, G.M, W, H, n)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    try:
        got, _ = run_native_strips(G.M, d_clip, W, H, pf, VL, world, general=general, Q=Q)
        ok = got == want
    except Exception as ex:  # noqa: BLE001
        got, ok = b"", False
        print("   ", repr(ex)[:300])
    if with_peer and world > 1 and ok:
        try:
            pg, pstats, forms = run_peer_threads(G.M, d_clip, W, H, pf, VL, world, calls=2, Q=Q)
            ok = all(g == want for g in pg)
            peer_runs += 1
            peer_form += pstats[0]["peer_sequences"]
            peer_fallbacks += pstats[0]["giveups"]
            if not ok:
                print("    the PEER transport differs", pstats[0], forms)
        except Exception as ex:  # noqa: BLE001
            ok = False
            print("   peer:", repr(ex)[:300])
    bad += not ok
    print("case %2d %4dx%-4d ranks=%d n=%d pf=%3d VL=%d Q=%d %s  %7d bytes  %s" % (case, W, H, world, n, pf, VL, Q, "general" if general else "fused  ", len(want), "ok" if ok else "MISMATCH"), flush=True)
if with_peer:
    print("peer transport: %d cases x 2 sequences, %d sequences ran in the peer form, %d fell back" % (peer_runs, peer_form, peer_fallbacks))
print("mismatches:", bad)
sys.exit(1 if bad else 0)
