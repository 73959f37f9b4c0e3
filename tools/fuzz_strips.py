#!/usr/bin/env python3
"""Ad-hoc differential run of strip mode (config c5's path) against the oracle: random frame sizes, 1 - 8 ranks (threads on
one GPU, the in-process communicator) with strips down to ONE macroblock row, every VECTOR_LEVEL / Q_LEVEL, GOP lengths
from intra-only to 255, sequences that end inside a GOP, every content kind of tests/test_gpu_fuzz.py; the fused edge-row
kernel and the general form (pack / unpack kernels) of the step.
usage (GPU box): python tools/fuzz_strips.py [cases] [seed] [peer] [turns]
"peer": every case with two or more ranks also through the peer transport (landing blocks, arrival counters; a hardware queue per rank is
asked for so that the ranks' launches do not wait behind each other's waiting blocks) - the number of sequences that really ran in the
peer form is printed at the end.
"turns": every case also with TWO sequences in flight per rank from one thread (m2v_strip_encode_begin / _end on two handles, the case's clip and
its first frames alone taking turns - different GOP counts on one communicator -, the output rank rotating), in the peer form when "peer" is
given too, else through the in-process communicator."""
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
from test_gpu_strip_peer import run_peer_threads, run_turns

with_peer = "peer" in sys.argv[3:]
with_turns = "turns" in sys.argv[3:]
turn_runs = 0
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
    if with_turns and ok:
        try:
            short = clip[:max(1, n // 2)]
            want2 = [want, orc.encode(short, W // 16, H // 16, pf, 7, 7, VL, Q)]
            d2 = [d_clip, torch.from_numpy(np.ascontiguousarray(short)).to("cuda:0")]
            if world == 1:
                encs = [G.M.Mpeg2Encoder(7, 7, VL, Q) for _ in range(2)]
                outs = [torch.empty(G.M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
                try:
                    for k in range(2):
                        G.M.parallel.encode_strips_native_begin(encs[k], None, 0, 1, d2[k], W // 16, H // 16, pf, outs[k])
                    tg = {k: G.M.parallel.encode_strips_native_end(encs[k], outs[k], 0).cpu().numpy().tobytes() for k in range(2)}
                finally:
                    for e_ in encs:
                        e_.close()
            else:
                tg, _ = run_turns(G.M, d2, W, H, pf, VL, world, turns=5, use_peer=with_peer and not general, Q=Q, general=general)
            ok = all(tg[k] == want2[k % 2] for k in tg) and len(tg) == (2 if world == 1 else 5)
            turn_runs += 1
            if not ok:
                print("    two sequences IN FLIGHT differ", sorted(tg), [tg[k] == want2[k % 2] for k in sorted(tg)])
        except Exception as ex:  # noqa: BLE001
            ok = False
            print("   turns:", repr(ex)[:300])
    bad += not ok
    print("case %2d %4dx%-4d ranks=%d n=%d pf=%3d VL=%d Q=%d %s  %7d bytes  %s" % (case, W, H, world, n, pf, VL, Q, "general" if general else "fused  ", len(want), "ok" if ok else "MISMATCH"), flush=True)
if with_peer:
    print("peer transport: %d cases x 2 sequences, %d sequences ran in the peer form, %d fell back" % (peer_runs, peer_form, peer_fallbacks))
if with_turns:
    print("two sequences in flight (begin / end): %d cases" % turn_runs)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
