#!/usr/bin/env python3
"""What ONE rank of an N-GPU strip job (config c5: 2048x2048, 128 macroblock rows) does, timed on one real GPU.

The rank runs the native loop (m2v_strip_encode) over its 1/N strip with a `solo` communicator: the halo it receives is a
device copy of its own rows, the sizes it "gathers" are its own.  The stream is NOT valid; the timing is what matters:
kernel time of a 1/N strip per GOP step (are 20 480 wavefronts enough to fill 256 CUs?), launch gaps, host time per step,
the serial part on the output rank (sizes, gather, final assembly).  Against the same loop over the whole frame on the same
GPU this bounds the strong-scaling curve from the compute side (the xGMI latency of 18 KB per frame is not in it).

    python tools/strip_solo.py [--world 8] [--gops 10] [--steps 20]      -> one JSON line per rank position tried"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--gops", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--rccl", type=int, default=0, help="1: the rows go through a 1-rank RCCL communicator (ncclSend / ncclRecv to itself) instead of device copies")
    ap.add_argument("--peer", type=int, default=0, help="1: the peer transport on top (the edge blocks store their rows into the rank's own landing block and "
                                                     "count their arrival there: one launch per GOP step, no exchange step)")
    ap.add_argument("--ablate", type=int, default=0, help="-DM2V_DEBUG library, option ablate (results invalid): bit 22 plain halo stores, 23 no arrival count, 24 no wait, 25 plain window loads")
    ap.add_argument("--graph", type=int, nargs="+", default=[0, 1], help="option strip_graph: 0 = the sequence call by call, 1 = one recorded hipGraph launch")
    ap.add_argument("--split", type=int, default=-1, help="option split_streams of the handle (GOP groups on a stream each; default: the library's)")
    args = ap.parse_args()
    import torch
    import m2v_load
    M = m2v_load.load()
    W = H = 2048
    pf = 8
    n = args.gops * (pf + 1)
    clip = M.synth.clip_torch(W, H, n, clip_index=0, device="cuda:0")
    out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    base = None
    for world, graph in [(w, gph) for w in args.world for gph in args.graph]:
        for rank in sorted({0, world // 2}):                      # the output rank (one neighbour + final assembly) and an inner rank
            enc = M.Mpeg2Encoder(7, 7, 3, 2, debug=bool(args.ablate))
            if args.ablate:
                enc.set_option("ablate", args.ablate)
            enc.set_option("strip_graph", graph)
            if args.split >= 0:
                enc.set_option("split_streams", args.split)
            cbase = M.StripComm.solo(world, rccl=bool(args.rccl), debug=bool(args.ablate)) if world > 1 else None
            comm = M.StripComm.peer(cbase, rank, 0) if (cbase is not None and args.peer) else cbase
            try:
                run = lambda: M.parallel.encode_strips_native(enc, comm, rank, world, clip, 128, 128, pf, out if rank == 0 else None)   # noqa: E731
                t0 = time.perf_counter()
                while time.perf_counter() - t0 < 1.0:
                    run()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    run()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / args.steps
                host_timed = enc.strip_stats()["host_us_per_step"]
                gst = enc.strip_graph_stats()
                enc.set_option("profile", 1)
                run(); run()
                st = enc.strip_stats()
                ks = {name: round(enc.kernel_stats(k)[1], 3) for k, name in ((0, "k_mb_P"), (1, "k_mb_I"), (4, "scans"), (3, "k_assemble"), (2, "final_assembly"))}
                if world == 1 and base is None:
                    base = dt
                print(json.dumps({"world": world, "rank": rank, "transport": comm.kind if comm is not None else None,
                                  "form": enc.strip_last_form(), "peer": comm.peer_stats() if (comm is not None and args.peer) else None,
                                  "strip_graph": graph, "graph_launches": gst["launches"], "graph_broken": gst["broken"],
                                  "host_us_per_gop_step_timed": round(host_timed, 1), "split_streams": args.split if args.split >= 0 else "default", "ms_per_sequence": round(dt * 1e3, 3),
                                  "speedup_vs_one_rank": round(base / dt, 2) if base else None,
                                  "ideal": world, "host_us_per_gop_step_profiled_call_by_call": round(st["host_us_per_step"], 1),
                                  "of_which_inside_the_communicator": round(st["comm_us_per_step"], 1),
                                  "halo_ms": {"total": round(st["halo_total"], 3), "exposed": round(st["halo_exposed"], 3)},
                                  "sizes_gather_assembly_ms": round(st["gather"], 3), "kernel_ms": ks}))
                sys.stdout.flush()
            finally:
                enc.close()
                if comm is not None and comm is not cbase:
                    comm.close()
                if cbase is not None:
                    cbase.close()


if __name__ == "__main__":
    main()
