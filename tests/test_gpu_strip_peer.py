"""-m gpu: the PEER transport of strip mode (config c5) on ONE GPU.

The macroblock kernel of a strip's edge rows stores their outer rows of the reconstruction straight into the neighbour's landing
block and counts its arrival there; a GOP step is one launch, there is no exchange step (csrc/m2v_comm.hpp PeerComm, k_mb<.., EDGE,
PEER>).  What one GPU can show: (1) ranks = threads of this process (landing blocks reached by pointer), byte-identical to the
oracle; (2) ranks = PROCESSES sharing the GPU, landing blocks reached through hipIpc handles, sizes and strips through a
caller-supplied exchange (torch.distributed / gloo behind m2v_comm_init_callbacks); (3) one rank of N alone (`solo` base): the same
bytes as the other solo transports deliver; (4) a wait that runs out of budget is not an error: the sequence is encoded again through
the base communicator, which the communicator then stays with; (5) a rank whose own work fails takes nobody down with it.
Between two GPUs the transport has never run: tests/test_gpu_multidevice.py holds that case."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_peer_threads(M, d_clip, W, H, pf, VL, world, calls=1, halo_bytes=0, debug=False, before=None, Q=2):
    """`world` handles, one host thread each, all on GPU 0, a local communicator underneath and a peer communicator per rank on
    top.  -> (stream bytes of every call, [peer_stats of every rank], [strip_last_form of every rank])"""
    import threading
    import torch
    encs = [M.Mpeg2Encoder(7, 7, VL, Q, debug=debug) for _ in range(world)]
    base = M.StripComm.local(world, debug=debug)
    clips = d_clip if isinstance(d_clip, (list, tuple)) else [d_clip]          # a list: call k encodes clips[k % len]
    out = torch.empty(M.parallel.strip_output_bound(max(int(c.shape[0]) for c in clips), W, H), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    peers, got, errs = [None] * world, [], []

    def work(r):
        try:
            peers[r] = M.StripComm.peer(base, r, 0, halo_bytes)          # collective: every thread makes the call
            for k in range(calls):
                if before is not None:
                    before(r, k, encs[r], peers[r])
                o = M.parallel.encode_strips_native(encs[r], peers[r], r, world, clips[k % len(clips)], W // 16, H // 16, pf, out if r == 0 else None)
                if r == 0:
                    got.append(o.cpu().numpy().tobytes())
        except Exception as ex:  # noqa: BLE001
            errs.append((r, ex))
    try:
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=180)
        assert not any(t.is_alive() for t in th), "a rank is stuck"
        assert not errs, errs
        return got, [p.peer_stats() for p in peers], [e.strip_last_form() for e in encs]
    finally:
        for p in peers:
            if p is not None:
                p.close()
        for e in encs:
            e.close()
        base.close()


@pytest.mark.parametrize("world,W,H,pf,VL", [(2, 128, 96, 4, 3), (3, 96, 160, 2, 2), (4, 160, 128, 3, 1), (4, 64, 64, 3, 3), (4, 64, 128, 1, 3)])
def test_peer_transport_threads_equal_oracle(world, W, H, pf, VL):
    """strips of 1 .. 4 rows (one row: the same block is the strip's first AND last row, stores both ways, waits both ways; the
    smallest frame the module accepts is 4 x 4 macroblocks, RTL:985-991); three sequences in a row on the same communicator (the
    counter sets alternate).  Whether a wait ever ran out of budget depends on how the GPU schedules the ranks' launches (four ranks'
    streams share the process's four hardware queues with torch's) - the bytes must not."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    n = 2 * (pf + 1) + 1
    clip = M.synth.clip(W, H, n, clip_index=300 + world, scene_len=4)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    got, stats, forms = run_peer_threads(M, d_clip, W, H, pf, VL, world, calls=3)
    assert all(g == want for g in got)
    assert all(s["peer_sequences"] >= 1 for s in stats), stats
    assert len({(s["peer_sequences"], s["giveups"], s["fell_back"]) for s in stats}) == 1, "the ranks disagree about what happened: %r" % (stats,)
    print("peer transport, %d ranks as threads: %r, last form %r" % (world, stats[0], forms[0]))


@pytest.mark.parametrize("world", [2, 4])
def test_peer_counter_sets_survive_sequences_of_different_gop_counts(world):
    """8 GOPs, 1 GOP, 8 GOPs, 1 GOP, 8 GOPs on ONE peer communicator: a sequence clears the arrival counters of the NEXT sequence's set,
    and that set was last used two sequences earlier - by a sequence of MORE GOPs than the one that clears it.  Every line any earlier
    sequence counted on has to go, or GOP slots 1..7 of the third sequence start at the first one's final counts, its waits pass at
    once and the edge rows read landing buffers the neighbour has not filled yet."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, VL = 128, 128, 1, 3
    long_clip = M.synth.clip(W, H, 16, clip_index=340 + world, scene_len=5)
    short_clip = M.synth.clip(W, H, 2, clip_index=350 + world)
    want = [orc.encode(c, W // 16, H // 16, pf, 7, 7, VL, 2) for c in (long_clip, short_clip)]
    d = [torch.from_numpy(np.ascontiguousarray(c)).to("cuda:0") for c in (long_clip, short_clip)]
    got, stats, forms = run_peer_threads(M, d, W, H, pf, VL, world, calls=5)
    assert [g == want[k % 2] for k, g in enumerate(got)] == [True] * 5
    assert len({(s["peer_sequences"], s["giveups"], s["fell_back"]) for s in stats}) == 1, stats
    print("alternating GOP counts, %d ranks: %r, last form %r" % (world, stats[0], forms[0]))


def run_turns(M, d_clips, W, H, pf, VL, world, turns, use_peer=True, rotate=True, Q=2, general=False):
    """`world` rank threads on GPU 0; every rank keeps TWO strip sequences in flight from its one thread: two handles taking turns
    (m2v_strip_encode_begin / _end), a peer communicator each over ONE shared local base communicator; sequence k encodes
    d_clips[k % len] and, with rotate, is assembled on rank k % world.  -> {sequence index: stream bytes}, [peer stats of rank 0's two]"""
    import threading
    import torch
    base = M.StripComm.local(world)
    cap = M.parallel.strip_output_bound(max(int(c.shape[0]) for c in d_clips), W, H)
    torch.cuda.synchronize()
    got, errs, stats = {}, [], [None]
    lock = threading.Lock()

    def work(r):
        encs, comms, outs = [], [], []
        try:
            for k in range(2):
                encs.append(M.Mpeg2Encoder(7, 7, VL, Q))
                if general:
                    encs[-1].set_option("dct_mfma", 0)            # the general form of the step (pack / unpack kernels, an exchange stream)
                comms.append(M.StripComm.peer(base, r, 0) if use_peer else base)      # collective: every thread, the same order
                outs.append(torch.empty(cap, dtype=torch.uint8, device="cuda:0"))
            busy = [None, None]

            def collect(h):
                seq, dst = busy[h]
                o = M.parallel.encode_strips_native_end(encs[h], outs[h], r, dst)
                if r == dst:
                    with lock:
                        got[seq] = o.cpu().numpy().tobytes()
                busy[h] = None
            for seq in range(turns):
                h = seq % 2
                if busy[h] is not None:
                    collect(h)
                dst = seq % world if rotate else 0
                M.parallel.encode_strips_native_begin(encs[h], comms[h], r, world, d_clips[seq % len(d_clips)], W // 16, H // 16, pf, outs[h], dst=dst)
                busy[h] = (seq, dst)
            for h in ((turns % 2), 1 - (turns % 2)):                  # oldest first
                if busy[h] is not None:
                    collect(h)
            if r == 0 and use_peer:
                stats[0] = [c.peer_stats() for c in comms]
        except Exception as ex:  # noqa: BLE001
            errs.append((r, ex))
        finally:
            if use_peer:
                for c in comms:
                    c.close()
            for e in encs:
                e.close()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    try:
        for t in th:
            t.start()
        for t in th:
            t.join(timeout=240)
        assert not any(t.is_alive() for t in th), "a rank is stuck"
        assert not errs, errs
        return got, stats[0]
    finally:
        base.close()


@pytest.mark.parametrize("world,use_peer", [(2, True), (4, True), (3, False), (1, False)])
def test_two_strip_sequences_in_flight_from_one_thread(world, use_peer):
    """m2v_strip_encode_begin / _end: nine sequences of two different clips (different GOP counts), two in flight per rank, the output rank
    rotating; every stream is the oracle's.  With the peer form every handle has its own landing block over the one base communicator;
    without it (world 3: the base communicator's own exchange inside _begin; world 1: nothing to exchange) the halves are the same."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, VL = 128, 96, 2, 3
    clips = [M.synth.clip(W, H, 9, clip_index=360 + world, scene_len=4), M.synth.clip(W, H, 4, clip_index=370 + world)]
    want = [orc.encode(c, W // 16, H // 16, pf, 7, 7, VL, 2) for c in clips]
    d = [torch.from_numpy(np.ascontiguousarray(c)).to("cuda:0") for c in clips]
    if world == 1:
        # one rank: no communicator at all
        enc = [M.Mpeg2Encoder(7, 7, VL, 2) for _ in range(2)]
        outs = [torch.empty(M.parallel.strip_output_bound(9, W, H), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
        try:
            M.parallel.encode_strips_native_begin(enc[0], None, 0, 1, d[0], W // 16, H // 16, pf, outs[0])
            M.parallel.encode_strips_native_begin(enc[1], None, 0, 1, d[1], W // 16, H // 16, pf, outs[1])
            with pytest.raises(M.M2VError):                       # busy between the halves
                enc[0].push_frames(W // 16, H // 16, pf, clips[0][:1])
            assert M.parallel.encode_strips_native_end(enc[0], outs[0], 0).cpu().numpy().tobytes() == want[0]
            assert M.parallel.encode_strips_native_end(enc[1], outs[1], 0).cpu().numpy().tobytes() == want[1]
            with pytest.raises(M.M2VError):                       # nothing in flight any more
                enc[0].strip_encode_end()
            assert enc[0].encode(clips[1], W // 16, H // 16, pf) == want[1]          # and the handle is free again
        finally:
            for e in enc:
                e.close()
        return
    got, stats = run_turns(M, d, W, H, pf, VL, world, turns=9, use_peer=use_peer)
    assert sorted(got) == list(range(9))
    assert [got[k] == want[k % 2] for k in range(9)] == [True] * 9
    print("two in flight, %d ranks, peer=%s: %r" % (world, use_peer, stats))


PEER_THREADS_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import torch
import m2v_load
from oracle import m2v_oracle_ctypes as orc
from test_gpu_strip_peer import run_peer_threads
M = m2v_load.load()
out = []
for world, W, H, pf, VL in json.loads(sys.argv[2]):
    n = 2 * (pf + 1) + 1
    clip = M.synth.clip(W, H, n, clip_index=330 + world, scene_len=4)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    got, stats, forms = run_peer_threads(M, d_clip, W, H, pf, VL, world, calls=3)
    out.append({"case": [world, W, H, pf, VL], "identical": all(g == want for g in got), "stats": stats[0], "agree": all(s == stats[0] for s in stats), "forms": forms})
print("RESULT " + json.dumps(out))
'''


def test_peer_transport_many_ranks_with_a_hardware_queue_each(tmp_path):
    """Four and eight ranks as threads in a process of its own that asks the HIP runtime for sixteen hardware queues
    (GPU_MAX_HW_QUEUES, read when the runtime starts): with a queue per rank no launch waits behind another rank's waiting blocks, and
    strips of ONE macroblock row with neighbours on both sides run in the peer form for real.  The bytes are the oracle's whatever
    the scheduling; how many sequences ran in the peer form is printed."""
    import json
    script = tmp_path / "child.py"
    script.write_text(PEER_THREADS_CHILD)
    cases = [(2, 256, 256, 4, 3), (4, 64, 64, 3, 3), (8, 64, 128, 2, 3), (8, 256, 256, 4, 3)]
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, str(script), ROOT, json.dumps(cases)], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])
    for c in res:
        print("peer transport, sixteen hardware queues:", c)
        assert c["identical"] and c["agree"], c
    # with a queue per stream the sequences really run in the peer form (how the runtime deals queues is its own business: one case
    # of the four without a single wait out of budget is what is asked for; all four is what has been seen)
    assert any(c["stats"] == {"peer_sequences": 3, "giveups": 0, "fell_back": False} and set(c["forms"]) == {"peer"} for c in res), res


def test_a_wait_out_of_budget_falls_back_to_the_base_communicator():
    """budget 0: the first wait that does not find its count at once gives up.  Rank 1 is held back before its first call so that rank 0's
    step-1 edge blocks certainly arrive first.  Every rank reads the retry mark in the size table, the sequence is encoded again
    through the local communicator - same bytes -, and the communicator stays there: call 2 runs call by call."""
    import time
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 128, 128, 3, 8
    clip = M.synth.clip(W, H, n, clip_index=311)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    os.environ["M2V_PEER_BUDGET_US"] = "0"
    try:
        def before(r, k, enc, peer):
            if r == 1 and k == 0:
                time.sleep(0.3)
        got, stats, forms = run_peer_threads(M, d_clip, W, H, pf, 3, 2, calls=2, before=before)
    finally:
        del os.environ["M2V_PEER_BUDGET_US"]
    assert got == [want, want]
    assert stats[0] == stats[1] == {"peer_sequences": 1, "giveups": 1, "fell_back": True}, stats
    assert forms == ["calls", "calls"]


def test_one_rank_of_n_alone_peer_equals_the_other_solo_transports():
    """the timing aid: rank r of N alone on the GPU, its own rows coming back as the neighbours' - by device copy, by RCCL to itself,
    or stored by the edge blocks themselves into the rank's own landing block: the same bytes"""
    import torch
    import m2v_load
    M = m2v_load.load()
    W, H, pf, n = 128, 256, 4, 12
    d_clip = torch.from_numpy(np.ascontiguousarray(M.synth.clip(W, H, n, clip_index=312, scene_len=7))).to("cuda:0")
    out = torch.zeros(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    for world, rank in ((4, 1), (4, 0), (8, 7), (16, 5)):
        res = {}
        for kind in ("copy", "peer"):
            enc = M.Mpeg2Encoder(7, 7, 3, 2)
            base = M.StripComm.solo(world)
            comm = M.StripComm.peer(base, rank, 0) if kind == "peer" else base
            try:
                enc.set_option("strip_graph", 0)
                for _ in range(3):
                    out.zero_()
                    torch.cuda.synchronize()
                    o = M.parallel.encode_strips_native(enc, comm, rank, world, d_clip, W // 16, H // 16, pf, out, dst=rank)
                    res.setdefault(kind, []).append(o.cpu().numpy().tobytes())
                if kind == "peer":
                    assert comm.peer_stats() == {"peer_sequences": 3, "giveups": 0, "fell_back": False}
                    assert enc.strip_last_form() == "peer"
            finally:
                if comm is not base:
                    comm.close()
                enc.close()
                base.close()
        assert len(set(res["copy"] + res["peer"])) == 1 and len(res["copy"][0]) > 1000, (world, rank)


def test_config_c5_geometry_peer_transport_8_ranks_as_threads(tmp_path):
    """the real size: 2048x2048, one GOP of 1 I + 8 P (and the first frames of a second one), 8 ranks x 16 macroblock rows as threads
    on one GPU, in a process that asks for a hardware queue per rank - with the default four, eight ranks' launches wait behind each
    other's waiting blocks, a wait runs out of budget and the sequence falls back (legitimately: the stream is the oracle's either
    way, which the second half of this test checks in this process)."""
    import json
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    script = tmp_path / "child.py"
    script.write_text(PEER_THREADS_CHILD.replace("n = 2 * (pf + 1) + 1", "n = pf + 3"))
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, str(script), ROOT, json.dumps([(8, 2048, 2048, 8, 3)])], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])[0]
    print("c5, 8 ranks as threads, sixteen hardware queues:", res)
    assert res["identical"] and res["agree"]
    M = m2v_load.load()
    W = H = 2048
    pf, n = 8, 9
    d_clip = M.synth.clip_torch(W, H, n, clip_index=58, device="cuda:0", scene_len=5)
    want = orc.encode(d_clip.cpu().numpy(), 128, 128, pf, 7, 7, 3, 2)
    got, stats, forms = run_peer_threads(M, d_clip, W, H, pf, 3, 8, calls=2)
    assert got == [want, want]
    print("c5, 8 ranks as threads, the process's default queues:", stats[0], forms)


def test_a_rank_whose_own_work_fails_with_the_peer_transport():
    """rank 1 of three fails locally (injected, -DM2V_DEBUG library): its kernels never run, the neighbours' waits run out of budget
    (1 ms here), rank 1 marks its sizes, every rank returns an error from the same call - nobody hangs; a sequence afterwards works."""
    import ctypes
    import threading
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 96, 96, 2, 6
    clip = M.synth.clip(W, H, n, clip_index=313)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    encs = [M.Mpeg2Encoder(7, 7, 3, 2, debug=True) for _ in range(3)]
    L = encs[0]._L
    os.environ["M2V_PEER_BUDGET_US"] = "1000"
    base = M.StripComm.local(3, debug=True)
    peers = [None] * 3
    try:
        def run(bad):
            rc, msg, nbs = [None] * 3, [None] * 3, [0]

            def work(r):
                if peers[r] is None:
                    peers[r] = M.StripComm.peer(base, r, 0)
                nb = ctypes.c_size_t(0)
                rc[r] = L.m2v_strip_encode(encs[r]._h, peers[r].handle, r, 3, 0, W // 16, H // 16, pf, d_clip.data_ptr(), n,
                                           out.data_ptr() if r == 0 else None, out.numel() if r == 0 else 0, ctypes.byref(nb), None)
                msg[r] = L.m2v_last_error(encs[r]._h)
                if r == 0:
                    nbs[0] = nb.value
            if bad is not None:
                encs[bad].set_option("ablate", 1 << 21)
            th = [threading.Thread(target=work, args=(r,)) for r in range(3)]
            for t in th:
                t.start()
            for t in th:
                t.join(timeout=60)
            assert not any(t.is_alive() for t in th), "a rank is still waiting"
            if bad is not None:
                encs[bad].set_option("ablate", 0)
            return rc, msg, nbs[0]
        rc, msg, _ = run(1)
        assert rc[1] < 0 and b"injected failure" in msg[1]
        assert rc[0] < 0 and rc[2] < 0, (rc, msg)
    finally:
        del os.environ["M2V_PEER_BUDGET_US"]
        for p in peers:
            if p is not None:
                p.close()
        base.close()
    # the local communicator was aborted by the failing call; a fresh pair works with the same handles
    base = M.StripComm.local(3, debug=True)
    peers = [None] * 3
    try:
        rc, msg, nb = run(None)
        assert rc == [0, 0, 0], msg
        assert out[:nb].cpu().numpy().tobytes() == want
    finally:
        for p in peers:
            if p is not None:
                p.close()
        base.close()
        for e in encs:
            e.close()


# ---- ranks = processes sharing the GPU: landing blocks through hipIpc handles, everything else through the caller's exchange ----
CHILD = r'''
import os, sys, ctypes
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, port, outfile = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
import torch
import torch.distributed as dist
import m2v_load
M = m2v_load.load()
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", port
dist.init_process_group("gloo", rank=rank, world_size=world)
# the HIP runtime this process already holds (torch's copy): the same file again gives the same instance, never a second runtime
hip = ctypes.CDLL([ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln][0])
hip.hipStreamSynchronize.argtypes = [ctypes.c_void_p]
hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
D2H, H2D = 2, 1

def down(ptr, n):                               # device -> numpy
    a = np.empty(n, np.uint8)
    assert hip.hipMemcpy(a.ctypes.data, ptr, n, D2H) == 0
    return a

def up(ptr, a):
    a = np.ascontiguousarray(a)
    assert hip.hipMemcpy(ptr, a.ctypes.data, a.nbytes, H2D) == 0

def halo(r, su, ru, sd, rd, n, stream):         # only used if the peer form falls back
    hip.hipStreamSynchronize(stream)
    ops, back = [], []
    for s_, r_, peer in ((su, ru, r - 1), (sd, rd, r + 1)):
        if s_:
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(down(s_, n)), peer))
            t = torch.empty(n, dtype=torch.uint8)
            back.append((r_, t))
            ops.append(dist.P2POp(dist.irecv, t, peer))
    for q in dist.batch_isend_irecv(ops):
        q.wait()
    for p_, t in back:
        up(p_, t.numpy())
    return 0

def allgather(r, src, dst, count, stream):
    hip.hipStreamSynchronize(stream)
    mine = torch.from_numpy(down(src, 8 * count))
    every = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    up(dst, torch.cat(every).numpy())
    return 0

def gather(r, dstrank, strip, sizes, bufs, stream):
    hip.hipStreamSynchronize(stream)
    if r != dstrank:
        if sizes[r]:
            dist.send(torch.from_numpy(down(strip, sizes[r])), dstrank)
    else:
        for k in range(world):
            if k != dstrank and sizes[k]:
                t = torch.empty(sizes[k], dtype=torch.uint8)
                dist.recv(t, k)
                up(bufs[k], t.numpy())
    return 0

W, H, pf, n = 192, 128, 4, 12
clip = M.synth.clip(W, H, n, clip_index=320, scene_len=5)
d_clip = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
out = torch.empty(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0") if rank == 0 else None
torch.cuda.synchronize()
enc = M.Mpeg2Encoder(7, 7, 3, 2)
base = M.StripComm.callbacks(world, halo, allgather, gather)
peer = M.StripComm.peer(base, rank, 0, connect=False)
# descriptors by the test's own means (here: gloo), then m2v_comm_peer_connect - what a caller without a base all-gather would do
mine = torch.frombuffer(bytearray(peer.peer_export()), dtype=torch.uint8)
every = [torch.empty_like(mine) for _ in range(world)]
dist.all_gather(every, mine)
peer.peer_connect(bytes(every[rank - 1].numpy()) if rank > 0 else None, bytes(every[rank + 1].numpy()) if rank < world - 1 else None)
res = []
for k in range(3):
    o = M.parallel.encode_strips_native(enc, peer, rank, world, d_clip, W // 16, H // 16, pf, out)
    if rank == 0:
        res.append(o.cpu().numpy().tobytes())
st = peer.peer_stats()
form = enc.strip_last_form()
dist.barrier()                                   # nobody frees a landing block a neighbour may still be storing into
peer.close(); base.close(); enc.close()
if rank == 0:
    from oracle import m2v_oracle_ctypes as orc
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    open(outfile, "w").write(repr({"identical": [r == want for r in res], "stats": st, "form": form}))
dist.barrier()
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world", [2, 3])
def test_peer_transport_between_processes_through_ipc_handles(world, tmp_path):
    """`world` PROCESSES sharing GPU 0: every rank's landing block is mapped into its neighbours' address space with
    hipIpcOpenMemHandle (descriptors exchanged by the test over gloo), the edge blocks of one process store into the memory of
    another and count their arrival there; sizes and strips go through m2v_comm_init_callbacks (gloo again).  Byte-identical to the
    oracle, three sequences in a row."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    script, outfile = tmp_path / "child.py", tmp_path / "result.txt"
    script.write_text(CHILD)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    ps = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), str(world), port, str(outfile)], env=env, cwd=ROOT) for r in range(world)]
    try:
        rcs = [p.wait(timeout=300) for p in ps]
    finally:
        for p in ps:
            if p.poll() is None:
                p.kill()
    assert rcs == [0] * world
    res = eval(outfile.read_text())
    assert res["identical"] == [True, True, True], res
    assert res["stats"]["peer_sequences"] >= 1
    print("peer transport between %d processes (hipIpc):" % world, res)
