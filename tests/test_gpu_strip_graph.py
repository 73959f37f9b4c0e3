"""-m gpu: the strip sequence of m2v_strip_encode as a recorded hipGraph (option "strip_graph"), proved on ONE GPU.

A call enqueues - per GOP step - the edge-row launch, the exchange, the interior rows, the events that order them; then the
strip's slices and the all-gather of the sizes.  From the second call of a shape on that is ONE graph launch.  What can be checked
with one GPU: (1) world = 1 (a valid stream): recorded == call by call == oracle, also when the input pointer changes between
the launches; (2) one rank of N with the `solo` communicator (device copies in place of the neighbours; NOT a valid stream, but
deterministic): recorded == call by call, byte for byte; (3) the same with RCCL's own send / recv kernels inside the recording
(a 1-rank communicator, ncclSend / ncclRecv to itself) - so RCCL-in-capture has run before an 8-GPU node ever sees it."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_world1_recorded_sequence_equals_oracle_and_follows_the_input_pointer():
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 160, 96, 3, 9
    clips = [M.synth.clip(W, H, n, clip_index=200 + k, scene_len=5) for k in range(2)]
    wants = [orc.encode(c, W // 16, H // 16, pf, 7, 7, 3, 2) for c in clips]
    d_clips = [torch.from_numpy(np.ascontiguousarray(c)).to("cuda:0") for c in clips]
    out = torch.zeros(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    try:
        # call 1 enqueues call by call (every buffer a recording would reference is allocated BEFORE the shape is noted, the output
        # rank's assembly tables included), call 2 - the shape comes a second time - records and launches, 3.. launch; the clip
        # (another buffer) changes
        order = [0, 0, 0, 0, 1, 0, 1]
        for i, k in enumerate(order):
            out.zero_()
            got = M.parallel.encode_strips_native(enc, None, 0, 1, d_clips[k], W // 16, H // 16, pf, out)
            assert got.cpu().numpy().tobytes() == wants[k], "call %d" % i
            st = enc.strip_graph_stats()
            assert not st["broken"], enc._L.m2v_last_error(enc._h)
            assert st["last_call_was_graph"] == (i >= 1), "call %d" % i
        assert st["recordings"] == 1 and st["launches"] == len(order) - 1
        # another shape: recorded anew; the port path and the resident entry in between are not disturbed
        assert enc.encode(clips[0], W // 16, H // 16, pf) == wants[0]
        short = orc.encode(clips[1][:5], W // 16, H // 16, pf, 7, 7, 3, 2)
        for i in range(4):
            got = M.parallel.encode_strips_native(enc, None, 0, 1, d_clips[1][:5], W // 16, H // 16, pf, out)
            assert got.cpu().numpy().tobytes() == short
        assert enc.strip_graph_stats()["recordings"] == 2 and enc.strip_graph_stats()["last_call_was_graph"]
        enc.set_option("strip_graph", 0)
        got = M.parallel.encode_strips_native(enc, None, 0, 1, d_clips[0], W // 16, H // 16, pf, out)
        assert got.cpu().numpy().tobytes() == wants[0] and not enc.strip_graph_stats()["last_call_was_graph"]
    finally:
        enc.close()


def _solo_bytes(M, d_clip, W, H, pf, world, rank, rccl, graph, calls, general=False):
    """one rank of `world` alone on the GPU: the bytes rank `rank` (the output rank) assembles, after every call"""
    import torch
    n = int(d_clip.shape[0])
    out = torch.zeros(M.parallel.strip_output_bound(n, W, H), dtype=torch.uint8, device="cuda:0")
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    comm = M.StripComm.solo(world, rccl=rccl)
    got, stats = [], None
    try:
        enc.set_option("strip_graph", 1 if graph else 0)
        if general:
            enc.set_option("dct_mfma", 0)
        for _ in range(calls):
            out.zero_()
            torch.cuda.synchronize()
            o = M.parallel.encode_strips_native(enc, comm, rank, world, d_clip, W // 16, H // 16, pf, out, dst=rank)
            got.append(o.cpu().numpy().tobytes())
        stats = enc.strip_graph_stats()
        assert not stats["broken"], enc._L.m2v_last_error(enc._h)
    finally:
        enc.close()
        comm.close()
    return got, stats


@pytest.mark.parametrize("rccl", [False, True])
@pytest.mark.parametrize("world,rank,general", [(4, 1, False), (4, 0, False), (8, 7, False), (3, 1, True)])
def test_one_rank_of_n_recorded_equals_call_by_call(world, rank, rccl, general):
    """an inner rank (two neighbours), the first and the last rank (one neighbour); general = the form with pack / unpack kernels
    and the exchange on a stream of its own (options conformant / dct_mfma = 0)"""
    import torch
    import m2v_load
    M = m2v_load.load()
    W, H, pf, n = 128, 256, 4, 12            # 16 macroblock rows: strips of 2 .. 6 rows
    d_clip = torch.from_numpy(np.ascontiguousarray(M.synth.clip(W, H, n, clip_index=210, scene_len=7))).to("cuda:0")
    ref, st0 = _solo_bytes(M, d_clip, W, H, pf, world, rank, rccl, False, 2, general)
    assert ref[0] == ref[1] and len(ref[0]) > 1000 and st0["launches"] == 0
    got, st = _solo_bytes(M, d_clip, W, H, pf, world, rank, rccl, True, 5, general)
    import os
    if general or os.environ.get("GPU_MAX_HW_QUEUES") == "1":
        # general: enqueued call by call whatever the option says (see m2v_strip_encode); one hardware queue: no recording of a form with
        # parallel branches (hipGraphLaunch of one crashes inside the runtime there)
        assert st["recordings"] == 0 and st["launches"] == 0
    else:
        assert st["recordings"] == 1 and st["launches"] == 4 and st["last_call_was_graph"]
    for i, g in enumerate(got):
        assert g == ref[0], "call %d differs from the call-by-call sequence" % i


def test_solo_copies_and_solo_rccl_move_the_same_bytes():
    """the two solo transports are interchangeable: RCCL's send / recv to itself delivers what the device copy delivers"""
    import torch
    import m2v_load
    M = m2v_load.load()
    W, H, pf, n = 128, 256, 4, 12
    d_clip = torch.from_numpy(np.ascontiguousarray(M.synth.clip(W, H, n, clip_index=211))).to("cuda:0")
    a, _ = _solo_bytes(M, d_clip, W, H, pf, 4, 2, False, False, 1)
    b, _ = _solo_bytes(M, d_clip, W, H, pf, 4, 2, True, False, 1)
    assert a[0] == b[0]


def test_rccl_send_recv_inside_a_recording():
    """ncclGroupStart; ncclSend / ncclRecv (to the rank itself); ncclGroupEnd recorded by stream capture, instantiated, launched
    three times: the bytes arrive every time.  And the in-process communicator says that it cannot be recorded."""
    import ctypes
    import torch
    import m2v_load
    M = m2v_load.load()
    comm = M.StripComm.rccl(0, 1, 0)
    try:
        a = torch.arange(1 << 16, dtype=torch.int32, device="cuda:0").view(torch.uint8)
        b = torch.zeros_like(a)
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        comm.selftest_captured(0, a.data_ptr(), b.data_ptr(), a.numel(), s.cuda_stream, launches=3)
        assert torch.equal(a, b)
        b.zero_()
        a.add_(1)
        torch.cuda.synchronize()
        comm.selftest_captured(0, a.data_ptr(), b.data_ptr(), a.numel(), 0, launches=1)       # on a stream of the call's own
        assert torch.equal(a, b)
    finally:
        comm.close()
    loc = M.StripComm.local(2)
    try:
        L = M.lib()
        assert L.m2v_comm_selftest_captured(loc.handle, 0, a.data_ptr(), b.data_ptr(), 16, None, 1) == -4
        assert b"cannot be recorded" in L.m2v_comm_last_error()
    finally:
        loc.close()


def test_config_c5_geometry_one_rank_of_eight_recorded():
    """the real size: 2048x2048, one GOP of 1 I + 8 P, rank 3 of 8 (16 macroblock rows, two neighbours), RCCL kernels in the
    recording; recorded == call by call, and the host time per GOP step is printed for both"""
    import torch
    import m2v_load
    M = m2v_load.load()
    W = H = 2048
    pf, n = 8, 9
    d_clip = M.synth.clip_torch(W, H, n, clip_index=57, device="cuda:0", scene_len=5)
    ref, _ = _solo_bytes(M, d_clip, W, H, pf, 8, 3, True, False, 1)
    got, st = _solo_bytes(M, d_clip, W, H, pf, 8, 3, True, True, 4)
    import os
    assert st["launches"] == (0 if os.environ.get("GPU_MAX_HW_QUEUES") == "1" else 3) and all(g == ref[0] for g in got)      # (one queue: never recorded)
