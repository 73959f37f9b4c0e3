"""-m gpu: error behaviour of the C-ABI (codes of include/m2v_mi355x.h), reset, options."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_overflow_reset_and_options():
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    L = M.lib()
    clip = M.synth.clip(96, 64, 4, clip_index=120)
    want = orc.encode(clip, 6, 4, 3, 6, 6, 3, 2)
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    try:
        d_in = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
        small = torch.zeros(256, dtype=torch.uint8, device="cuda:0")
        n = ctypes.c_size_t(0)
        # output buffer too small: M2V_E_OVERFLOW, nothing written beyond the buffer, handle stays usable
        r = L.m2v_encode_resident(enc._h, 6, 4, 3, d_in.data_ptr(), 4, small.data_ptr(), small.numel(), ctypes.byref(n), None)
        assert r == -6 and b"too small" in L.m2v_last_error(enc._h)
        big = torch.empty(1 << 20, dtype=torch.uint8, device="cuda:0")
        nb = enc.encode_resident(d_in.data_ptr(), 4, big.data_ptr(), big.numel(), 6, 4, 3)
        assert big[:nb].cpu().numpy().tobytes() == want
        # busy handle refuses the resident entry; reset drops the sequence in flight (rstn, RTL:1028-1039)
        enc.push_frames(6, 4, 3, clip[:2])
        assert enc.busy
        r = L.m2v_encode_resident(enc._h, 6, 4, 3, d_in.data_ptr(), 4, big.data_ptr(), big.numel(), ctypes.byref(n), None)
        assert r == -4
        assert L.m2v_reset(enc._h) == 0 and not enc.busy
        assert enc.pull() == (b"", False)
        assert enc.encode(clip, 6, 4, 3) == want                       # a fresh sequence after the reset
        # options
        assert L.m2v_set_option(enc._h, b"batch_frames", 0) == -1
        assert L.m2v_set_option(enc._h, b"no_such_option", 1) == -1
        assert L.m2v_set_option(enc._h, b"batch_frames", 2) == 0
        assert enc.encode(clip, 6, 4, 3) == want
        assert L.m2v_set_option(enc._h, b"batch_frames", 100000) == -1  # beyond the sanity bound: refused, with a text
        assert b"65536" in L.m2v_last_error(enc._h)
        assert L.m2v_set_option(enc._h, b"batch_frames", 1000) == 0     # no 200-frame cap any more: offsets inside a chunk are 64-bit
        assert enc.encode(clip, 6, 4, 3) == want
        # zero frames: the sequence never starts (stop while idle does nothing, RTL:1090)
        assert enc.encode_resident(d_in.data_ptr(), 0, big.data_ptr(), big.numel(), 6, 4, 3) == 0
        enc.sequence_stop()
        assert not enc.busy
    finally:
        enc.close()


def test_two_handles_are_independent():
    """config c4 in one process: two encoder instances interleaved on one GPU do not disturb each other."""
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    a, b = M.Mpeg2Encoder(6, 6, 3, 2), M.Mpeg2Encoder(5, 5, 1, 4)
    try:
        ca, cb = M.synth.clip(128, 96, 5, clip_index=121), M.synth.clip(64, 64, 6, clip_index=122)
        for k in range(5):
            a.push_frames(8, 6, 2, ca[k:k + 1])
            b.push_frames(4, 4, 4, cb[k:k + 1])
        b.push_frames(4, 4, 4, cb[5:6])
        a.sequence_stop()
        b.sequence_stop()
        assert a.pull_all() == orc.encode(ca, 8, 6, 2, 6, 6, 3, 2)
        assert b.pull_all() == orc.encode(cb, 4, 4, 4, 5, 5, 1, 4)
    finally:
        a.close()
        b.close()


def test_reset_in_the_middle_of_a_strip_sequence():
    """m2v_reset between m2v_strip_begin and m2v_strip_finish: the handle is idle again, full-frame geometry, and the
    other entry points refuse to run while the strip sequence is open"""
    import ctypes
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    L = M.lib()
    clip = M.synth.clip(96, 96, 4, clip_index=123)
    want = orc.encode(clip, 6, 6, 3, 6, 6, 3, 2)
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    try:
        d_in = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
        big = torch.empty(1 << 20, dtype=torch.uint8, device="cuda:0")
        steps, hb = enc.strip_begin(d_in.data_ptr(), 4, 6, 6, 3, 2, 4)
        assert steps == 4 and hb > 0
        enc.strip_step(0, 0, 0)
        n = ctypes.c_size_t(0)
        assert L.m2v_encode_resident(enc._h, 6, 6, 3, d_in.data_ptr(), 4, big.data_ptr(), big.numel(), ctypes.byref(n), None) == -4
        assert L.m2v_push_frames(enc._h, 6, 6, 3, clip.ctypes.data, 1) == -4
        assert L.m2v_strip_begin(enc._h, 6, 6, 3, d_in.data_ptr(), 4, 0, 2, None) == -4
        assert L.m2v_reset(enc._h) == 0 and not enc.busy
        nb = enc.encode_resident(d_in.data_ptr(), 4, big.data_ptr(), big.numel(), 6, 6, 3)      # whole frames again
        assert big[:nb].cpu().numpy().tobytes() == want
        assert enc.encode(clip, 6, 6, 3) == want
        steps2, _ = enc.strip_begin(d_in.data_ptr(), 4, 6, 6, 3, 0, 6)                          # and strips work again
        assert steps2 == 4
        assert L.m2v_reset(enc._h) == 0
    finally:
        enc.close()


def test_create_failure_reports_text_without_a_handle():
    import ctypes
    import m2v_load
    M = m2v_load.load()
    L = M.lib()
    err = ctypes.c_int(0)
    assert not L.m2v_create(9, 6, 3, 2, 0, ctypes.byref(err)) and err.value == -1
    assert b"XL" in L.m2v_last_error(None)
    assert not L.m2v_create(6, 6, 3, 2, 99, ctypes.byref(err)) and err.value == -2
    assert b"device" in L.m2v_last_error(None)
    # the shipped library has no profiling / dump switches
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    try:
        assert L.m2v_set_option(enc._h, b"ablate", 1) == -1 and L.m2v_set_option(enc._h, b"keep_recon", 1) == -1
        assert b"unknown option" in L.m2v_last_error(enc._h)
    finally:
        enc.close()


def test_config_c4_eight_handles_from_eight_threads():
    """BASELINE config c4 as far as one GPU can show it: 8 independent 1920x1152 sequences (one whole GOP of 1 I + 8 P
    each, VECTOR_LEVEL 3, Q_LEVEL 2), every handle created, driven and destroyed by its own thread at the same time
    (the first m2v_create on a device uploads the constant tables: once, whichever thread gets there first).
    Every stream byte-identical to the oracle's."""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    M.lib()
    orc.build()
    W, H, pf, n, T = 1920, 1152, 8, 9, 8
    clips = [M.synth.clip_torch(W, H, n, clip_index=200 + t, device="cuda:0", scene_len=4 + t) for t in range(T)]
    host = [c.cpu().numpy() for c in clips]
    torch.cuda.synchronize()
    gate = threading.Barrier(T)
    got = [None] * T

    def worker(t):
        gate.wait()                                            # all eight m2v_create calls race
        enc = M.Mpeg2Encoder(7, 7, 3, 2, device=0)
        try:
            d_out = torch.empty(n * W * H, dtype=torch.uint8, device="cuda:0")
            for _ in range(2):                                 # twice: buffers are reused while the other handles run
                nb = enc.encode_resident(clips[t].data_ptr(), n, d_out.data_ptr(), d_out.numel(), 120, 72, pf)
            got[t] = d_out[:nb].cpu().numpy().tobytes()
        finally:
            enc.close()

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    with ThreadPoolExecutor(T) as ex:                          # the oracle drops the GIL: one GOP per core
        want = list(ex.map(lambda c: orc.encode(c, 120, 72, pf, 7, 7, 3, 2), host))
    for t in range(T):
        assert got[t] is not None, "thread %d died" % t
        assert got[t] == want[t], "sequence %d" % t


def test_strip_encode_refuses_bad_ranks_and_leaves_the_handle_usable():
    """m2v_strip_encode: no communicator for more than one rank, a communicator of another size, more ranks than macroblock rows,
    an output rank without a buffer - each refused with a code and a text, and the handle encodes normally afterwards."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    clip = M.synth.clip(96, 64, 5, clip_index=131)
    want = orc.encode(clip, 6, 4, 2, 6, 6, 3, 2)
    d_in = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    out = torch.empty(1 << 20, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    c3 = M.StripComm.local(3)
    L, n = enc._L, ctypes.c_size_t(0)

    def call(comm, rank, world, dst, d_out, cap):
        return L.m2v_strip_encode(enc._h, comm, rank, world, dst, 6, 4, 2, d_in.data_ptr(), 5, d_out, cap, ctypes.byref(n), None)
    try:
        assert call(None, 0, 2, 0, out.data_ptr(), out.numel()) == -1 and b"communicator" in L.m2v_last_error(enc._h)
        assert call(c3.handle, 0, 2, 0, out.data_ptr(), out.numel()) == -1            # a communicator of three for a world of two
        assert call(None, 0, 5, 0, out.data_ptr(), out.numel()) == -1                 # four macroblock rows cannot make five strips
        assert call(None, 1, 1, 0, out.data_ptr(), out.numel()) == -1                 # rank outside the world
        assert call(None, 0, 1, 0, None, 0) == -1 and b"d_out" in L.m2v_last_error(enc._h)
        assert call(None, 0, 1, 0, out.data_ptr(), 64) == -6                          # M2V_E_OVERFLOW: the stream does not fit
        assert not enc.busy
        assert call(None, 0, 1, 0, out.data_ptr(), out.numel()) == 0
        assert out[:n.value].cpu().numpy().tobytes() == want
        assert enc.encode(clip, 6, 4, 2) == want                                      # and the port path after it
    finally:
        c3.close()
        enc.close()


def test_resident_begin_end_two_handles_taking_turns():
    """m2v_encode_resident_begin / _end: two handles keep two sequences in flight (what bench.py's timed loop does); the bytes are
    the oracle's for both, every time; between the two halves the handle refuses everything else."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    clips = [M.synth.clip(160, 96, 7, clip_index=140 + k, scene_len=4) for k in range(2)]
    wants = [orc.encode(c, 10, 6, 2, 6, 6, 3, 2) for c in clips]
    d_in = [torch.from_numpy(np.ascontiguousarray(c)).to("cuda:0") for c in clips]
    d_out = [torch.zeros(1 << 20, dtype=torch.uint8, device="cuda:0") for _ in range(2)]
    torch.cuda.synchronize()
    encs = [M.Mpeg2Encoder(6, 6, 3, 2) for _ in range(2)]
    L = encs[0]._L
    try:
        assert L.m2v_encode_resident_end(encs[0]._h, None) == -4                      # nothing in flight
        busy = [False, False]
        for i in range(9):
            h = i % 2
            if busy[h]:
                n = encs[h].encode_resident_end()
                assert d_out[h][:n].cpu().numpy().tobytes() == wants[h], "step %d" % i
                d_out[h].zero_()
                torch.cuda.synchronize()
            encs[h].encode_resident_begin(d_in[h].data_ptr(), 7, d_out[h].data_ptr(), d_out[h].numel(), 10, 6, 2)
            busy[h] = True
            # in flight: nothing else is accepted
            n0 = ctypes.c_size_t(0)
            assert L.m2v_encode_resident(encs[h]._h, 10, 6, 2, d_in[h].data_ptr(), 7, d_out[h].data_ptr(), d_out[h].numel(), ctypes.byref(n0), None) == -4
            assert L.m2v_push_frames(encs[h]._h, 10, 6, 2, clips[h].ctypes.data, 1) == -4
        for h in range(2):
            n = encs[h].encode_resident_end()
            assert d_out[h][:n].cpu().numpy().tobytes() == wants[h]
            assert encs[h].encode(clips[h], 10, 6, 2) == wants[h]                      # the port path afterwards
        # an empty sequence, and a reset while one is in flight
        encs[0].encode_resident_begin(d_in[0].data_ptr(), 0, d_out[0].data_ptr(), d_out[0].numel(), 10, 6, 2)
        assert encs[0].encode_resident_end() == 0
        encs[0].encode_resident_begin(d_in[0].data_ptr(), 7, d_out[0].data_ptr(), d_out[0].numel(), 10, 6, 2)
        assert L.m2v_reset(encs[0]._h) == 0
        assert encs[0].encode(clips[0], 10, 6, 2) == wants[0]
    finally:
        for e in encs:
            e.close()


def test_destroy_and_reset_wait_for_a_sequence_on_a_caller_stream():
    """m2v_encode_resident_begin on a CALLER's stream: the scans, the assembly and the control read-back are queued there, not on the
    handle's stream.  m2v_reset must wait for them before the next call rewrites the work buffers, m2v_destroy before it frees them
    (round-3 advisor).  A long clip keeps the GPU busy for milliseconds after _begin returns; the stream of the sequence enqueued
    right after the reset is the oracle's, and destroying a handle in flight leaves the process and the GPU in working order."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    big = M.synth.clip(640, 480, 36, clip_index=150)
    small = M.synth.clip(96, 64, 4, clip_index=151)
    want_small = orc.encode(small, 6, 4, 3, 6, 6, 3, 2)
    d_big = torch.from_numpy(np.ascontiguousarray(big)).to("cuda:0")
    d_small = torch.from_numpy(np.ascontiguousarray(small)).to("cuda:0")
    d_out = torch.zeros(8 << 20, dtype=torch.uint8, device="cuda:0")
    mine = torch.cuda.Stream()
    torch.cuda.synchronize()
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    L = enc._L
    try:
        enc.encode_resident_begin(d_big.data_ptr(), 36, d_out.data_ptr(), d_out.numel(), 40, 30, 8, stream=mine.cuda_stream)
        assert L.m2v_set_option(enc._h, b"split_streams", 1) == -4         # planned with the current options: not while in flight
        assert L.m2v_reset(enc._h) == 0
        assert mine.query(), "m2v_reset returned while the sequence was still running on the caller's stream"
        n = enc.encode_resident(d_small.data_ptr(), 4, d_out.data_ptr(), d_out.numel(), 6, 4, 3)
        assert d_out[:n].cpu().numpy().tobytes() == want_small
        # a stray _end after a SYNCHRONOUS empty sequence has nothing to answer
        assert enc.encode_resident(d_small.data_ptr(), 0, d_out.data_ptr(), d_out.numel(), 6, 4, 3) == 0
        assert L.m2v_encode_resident_end(enc._h, None) == -4
        enc.encode_resident_begin(d_big.data_ptr(), 36, d_out.data_ptr(), d_out.numel(), 40, 30, 8, stream=mine.cuda_stream)
    finally:
        enc.close()                                                         # m2v_destroy with the sequence in flight
    assert mine.query(), "m2v_destroy returned while the sequence was still running on the caller's stream"
    torch.cuda.synchronize()
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    try:
        n = enc.encode_resident(d_small.data_ptr(), 4, d_out.data_ptr(), d_out.numel(), 6, 4, 3)
        assert d_out[:n].cpu().numpy().tobytes() == want_small
    finally:
        enc.close()
