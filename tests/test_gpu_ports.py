"""-m gpu: the port-level contract through the C-ABI (beats in, 32-byte words out), against the oracle.

Mirrors what SIM/tb_mpeg2encoder.v exercises (three sequences back to back on one instance, TB:150) and the
cases SURVEY.md 8(f1) lists: mid-frame stop / black fill, size clamp, pframes 0..255, bubbles."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    return m2v_load.load(), orc


def beats_of(frame_block):
    """[n,3,H,W] -> three flat arrays in beat order (raster, 4 px per beat)"""
    f = np.ascontiguousarray(frame_block)
    return f[:, 0].reshape(-1), f[:, 1].reshape(-1), f[:, 2].reshape(-1)


def test_three_sequences_back_to_back(env):
    """TB:150 - one encoder instance, three different videos, different sizes."""
    M, orc = env
    enc = M.Mpeg2Encoder(7, 6, 3, 2)
    try:
        for k, (W, H, n) in enumerate([(288, 208, 4), (640, 320, 3), (160, 704, 3)]):
            clip = M.synth.clip(W, H, n, clip_index=20 + k)
            want = orc.encode(clip, W // 16, H // 16, 23, 7, 6, 3, 2)
            assert not enc.busy
            got = enc.encode(clip, W // 16, H // 16, 23)
            assert got == want, "sequence %d" % k
            assert not enc.busy
    finally:
        enc.close()


def test_irregular_beat_batches_and_stop_with_last(env):
    """Bubbles (TB:233) = arbitrary beat batch sizes; stop raised together with the last beat (RTL:1082-1083)."""
    M, orc = env
    W, H, n, pf = 96, 64, 5, 2
    clip = M.synth.clip(W, H, n, clip_index=30)
    want = orc.encode(clip, 6, 4, pf, 6, 6, 3, 2)
    y, u, v = (a.reshape(-1) for a in (clip[:, 0], clip[:, 1], clip[:, 2]))
    # beats are per frame: frame f beat b -> pixels of frame f; build per-frame beat streams
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    try:
        rng = np.random.default_rng(5)
        bpf = W * H // 4
        for f in range(n):
            fy, fu, fv = clip[f, 0].reshape(-1), clip[f, 1].reshape(-1), clip[f, 2].reshape(-1)
            b = 0
            while b < bpf:
                take = int(min(bpf - b, rng.integers(1, 700)))
                last = f == n - 1 and b + take == bpf
                enc.push_beats(6, 4, pf, fy[4 * b:4 * (b + take)], fu[4 * b:4 * (b + take)], fv[4 * b:4 * (b + take)],
                               stop_with_last=last)
                b += take
        assert enc.busy
        got = enc.pull_all()
        assert got == want and not enc.busy
    finally:
        enc.close()


@pytest.mark.parametrize("cut", [1, 333, 96 * 64 // 4 - 1])
def test_stop_inside_a_frame(env, cut):
    """i_sequence_stop mid-frame: the frame is completed with Y=0, U=V=0x80 (RTL:1036-1056)."""
    M, orc = env
    W, H = 96, 64
    clip = M.synth.clip(W, H, 3, clip_index=31)
    bpf = W * H // 4
    nbeats = 2 * bpf + cut
    want = orc.encode(clip, 6, 4, 2, 6, 6, 3, 2, nbeats=nbeats)
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    try:
        got = enc.encode(clip, 6, 4, 2, nbeats=nbeats)
        assert got == want
        # beats pushed while the sequence is ending are dropped (RTL:1045-1058): push, stop, push again before pulling
        enc.push_frames(6, 4, 2, clip[:1])
        enc.sequence_stop()
        enc.push_frames(6, 4, 2, clip[1:2])          # dropped
        got2 = enc.pull_all()
        assert got2 == orc.encode(clip[:1], 6, 4, 2, 6, 6, 3, 2)
    finally:
        enc.close()


def test_chunk_boundary_inside_a_gop_streaming(env):
    """Frames are encoded in chunks of batch_frames; a GOP that straddles chunks continues from the persisted
    reconstruction, and the stream does not depend on the chunking."""
    M, orc = env
    W, H, n, pf = 128, 96, 11, 4
    clip = M.synth.clip(W, H, n, clip_index=32)
    want = orc.encode(clip, 8, 6, pf, 7, 7, 3, 2)
    for bf in (1, 2, 3, 7, 64):
        enc = M.Mpeg2Encoder(7, 7, 3, 2)
        try:
            enc.set_option("batch_frames", bf)
            got = enc.encode(clip, 8, 6, pf)
            assert got == want, "batch_frames=%d" % bf
        finally:
            enc.close()


def test_size_clamp_and_pframes_255(env):
    """Out-of-range sizes are clamped like RTL:985-991; i_pframes_count = 255 gives one I frame then only P."""
    M, orc = env
    enc = M.Mpeg2Encoder(4, 4, 1, 2)                       # max 256x256
    try:
        assert enc.geometry(20, 2) == (256, 64)            # 20 > 2^XL -> clamped; 2 < 4 -> clamped
        clip = M.synth.clip(256, 64, 3, clip_index=33)
        want = orc.encode(clip, 20, 2, 255, 4, 4, 1, 2)
        assert enc.encode(clip, 20, 2, 255) == want
        assert want.count(b"\x00\x00\x01\xb8") == 1
    finally:
        enc.close()


def test_pull_granularity(env):
    """m2v_pull hands out whole 32-byte words in order; `last` arrives with the final word only."""
    M, orc = env
    clip = M.synth.clip(64, 64, 2, clip_index=34)
    want = orc.encode(clip, 4, 4, 1, 4, 4, 2, 3)
    enc = M.Mpeg2Encoder(4, 4, 2, 3)
    try:
        enc.push_frames(4, 4, 1, clip)
        enc.sequence_stop()
        out, lasts = b"", []
        while True:
            b, last = enc.pull(100)                    # cap not a multiple of 32 -> 96 bytes at a time
            assert len(b) % 32 == 0 and len(b) <= 96
            out += b
            lasts.append(last)
            if last:
                break
        assert out == want and lasts.count(True) == 1
    finally:
        enc.close()


@pytest.mark.parametrize("use_async", [1, 0])
def test_double_buffered_chunks_with_interleaved_pulls(env, use_async):
    """The port path keeps up to two chunks in flight (option "async"); pulling while frames are still being
    pushed returns whatever is complete, in stream order, and the bytes do not depend on the overlap."""
    M, orc = env
    W, H, n, pf = 192, 128, 23, 5
    clip = M.synth.clip(W, H, n, clip_index=36)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 2, 2)
    enc = M.Mpeg2Encoder(7, 7, 2, 2)
    try:
        enc.set_option("batch_frames", 4)
        enc.set_option("async", use_async)
        out = b""
        early = 0
        for k in range(0, n, 3):                                   # pushes that do not line up with the chunks
            enc.push_frames(W // 16, H // 16, pf, clip[k:k + 3])
            b, last = enc.pull(1 << 20)
            assert not last and len(b) % 32 == 0
            early += len(b)
            out += b
        assert enc.busy
        enc.sequence_stop()
        while True:
            b, last = enc.pull(4096)
            out += b
            if last:
                break
        assert out == want
        assert not enc.busy
        if not use_async:
            assert early > 0                                       # synchronous chunks are complete when push returns
        # the same handle, next sequence, beat interface across the stage flip
        y, u, v = beats_of(clip[:9])
        want2 = orc.encode(clip[:9], W // 16, H // 16, 2, 7, 7, 2, 2)
        step = 4 * 1237
        for o in range(0, y.size, step):
            enc.push_beats(W // 16, H // 16, 2, y[o:o + step], u[o:o + step], v[o:o + step], stop_with_last=(o + step >= y.size))
        got2 = b""
        while True:
            b, last = enc.pull(1 << 16)
            got2 += b
            if last:
                break
        assert got2 == want2
    finally:
        enc.close()


@pytest.mark.parametrize("layout", ["yuv24", "uyv24", "yuvx32", "ayuv32"])
def test_packed_444_layouts(env, layout):
    """m2v_push_packed: the beats of RTL:25-28 delivered as interleaved samples; same stream as the planar beats."""
    M, orc = env
    W, H, n, pf = 80, 64, 5, 2
    clip = M.synth.clip(W, H, n, clip_index=37)                     # [n,3,H,W]
    want = orc.encode(clip, W // 16, H // 16, pf, 6, 6, 2, 2)
    y, u, v = clip[:, 0], clip[:, 1], clip[:, 2]
    pad = np.full_like(y, 0xA5)
    order = {"yuv24": (y, u, v), "uyv24": (u, y, v), "yuvx32": (y, u, v, pad), "ayuv32": (pad, y, u, v)}[layout]
    packed = np.stack(order, axis=-1).reshape(-1)                   # raster order, interleaved per pixel
    bpp = len(order)
    enc = M.Mpeg2Encoder(6, 6, 2, 2)
    try:
        cut = 4 * bpp * 5003                                        # pushes that do not line up with frames or blocks
        for o in range(0, packed.size, cut):
            enc.push_packed(W // 16, H // 16, pf, packed[o:o + cut], layout, stop_with_last=(o + cut >= packed.size))
        assert enc.pull_all() == want
    finally:
        enc.close()


def _packed_of(clip, layout):
    y, u, v = clip[:, 0], clip[:, 1], clip[:, 2]
    pad = np.full_like(y, 0x5A)
    order = {"yuv24": (y, u, v), "uyv24": (u, y, v), "yuvx32": (y, u, v, pad), "ayuv32": (pad, y, u, v)}[layout]
    return np.ascontiguousarray(np.stack(order, axis=-1)).reshape(-1), len(order)


@pytest.mark.parametrize("layout,page_locked", [("yuv24", True), ("yuv24", False), ("ayuv32", True), ("uyv24", False)])
def test_packed_frames_stay_packed_until_they_are_in_hbm(env, layout, page_locked):
    """Whole frames per call, several chunks (batch_frames 4), from page-locked memory (uploaded as they are, k_unpack444 turns them
    into planes on the device) and from ordinary memory (through the packed pinned staging); the caller's buffer is reused - and
    overwritten - as soon as a call returns.  Then the same with a stop in the middle of a frame: a packed frame's black fill is the
    macroblock kernel's (FrameJob::valid_beats), RTL:1036-1056."""
    import torch
    M, orc = env
    W, H, n, pf = 96, 64, 11, 3
    clip = M.synth.clip(W, H, n, clip_index=44)
    want = orc.encode(clip, W // 16, H // 16, pf, 6, 6, 3, 2)
    packed, bpp = _packed_of(clip, layout)
    fbytes = W * H * bpp
    enc = M.Mpeg2Encoder(6, 6, 3, 2)
    try:
        enc.set_option("batch_frames", 4)
        scratch = torch.empty(3 * fbytes, dtype=torch.uint8)
        scratch = (scratch.pin_memory() if page_locked else scratch).numpy()
        k = 0
        for take in (3, 1, 2, 3, 2):                                 # chunk boundaries inside and between the calls
            scratch[:take * fbytes] = packed[k * fbytes:(k + take) * fbytes]
            enc.push_packed(W // 16, H // 16, pf, scratch[:take * fbytes], layout)
            scratch[:] = 0xEE                                        # the bytes have left the caller's buffer when the call returns
            k += take
        assert k == n
        enc.sequence_stop()
        assert enc.pull_all() == want
        # stop inside frame 5 (beats of 2.5 rows of it delivered)
        cut_beats = 5 * (W * H // 4) + (2 * W + W // 2) // 4
        want_cut = orc.encode(clip[:6], W // 16, H // 16, pf, 6, 6, 3, 2, nbeats=cut_beats)
        scratch2 = torch.empty(cut_beats * 4 * bpp, dtype=torch.uint8)
        scratch2 = (scratch2.pin_memory() if page_locked else scratch2).numpy()
        scratch2[:] = packed[:cut_beats * 4 * bpp]
        enc.push_packed(W // 16, H // 16, pf, scratch2, layout, stop_with_last=True)
        assert enc.pull_all() == want_cut
    finally:
        enc.close()


def test_packed_and_planar_beats_mixed_inside_frames_and_chunks(env):
    """Every mixture a caller can produce through the two beat entry points: a frame begun with planar beats and finished with packed
    ones and the other way round, a 32-bit layout arriving in a chunk sized for a 24-bit one (the chunk leaves early), planar frames
    between packed ones in one chunk, page-locked and ordinary sources in turn."""
    import torch
    M, orc = env
    W, H, n, pf = 64, 64, 14, 4
    clip = M.synth.clip(W, H, n, clip_index=45)
    want = orc.encode(clip, W // 16, H // 16, pf, 6, 6, 2, 3)
    bpf = W * H // 4
    layouts = ["yuv24", "uyv24", "yuvx32", "ayuv32"]
    packs = {l: _packed_of(clip, l) for l in layouts}
    pinned = {l: torch.from_numpy(packs[l][0]).pin_memory().numpy() for l in layouts}
    yb, ub, vb = (np.ascontiguousarray(clip[:, c]).reshape(-1) for c in range(3))
    yp, up_, vp = (torch.from_numpy(a).pin_memory().numpy() for a in (yb, ub, vb))       # the three beat arrays page-locked: whole frames go up from there
    rng = np.random.default_rng(99)
    enc = M.Mpeg2Encoder(6, 6, 2, 3)
    try:
        enc.set_option("batch_frames", 5)
        b, total = 0, n * bpf
        while b < total:
            take = int(min(total - b, rng.choice([1, 7, bpf // 3, bpf, 2 * bpf + 5, 3 * bpf])))
            kind = int(rng.integers(0, 11))
            if kind == 0:
                enc.push_beats(W // 16, H // 16, pf, yb[4 * b:4 * (b + take)], ub[4 * b:4 * (b + take)], vb[4 * b:4 * (b + take)])
            elif kind >= 9:
                enc.push_beats(W // 16, H // 16, pf, yp[4 * b:4 * (b + take)], up_[4 * b:4 * (b + take)], vp[4 * b:4 * (b + take)])
            else:
                l = layouts[(kind - 1) % 4]
                src = pinned[l] if kind > 4 else packs[l][0]
                bpp = packs[l][1]
                enc.push_packed(W // 16, H // 16, pf, src[4 * bpp * b:4 * bpp * (b + take)], l)
            b += take
        enc.sequence_stop()
        assert enc.pull_all() == want
    finally:
        enc.close()


def test_long_sequence_many_chunks_in_flight(env):
    """1500 frames through the double-buffered port path in 30 chunks (FIFO compaction, stage reuse, the padding rule
    against the whole-sequence byte count chained on the device), pulled in small pieces while pushing."""
    M, orc = env
    W, H, n, pf = 64, 64, 1500, 7
    rng = np.random.default_rng(123)
    base = M.synth.clip(W, H, 60, clip_index=38)
    clip = base[rng.integers(0, 60, n)]                              # a long, jumpy sequence out of 60 distinct frames
    want = orc.encode(clip, 4, 4, pf, 5, 5, 2, 3)
    enc = M.Mpeg2Encoder(5, 5, 2, 3)
    try:
        enc.set_option("batch_frames", 50)
        out = []
        for k in range(0, n, 37):
            enc.push_frames(4, 4, pf, clip[k:k + 37])
            b, last = enc.pull(7 * 32)                               # drain slower than the encoder produces
            assert not last
            out.append(b)
        enc.sequence_stop()
        out.append(enc.pull_all())
        got = b"".join(out)
        assert len(got) == len(want) and got == want
        assert not enc.busy
    finally:
        enc.close()


def test_page_locked_frames_are_uploaded_directly_and_mix_with_beats():
    """m2v_push_frames from page-locked caller memory (option direct_upload, default on): the frames cross PCIe straight from
    the caller's buffer, which may be overwritten as soon as the call returns; beats and pageable frames pushed into the
    same chunk go through the pinned staging - every mixture must give the oracle's bytes."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, n, pf = 128, 96, 11, 3
    clip = M.synth.clip(W, H, n, clip_index=77, scene_len=4)
    want = orc.encode(clip, W // 16, H // 16, pf, 6, 6, 3, 2)
    bpf = W * H // 4
    for batch in (4, 3, 96):
        for direct in (1, 0):
            enc = M.Mpeg2Encoder(6, 6, 3, 2)
            try:
                enc.set_option("batch_frames", batch)
                enc.set_option("direct_upload", direct)
                scratch = torch.empty((2, 3, H, W), dtype=torch.uint8).pin_memory()
                sview = scratch.numpy()
                out = []
                k = 0
                while k < n:
                    kind = k % 3
                    if kind == 0 and k + 2 <= n:            # two frames from page-locked memory, then the buffer is trashed
                        sview[:] = clip[k:k + 2]
                        enc.push_frames(W // 16, H // 16, pf, sview)
                        sview[:] = 0x5A
                        k += 2
                    elif kind == 1:                         # one frame as beats, in two pieces
                        f = clip[k].reshape(3, -1)
                        cut = bpf // 3
                        enc.push_beats(W // 16, H // 16, pf, f[0][:cut * 4], f[1][:cut * 4], f[2][:cut * 4])
                        enc.push_beats(W // 16, H // 16, pf, f[0][cut * 4:], f[1][cut * 4:], f[2][cut * 4:])
                        k += 1
                    else:                                   # one pageable frame
                        enc.push_frames(W // 16, H // 16, pf, clip[k:k + 1])
                        k += 1
                    out.append(enc.pull()[0])
                enc.sequence_stop()
                out.append(enc.pull_all())
                assert b"".join(out) == want, "batch_frames=%d direct_upload=%d" % (batch, direct)
            finally:
                enc.close()


def test_deferred_upload_and_pull_into_give_the_same_stream():
    """option direct_upload = 2 (m2v_push_frames returns while its frames are still being read; the next push / stop / m2v_upload_wait
    releases them) and m2v_pull straight into the caller's buffer: byte-identical to the blocking path, GOP after GOP from page-locked
    memory, with a push that crosses a chunk boundary, three sequences on one handle."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 192, 128, 3, 22
    clip = M.synth.clip(W, H, n, clip_index=95, scene_len=6)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    pinned = torch.from_numpy(np.ascontiguousarray(clip)).pin_memory().numpy()
    out = np.zeros(len(want) + 4096, np.uint8)
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    try:
        enc.set_option("batch_frames", 8)
        for mode in (2, 1, 2):
            enc.set_option("direct_upload", mode)
            pos = 0
            for k in range(0, n, 5):                          # 5 frames per push, 8 per chunk: pushes that straddle chunks
                enc.push_frames(W // 16, H // 16, pf, pinned[k:k + 5])
                pos += enc.pull_into(out, pos)[0]
            enc.upload_wait()                                 # (what a caller that wants to overwrite its frames now would do)
            enc.sequence_stop()
            last = False
            while not last:
                m, last = enc.pull_into(out, pos)
                pos += m
            assert out[:pos].tobytes() == want, "direct_upload = %d" % mode
            assert not enc.busy
    finally:
        enc.close()


def test_push_frames_pull_is_push_then_pull():
    """m2v_push_frames_pull (both port groups in one call): byte-identical to the oracle from page-locked and from pageable memory, with
    pushes that complete a chunk exactly (the call then waits for the chunk's upload event), pushes that span several chunks, a push
    per frame, and a destination so small that words queue up in the FIFO behind it; the handle is reused for all of them."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 192, 128, 3, 24
    clip = M.synth.clip(W, H, n, clip_index=96, scene_len=7)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    pinned = torch.from_numpy(np.ascontiguousarray(clip)).pin_memory().numpy()
    enc = M.Mpeg2Encoder(7, 7, 3, 2)
    try:
        for src, batch, per_push, cap in ((pinned, 4, 4, None), (pinned, 4, 12, None), (pinned, 8, 1, None), (clip, 4, 4, None),
                                          (pinned, 4, 4, 64), (pinned, 4, 2, 4096), (clip, 8, 5, 96)):
            enc.set_option("batch_frames", batch)
            out = np.zeros(len(want) + 4096, np.uint8)
            pos, last = 0, False
            for k in range(0, n, per_push):
                window = out if cap is None else out[:min(out.size, pos + cap)]
                m, last = enc.push_frames_pull(W // 16, H // 16, pf, src[k:k + per_push], window, pos)
                pos += m
                assert not last
            enc.sequence_stop()
            drain_by_push = batch == 8 and per_push == 1      # one case drains through the combined call: its frames are dropped while the
            while not last:                                    # sequence ends (RTL:1045-1058), its pull half waits like m2v_pull
                if drain_by_push:
                    m, last = enc.push_frames_pull(W // 16, H // 16, pf, src[:1], out, pos)
                else:
                    m, last = enc.pull_into(out, pos)
                pos += m
            assert out[:pos].tobytes() == want, (batch, per_push, cap)
            assert not enc.busy
    finally:
        enc.close()


def test_deferred_upload_chunk_completed_by_the_call_itself():
    """option direct_upload = 2 with pushes that complete a chunk exactly and frames large enough that the transfer is still under way when
    the chunk's kernels are queued: the kernels must wait for THIS call's transfer although the call does not (the chunk's first launch
    sits behind an event on the upload stream; a missing event shows here as a stream encoded from the frames of the sequence before -
    two different clips take turns, so what is left in the device buffers is never what is wanted)."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, pf, n = 1280, 720, 3, 12
    clips = [M.synth.clip(W, H, n, clip_index=97 + i, scene_len=5) for i in range(2)]
    wants = [orc.encode(c, W // 16, H // 16, pf, 7, 7, 2, 2) for c in clips]
    pinned = [torch.from_numpy(np.ascontiguousarray(c)).pin_memory().numpy() for c in clips]
    out = np.zeros(max(len(w) for w in wants) + 4096, np.uint8)
    enc = M.Mpeg2Encoder(7, 7, 2, 2)
    try:
        enc.set_option("batch_frames", 4)
        rnd = 0
        for mode in (2, 1, 2):
            enc.set_option("direct_upload", mode)
            for rep in range(2):
                src, want = pinned[rnd & 1], wants[rnd & 1]
                rnd += 1
                pos = 0
                for k in range(0, n, 4):
                    enc.push_frames(W // 16, H // 16, pf, src[k:k + 4])
                    pos += enc.pull_into(out, pos)[0]
                enc.sequence_stop()
                last = False
                while not last:
                    m, last = enc.pull_into(out, pos)
                    pos += m
                assert out[:pos].tobytes() == want, "direct_upload = %d, sequence %d" % (mode, rnd)
    finally:
        enc.close()


def test_one_hardware_queue_does_not_stall_the_port_path():
    """With GPU_MAX_HW_QUEUES=1 every stream of the process shares ONE hardware queue: anything the port path queues that waits for the
    host - round 5 tried a gate kernel in front of a chunk's launches, released by the call - or any wait of the host for something queued
    behind such a thing shows as a stall of seconds.  A fresh process with one queue runs the shapes that would - chunks completed by a
    call, calls spanning chunks, no pulls in between (the stage-reuse wait), the combined call, a second larger geometry (buffers grow) -
    and must finish in seconds, bytes identical."""
    import subprocess
    import sys
    import time
    code = r'''
import sys, time
sys.path.insert(0, %r)
import numpy as np, torch, m2v_load
from oracle import m2v_oracle_ctypes as orc
M = m2v_load.load()
enc = M.Mpeg2Encoder(7, 7, 3, 2)
t0 = time.perf_counter()
for (W, H, n, pf, batch, per_push, pull) in ((192, 128, 24, 3, 4, 4, 1), (192, 128, 24, 3, 4, 12, 0), (320, 192, 20, 4, 5, 5, 2), (192, 128, 16, 3, 8, 3, 0)):
    clip = M.synth.clip(W, H, n, clip_index=140 + n, scene_len=5)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    src = torch.from_numpy(np.ascontiguousarray(clip)).pin_memory().numpy()
    enc.set_option("batch_frames", batch)
    out = np.zeros(len(want) + 4096, np.uint8)
    pos = 0
    for k in range(0, n, per_push):
        if pull == 2:
            pos += enc.push_frames_pull(W // 16, H // 16, pf, src[k:k + per_push], out, pos)[0]
        else:
            enc.push_frames(W // 16, H // 16, pf, src[k:k + per_push])
            if pull == 1:
                pos += enc.pull_into(out, pos)[0]
    enc.sequence_stop()
    last = False
    while not last:
        m, last = enc.pull_into(out, pos)
        pos += m
    assert out[:pos].tobytes() == want, (W, H, batch, per_push, pull)
enc.close()
print("SECONDS %%.2f" %% (time.perf_counter() - t0))
''' % ROOT
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    t0 = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    secs = float(r.stdout.split("SECONDS")[1])
    assert secs < 8.0, "the port path stalled (%.1f s for four small sequences)" % secs
