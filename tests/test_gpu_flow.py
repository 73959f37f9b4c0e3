"""-m gpu: option "flow" of the resident entry - every P frame of a chunk in ONE launch, macroblock rows handed from frame to frame
through completion counters (k_mb<.., FLOW>) - against the oracle and against the step-by-step launches, over the shapes that
change the plan: many GOPs, one GOP, GOPs of two frames, a short last GOP, GOPs that straddle chunk boundaries, all three
search ranges, the smallest frame (64x64: a cache line of the reference holds two pixel rows) and a wide one.  Plus the hand-off
under uneven load (two handles at once, hundreds of sequences, every stream compared) and the give-up path."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _encode(M, enc, d_in, n, W, H, pf, out):
    nb = enc.encode_resident(d_in.data_ptr(), n, out.data_ptr(), out.numel(), W // 16, H // 16, pf)
    return out[:nb].cpu().numpy().tobytes()


@pytest.mark.parametrize("W,H,n,pf,VL,batch", [
    (160, 96, 27, 8, 3, None),        # three GOPs
    (160, 96, 9, 8, 3, None),         # ONE GOP: every step depends on the one before it, nothing else to run
    (96, 160, 12, 1, 2, None),        # GOPs of two frames: every P frame's reference is an I frame (no in-launch hand-off at all)
    (128, 128, 23, 5, 1, None),       # short last GOP
    (128, 96, 30, 8, 3, 12),          # chunks of 12 frames: GOPs straddle the chunk boundaries (a step-0 P frame with a persisted reference)
    (64, 64, 19, 6, 3, None),         # smallest geometry
    (1920, 64, 10, 9, 3, None),       # wide: 120 macroblocks per row, 4 rows
    (320, 240, 40, 255, 3, None),     # i_pframes_count = 255: one I frame, 39 dependent steps
])
def test_flow_equals_step_by_step_equals_oracle(W, H, n, pf, VL, batch):
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    clip = M.synth.clip(W, H, n, clip_index=400 + n, scene_len=11)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, 2)
    d_in = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    out = torch.zeros(n * W * H * 3 + (1 << 16), dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    enc = M.Mpeg2Encoder(7, 7, VL, 2)
    try:
        if batch:
            enc.set_option("batch_frames", batch)
        for k in range(3):
            assert _encode(M, enc, d_in, n, W, H, pf, out) == want, "flow, call %d" % k
            assert enc.flow_state() == (True, 0)
        enc.set_option("flow", 0)
        assert _encode(M, enc, d_in, n, W, H, pf, out) == want
        assert enc.flow_state() == (False, 0)
        enc.set_option("flow", 1)
        assert _encode(M, enc, d_in, n, W, H, pf, out) == want and enc.flow_state() == (True, 0)
    finally:
        enc.close()


def test_flow_hand_off_under_uneven_load():
    """Two handles, two different clips, their sequences in flight at the same time for a few hundred rounds: the blocks of two FLOW
    launches share the GPU unevenly, a consumer's CU has just served other frames (L1 warm) - every stream of every round is
    compared with the first one, which is the oracle's."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    shapes = [(640, 480, 18, 8), (352, 288, 27, 2)]
    clips = [M.synth.clip(W, H, n, clip_index=420 + i, scene_len=7) for i, (W, H, n, pf) in enumerate(shapes)]
    wants = [orc.encode(c, W // 16, H // 16, pf, 7, 7, 3, 2) for c, (W, H, n, pf) in zip(clips, shapes)]
    d_in = [torch.from_numpy(np.ascontiguousarray(c)).to("cuda:0") for c in clips]
    outs = [torch.zeros(8 << 20, dtype=torch.uint8, device="cuda:0") for _ in shapes]
    refs = [torch.from_numpy(np.frombuffer(w, np.uint8).copy()).to("cuda:0") for w in wants]
    torch.cuda.synchronize()
    encs = [M.Mpeg2Encoder(7, 7, 3, 2) for _ in shapes]
    try:
        for rnd in range(300):
            for h, (W, H, n, pf) in enumerate(shapes):
                encs[h].encode_resident_begin(d_in[h].data_ptr(), n, outs[h].data_ptr(), outs[h].numel(), W // 16, H // 16, pf)
            for h in range(2):
                nb = encs[h].encode_resident_end()
                assert nb == len(wants[h]) and torch.equal(outs[h][:nb], refs[h]), "round %d, handle %d" % (rnd, h)
                outs[h][:nb].zero_()
            torch.cuda.synchronize()
        assert all(e.flow_state() == (True, 0) for e in encs)
    finally:
        for e in encs:
            e.close()


def test_flow_gives_up_and_the_sequence_is_encoded_again_step_by_step():
    """the -DM2V_DEBUG library, option ablate bit 5: no block of the FLOW launch ever sees its reference rows complete.  The blocks give
    up (~50 ms), the host notices, encodes the sequence again step by step - the bytes are the oracle's - and stays step by step."""
    import torch
    import m2v_load
    from oracle import m2v_oracle_ctypes as orc
    M = m2v_load.load()
    W, H, n, pf = 160, 96, 18, 8
    clip = M.synth.clip(W, H, n, clip_index=430)
    want = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    d_in = torch.from_numpy(np.ascontiguousarray(clip)).to("cuda:0")
    out = torch.zeros(4 << 20, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    enc = M.Mpeg2Encoder(7, 7, 3, 2, debug=True)
    try:
        assert _encode(M, enc, d_in, n, W, H, pf, out) == want and enc.flow_state() == (True, 0)
        enc.set_option("ablate", 32)
        assert _encode(M, enc, d_in, n, W, H, pf, out) == want
        assert enc.flow_state() == (False, 1)
        assert _encode(M, enc, d_in, n, W, H, pf, out) == want and enc.flow_state() == (False, 1)      # no second attempt
        enc.set_option("ablate", 0)
        enc.set_option("flow", 2)                                                                       # re-armed
        assert _encode(M, enc, d_in, n, W, H, pf, out) == want and enc.flow_state() == (True, 1)
        # the same through the two halves of the call
        enc.set_option("ablate", 32)
        enc.encode_resident_begin(d_in.data_ptr(), n, out.data_ptr(), out.numel(), W // 16, H // 16, pf)
        nb = enc.encode_resident_end()
        assert out[:nb].cpu().numpy().tobytes() == want and enc.flow_state() == (False, 2)
    finally:
        enc.close()
