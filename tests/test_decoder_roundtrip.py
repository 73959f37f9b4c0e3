"""The encoder's streams through an independent MPEG-2 decoder written from ISO/IEC 13818-2 (fpga-mpeg2-encoder_amd/decoder.py):
every bit must parse, the decoded modes/vectors must be the encoder's, and with the RTL's documented deviations
switched on the decoder must reproduce the encoder's own reconstruction exactly."""
import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()
dec = M.decoder


@pytest.mark.parametrize("W,H,n,pf,VL,Q,ci", [(64, 64, 3, 2, 3, 2, 80), (96, 64, 4, 3, 1, 1, 81), (64, 96, 3, 1, 2, 4, 82),
                                               (80, 64, 2, 0, 3, 3, 83)])
def test_decoder_reproduces_encoder_reconstruction(W, H, n, pf, VL, Q, ci):
    clip = M.synth.clip(W, H, n, clip_index=ci, scene_len=2)
    data, d = orc.encode(clip, W // 16, H // 16, pf, 7, 7, VL, Q, dump=True)
    out = dec.decode(data, quirks=True)
    assert (out.width, out.height) == (W, H) and len(out.frames) == n
    assert out.sequence["frame_rate_code"] == 2 and out.sequence["bit_rate"] == 10000 and out.sequence["profile_level"] == 0x44
    for f in range(n):
        pic = out.pictures[f]
        assert pic["temporal_reference"] == f % (pf + 1) and pic["type"] == (1 if f % (pf + 1) == 0 else 2)
        for mb, info in enumerate(out.mbs[f]):
            assert info["intra"] == (d["mb_inter"][f][mb] == 0)
            assert info["cbp"] == d["mb_cbp"][f][mb]
            if not info["intra"]:
                assert info["mv"] == (d["mb_mvx"][f][mb], d["mb_mvy"][f][mb])
        rec = np.concatenate([p.reshape(-1) for p in out.frames[f]])
        assert np.array_equal(rec, d["recon"][f]), "frame %d" % f
    assert len(out.gops) == (n + pf) // (pf + 1) and all(g["closed_gop"] == 1 for g in out.gops)


def test_conformant_decoder_parses_and_stays_close():
    """With the deviations OFF (plain ISO decoding) the stream still parses completely and the picture stays close to
    the source; the small drift is the documented cost of the RTL's non-standard rounding."""
    W, H, n, pf = 96, 64, 5, 4
    clip = M.synth.clip(W, H, n, clip_index=84)
    data, d = orc.encode(clip, 6, 4, pf, 7, 7, 3, 2, dump=True)
    iso = dec.decode(data, quirks=False)
    rtl = dec.decode(data, quirks=True)
    for f in range(n):
        src_y = clip[f, 0]
        assert dec.psnr(rtl.frames[f][0], src_y) > 30.0
        assert dec.psnr(iso.frames[f][0], src_y) > 28.0
        assert dec.psnr(iso.frames[f][0], rtl.frames[f][0]) > 30.0


def test_black_fill_and_degenerate_streams_parse():
    clip = M.synth.degenerate("checker", 64, 64, 2)
    data = orc.encode(clip, 4, 4, 1, 7, 7, 3, 2, nbeats=64 * 64 // 4 + 100)
    out = dec.decode(data, quirks=True)
    assert len(out.frames) == 2
    assert (out.frames[1][0][32:] == 0).mean() > 0.9          # the tail of the second frame was black-filled
