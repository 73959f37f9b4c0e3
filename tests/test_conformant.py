"""CPU: the oracle's conformant mode (NOT a mode of the reference) against the standard.

The RTL's reconstruction loop deviates from ISO/IEC 13818-2 in four places (fpga-mpeg2-encoder_amd/decoder.py lists them), so a
standard decoder drifts away from the encoder's reference frames inside a GOP.  With conformant=True the oracle - and the
GPU path's option "conformant", which tests/test_gpu_parity.py compares with it byte for byte - follows the standard
there; the proof is that the repo's decoder with every quirk switched OFF reproduces the encoder's reconstruction exactly."""
import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc


@pytest.fixture(scope="module")
def M():
    orc.build()
    return m2v_load.load()


def _frames(dec, n):
    return [np.concatenate([p.reshape(-1) for p in dec.frames[f]]) for f in range(n)]


@pytest.mark.parametrize("W,H,n,pf,VL,Q,idx", [(96, 80, 9, 8, 3, 2, 5), (64, 64, 6, 5, 1, 1, 6), (112, 64, 5, 2, 2, 4, 7),
                                              (64, 80, 4, 0, 3, 3, 8)])
def test_standard_decoder_reproduces_the_conformant_reconstruction(M, W, H, n, pf, VL, Q, idx):
    clip = M.synth.clip(W, H, n, clip_index=idx, scene_len=4)
    es, d = orc.encode(clip, W // 16, H // 16, pf, XL=6, YL=6, VL=VL, Q=Q, dump=True, conformant=True)
    rec = d["recon"].reshape(n, -1)
    std = _frames(M.decoder.decode(es, quirks=False), n)
    assert all(np.array_equal(std[f], rec[f]) for f in range(n)), "a standard decoder must not drift in conformant mode"
    # ... and the reference's own mode does drift under a standard decoder (that is what the option is for)
    es0, d0 = orc.encode(clip, W // 16, H // 16, pf, XL=6, YL=6, VL=VL, Q=Q, dump=True)
    rec0 = d0["recon"].reshape(n, -1)
    assert all(np.array_equal(a, b) for a, b in zip(_frames(M.decoder.decode(es0, quirks=True), n), rec0))
    if pf:
        std0 = _frames(M.decoder.decode(es0, quirks=False), n)
        assert any(not np.array_equal(a, b) for a, b in zip(std0, rec0))


def test_conformant_flag_is_reset_and_streams_differ(M):
    clip = M.synth.clip(64, 64, 4, clip_index=9)
    a = orc.encode(clip, 4, 4, 3, XL=4, YL=4)
    b = orc.encode(clip, 4, 4, 3, XL=4, YL=4, conformant=True)
    assert orc.lib().m2v_oracle_get_conformant() == 0
    assert a != b and orc.encode(clip, 4, 4, 3, XL=4, YL=4) == a


def test_packed_iso_average_identity():
    """csrc/m2v_kernels.hpp avg4x4_iso: (a+b+c+d+2)>>2 = lerp(s,t,1) + (la & lc & ~(s^t) & 1) with s=(a+b)>>1,
    t=(c+d)>>1, la=(a^b)&1, lc=(c^d)&1 - over every (a,b) x (c,d) class that matters: all (s, la) x (t, lc)."""
    s, la, t, lc = np.meshgrid(np.arange(256), np.arange(2), np.arange(256), np.arange(2), indexing="ij")
    ok = (2 * s + la <= 510) & (2 * t + lc <= 510)
    want = (2 * s + la + 2 * t + lc + 2) >> 2
    got = ((s + t + 1) >> 1) + (la & lc & (~(s ^ t) & 1))
    assert np.array_equal(got[ok], want[ok]) and got[ok].max() <= 255
