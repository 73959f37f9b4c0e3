"""-m gpu: the HIP path, through the C-ABI, bit-exact against the CPU oracle."""
import hashlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    import gpu_util
    return gpu_util


KAT = {
    "gray": "d108c0900596548e210dcced6b8a348258872691c1eaeaa9f05dac60c90b00fa",
    "black": "42b21ce6907fdbd3536b16632cdcfafe2bd3a807a3bbd8723c5fbc95a74b8837",
}


@pytest.mark.parametrize("kind", ["gray", "black"])
def test_kat_64x64(G, kind):
    f = G.M.synth.degenerate(kind, 64, 64, 1)
    data = G.resident_encode(f, 4, 4, 0, XL=6, YL=6)
    assert len(data) == 160
    assert hashlib.sha256(data).hexdigest() == KAT[kind]


def test_intra_only_stages(G):
    f = G.M.synth.clip(640, 480, 3, clip_index=1)
    assert G.compare_stages(f, 40, 30, 0, XL=6, YL=5, Q=2) == []


def test_ip_gop_stages_small(G):
    f = G.M.synth.clip(128, 96, 9, clip_index=2)
    assert G.compare_stages(f, 8, 6, 8) == []


@pytest.mark.parametrize("VL,Q", [(1, 1), (2, 3), (3, 4), (1, 4), (2, 2)])
def test_parameter_matrix(G, VL, Q):
    f = G.M.synth.clip(160, 128, 5, clip_index=3 + VL)
    assert G.compare_stages(f, 10, 8, 3, XL=5, YL=5, VL=VL, Q=Q) == []


@pytest.mark.parametrize("kind", ["checker", "noise"])
def test_degenerate_content(G, kind):
    f = G.M.synth.degenerate(kind, 96, 64, 4)
    assert G.compare_stages(f, 6, 4, 2, XL=4, YL=4) == []


def test_one_chunk_of_600_frames(G):
    """batch_frames used to be capped at 200 (32-bit offsets inside a chunk); k_frame_scan now scans 64-bit sums (as three
    32-bit DPP scans), so one chunk can be any length: 600 frames = 9600 slices, more than the 8192 items the scan's threads
    fetch in their first round trip"""
    from oracle import m2v_oracle_ctypes as orc
    f = G.M.synth.clip(64, 64, 600, clip_index=19, scene_len=7)
    want = orc.encode(f, 4, 4, 2, XL=6, YL=6)
    assert G.resident_encode(f, 4, 4, 2, XL=6, YL=6, batch_frames=600) == want
    assert G.resident_encode(f, 4, 4, 2, XL=6, YL=6, batch_frames=250) == want


def test_multi_gop_and_chunking(G):
    from oracle import m2v_oracle_ctypes as orc
    f = G.M.synth.clip(96, 64, 14, clip_index=9, scene_len=5)
    want = orc.encode(f, 6, 4, 3, XL=6, YL=6)
    for bf in (96, 4, 5, 1):
        got = G.resident_encode(f, 6, 4, 3, XL=6, YL=6, batch_frames=bf)
        assert got == want, "batch_frames=%d" % bf


@pytest.mark.parametrize("streams", [1, 2, 3, 5, 8])
def test_split_streams_option(G, streams):
    """Option "split_streams": the closed GOPs of a chunk run as N independent groups on N HIP streams (fills the tail of
    every launch; default 2).  Same bytes, resident and port path, more / fewer GOPs than streams, odd and even counts."""
    from oracle import m2v_oracle_ctypes as orc
    for n, pf in ((15, 2), (12, 3), (3, 4), (21, 1)):
        f = G.M.synth.clip(112, 80, n, clip_index=11)
        want = orc.encode(f, 7, 5, pf, XL=6, YL=6)
        enc = G.M.Mpeg2Encoder(6, 6, 3, 2, device=0)
        try:
            enc.set_option("split_streams", streams)
            assert G.resident_encode(f, 7, 5, pf, XL=6, YL=6, enc=enc) == want
            enc.set_option("batch_frames", 7)
            assert enc.encode(f, 7, 5, pf) == want
        finally:
            enc.close()


@pytest.mark.parametrize("cu_pack", [0, 1, 2, 3, 4, 6, 7, 8])
def test_cu_pack_option(G, cu_pack):
    """Option "cu_pack" only permutes which block of a launch takes which macroblock (default 5, what every other test runs
    with): the bytes must not depend on it - launches smaller than one span of the permutation, just beyond one, and a few
    thousand blocks with a ragged tail; an out-of-range value is refused."""
    from oracle import m2v_oracle_ctypes as orc
    for (W, H, n, pf) in ((64, 64, 5, 2), (272, 160, 7, 3), (1008, 528, 4, 1)):
        f = G.M.synth.clip(W, H, n, clip_index=17)
        want = orc.encode(f, W // 16, H // 16, pf, XL=6, YL=6)
        enc = G.M.Mpeg2Encoder(6, 6, 3, 2, device=0)
        try:
            enc.set_option("cu_pack", cu_pack)
            assert G.resident_encode(f, W // 16, H // 16, pf, XL=6, YL=6, enc=enc) == want, (W, H)
            with pytest.raises(Exception):
                enc.set_option("cu_pack", 9)
        finally:
            enc.close()


@pytest.mark.parametrize("mfma", [0, 1])
@pytest.mark.parametrize("kind", ["synth", "noise", "checker", "flat255vs0"])
def test_dct_variants_are_bit_identical(G, mfma, kind):
    """Option "dct_mfma": the luma 2-D DCT on the matrix cores (two chained i8 GEMMs, 19-bit intermediate as three byte
    limbs; default) or on the integer v_dot4 / v_mad_i32_i24 path - both must give the oracle's bytes, on ordinary content
    and on content that drives the intermediate to its extremes (white noise, full-swing checkerboards, 255 against 0),
    for intra and inter macroblocks and every Q_LEVEL."""
    from oracle import m2v_oracle_ctypes as orc
    W, H, n = 96, 80, 4
    if kind == "synth":
        f = G.M.synth.clip(W, H, n, clip_index=12, scene_len=2)
    elif kind == "noise":
        f = np.random.default_rng(7).integers(0, 256, (n, 3, H, W), dtype=np.uint8)
    elif kind == "checker":
        f = G.M.synth.degenerate("checker", W, H, n)
    else:
        f = np.zeros((n, 3, H, W), np.uint8)
        f[0::2] = 255                                   # 255 against a reconstruction of 0 and back: the largest residuals
    for Q in (1, 2, 3, 4):
        for pf in (0, 3):
            want = orc.encode(f, W // 16, H // 16, pf, XL=6, YL=6, VL=2, Q=Q)
            enc = G.M.Mpeg2Encoder(6, 6, 2, Q, device=0)
            try:
                enc.set_option("dct_mfma", mfma)
                assert G.resident_encode(f, W // 16, H // 16, pf, XL=6, YL=6, VL=2, Q=Q, enc=enc) == want, "Q=%d pf=%d" % (Q, pf)
            finally:
                enc.close()


@pytest.mark.parametrize("VL,Q,pf", [(3, 2, 8), (2, 1, 3), (1, 4, 5), (3, 3, 0)])
def test_conformant_option_matches_the_oracles_conformant_mode(G, VL, Q, pf):
    """Option "conformant" (ISO reconstruction loop: +2 rounding, chroma vector toward zero, truncating inverse quantiser
    with mismatch control) is NOT the reference's behaviour; it is checked against the oracle's own conformant mode,
    which tests/test_conformant.py ties to a standard decoder.  The default mode is untouched (all other tests)."""
    from oracle import m2v_oracle_ctypes as orc
    f = G.M.synth.clip(144, 112, 11, clip_index=13, scene_len=6)
    want = orc.encode(f, 9, 7, pf, XL=6, YL=6, VL=VL, Q=Q, conformant=True)
    base = orc.encode(f, 9, 7, pf, XL=6, YL=6, VL=VL, Q=Q)
    enc = G.M.Mpeg2Encoder(6, 6, VL, Q, device=0)
    try:
        enc.set_option("conformant", 1)
        assert G.resident_encode(f, 9, 7, pf, XL=6, YL=6, VL=VL, Q=Q, enc=enc) == want
        assert enc.encode(f, 9, 7, pf) == want
        enc.set_option("conformant", 0)
        assert G.resident_encode(f, 9, 7, pf, XL=6, YL=6, VL=VL, Q=Q, enc=enc) == base
    finally:
        enc.close()
    if pf:
        assert want != base
