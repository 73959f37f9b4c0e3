"""The oracle against its pins: hand-derived known-answer bitstreams, header bytes quoted in
SURVEY.md 8-A.11, and regression hashes (tests/golden/make_golden.py)."""
import hashlib
import json
import os

import numpy as np
import pytest

import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

KAT_SHA = {
    "gray": "d108c0900596548e210dcced6b8a348258872691c1eaeaa9f05dac60c90b00fa",
    "black": "42b21ce6907fdbd3536b16632cdcfafe2bd3a807a3bbd8723c5fbc95a74b8837",
}


@pytest.mark.parametrize("kind", ["gray", "black"])
@pytest.mark.parametrize("XL,YL,VL", [(4, 4, 1), (6, 6, 3), (7, 5, 2)])
def test_hand_derived_kat(kind, XL, YL, VL):
    """64x64, one I frame, Q_LEVEL=2; independent of XL, YL, VECTOR_LEVEL."""
    f = M.synth.degenerate(kind, 64, 64, 1)
    data = orc.encode(f, 4, 4, 0, XL, YL, VL, 2)
    want = open(os.path.join(GOLD, "kat_%s_64x64.m2v" % kind), "rb").read()
    assert hashlib.sha256(want).hexdigest() == KAT_SHA[kind]
    assert data == want


def test_header_bytes_1920x1152():
    """Sequence / GOP / picture header bytes decoded field by field in SURVEY.md 8-A.11."""
    f = M.synth.degenerate("gray", 1920, 1152, 2)
    data = orc.encode(f, 120, 72, 1, 7, 7, 3, 2)
    seq = bytes.fromhex("00 00 01 B3 78 04 80 12 09 C4 20 00 00 00 01 B5 14 42 00 01 00 00 00 00 01 B5 23 05 05 05 "
                        "1E 02 24 00".replace(" ", ""))                 # 269 bits + 3 bits of padding = 34 bytes
    assert data[:34] == seq
    assert data[34:42] == bytes.fromhex("000001b800080040")          # first GOP header, closed_gop
    assert data[42:59] == bytes.fromhex("0000010000080000000001b581111bc000")   # I picture header + coding ext
    assert data[59:63] == bytes.fromhex("00000101") and (data[63] >> 2) == 0b001000   # slice 1, quantiser_scale_code 4
    p = data.find(bytes.fromhex("00000100005000038000"))                           # P picture header, temporal_reference 1
    assert p > 0 and data[p + 9:p + 18] == bytes.fromhex("000001b581111bc000")
    assert len(data) % 32 == 0 and data.rstrip(b"\x00").endswith(bytes.fromhex("000001b7"))


def test_regression_hashes():
    from golden import make_golden as mg
    pins = json.load(open(os.path.join(GOLD, "oracle_hashes.json")))
    for name, (W, H, n, ci, pf, XL, YL, VL, Q, sl) in mg.CASES.items():
        clip = M.synth.clip(W, H, n, clip_index=ci, scene_len=sl)
        assert hashlib.sha256(clip.tobytes()).hexdigest() == pins[name]["input_sha256"], "synthetic clip changed: " + name
        data = orc.encode(clip, W // 16, H // 16, pf, XL, YL, VL, Q)
        assert len(data) == pins[name]["bytes"] and hashlib.sha256(data).hexdigest() == pins[name]["sha256"], name


def test_timecode_and_gop_structure():
    """GOP header every pframes+1 frames carrying the frame number at 24 fps (RTL:2645-2656, 2685-2698)."""
    f = M.synth.degenerate("gray", 64, 64, 30)
    data = orc.encode(f, 4, 4, 4, 4, 4, 1, 2)
    gops, pos = [], 0
    while True:
        pos = data.find(b"\x00\x00\x01\xb8", pos)
        if pos < 0:
            break
        bits = int.from_bytes(data[pos + 4:pos + 8], "big")
        hh, mm, marker, ss, pic = bits >> 26, (bits >> 20) & 63, (bits >> 19) & 1, (bits >> 13) & 63, (bits >> 7) & 63
        assert marker == 1 and (bits >> 5) & 3 == 2
        gops.append((hh, mm, ss, pic))
        pos += 4
    assert gops == [(0, 0, n // 24, n % 24) for n in range(0, 30, 5)]
    assert data.count(b"\x00\x00\x01\x00") == 30


def test_stop_inside_frame_equals_black_fill():
    """i_sequence_stop mid-frame completes the frame with Y=0, U=V=0x80 (RTL:1036-1056)."""
    W, H = 96, 64
    clip = M.synth.clip(W, H, 3, clip_index=11)
    bpf = W * H // 4
    for cut in (1, 7, bpf // 2 + 5, bpf - 1):
        nbeats = 2 * bpf + cut
        got = orc.encode(clip, 6, 4, 2, 6, 6, 3, 2, nbeats=nbeats)
        filled = clip.copy().reshape(3, 3, H * W)
        filled[2, 0, cut * 4:] = 0
        filled[2, 1:, cut * 4:] = 0x80
        want = orc.encode(filled.reshape(3, 3, H, W), 6, 4, 2, 6, 6, 3, 2)
        assert got == want, cut
    assert orc.encode(clip, 6, 4, 2, 6, 6, 3, 2, nbeats=0) == b""      # stop while idle: nothing


def test_size_clamp():
    """RTL:985-991: sizes above 2^XL blocks clamp to 2^XL, below 4 clamp to 4; ports are XL+1 bits wide."""
    assert orc.geometry(100, 50, XL=6, YL=5) == (1024, 512)
    assert orc.geometry(0, 3, XL=6, YL=5) == (64, 64)
    assert orc.geometry(64, 32, XL=6, YL=5) == (1024, 512)
    assert orc.geometry(65, 33, XL=6, YL=5) == (1024, 512)
    assert orc.geometry(128 + 5, 5, XL=6, YL=5) == (80, 80)              # bit 7 does not exist on a 7-bit port
    assert M.clamp_geometry(100, 50, 6, 5) == (1024, 512) and M.clamp_geometry(0, 3, 6, 5) == (64, 64)
    f = M.synth.degenerate("gray", 64, 64, 1)
    assert orc.encode(f, 1, 2, 0, 4, 4, 1, 2) == orc.encode(f, 4, 4, 0, 4, 4, 1, 2)


def test_output_length_rule():
    """The stream is whole 32-byte words and one word is always emitted at the end (RTL:2932-2937)."""
    rng = np.random.default_rng(1)
    for n in range(1, 4):
        f = rng.integers(0, 256, (n, 3, 64, 64), dtype=np.uint8)
        data = orc.encode(f, 4, 4, 1, 4, 4, 1, 3)
        assert len(data) % 32 == 0
        end = data.rfind(b"\x00\x00\x01\xb7") + 4
        assert len(data) == (end // 32 + 1) * 32
