"""CPU: bench.py's whole-stream parity check.  The benchmark clip is compared GOP by GOP with the oracle's streams of
the single GOPs (that is what lets the all-cores CPU baseline double as the checker); the stitching rules it relies on -
closed GOPs, the time code as the only position-dependent field, end code + final-word padding - are verified here
against the oracle encoding the whole multi-GOP sequence in one go."""
import numpy as np

import bench
import m2v_load
from oracle import m2v_oracle_ctypes as orc

M = m2v_load.load()


def test_per_gop_comparison_accepts_the_oracles_own_sequence_and_catches_damage():
    W, H, pf, ngops = 96, 64, 2, 5
    gop = pf + 1
    clip = M.synth.clip(W, H, ngops * gop, clip_index=140, scene_len=4)
    whole = orc.encode(clip, W // 16, H // 16, pf, 7, 7, 3, 2)
    per_gop = [orc.encode(clip[k * gop:(k + 1) * gop], W // 16, H // 16, pf, 7, 7, 3, 2) for k in range(ngops)]
    assert bench.compare_with_per_gop_oracle(whole, per_gop, gop) == []
    # one flipped bit anywhere must be reported
    rng = np.random.default_rng(1)
    for _ in range(40):
        b = bytearray(whole)
        pos = int(rng.integers(0, len(b)))
        b[pos] ^= 1 << int(rng.integers(0, 8))
        assert bench.compare_with_per_gop_oracle(bytes(b), per_gop, gop) != [], "flip at byte %d went unnoticed" % pos
    # a missing GOP, a swapped pair
    assert bench.compare_with_per_gop_oracle(whole, per_gop[:-1], gop) != []
    assert bench.compare_with_per_gop_oracle(whole, [per_gop[1], per_gop[0]] + per_gop[2:], gop) != []


def test_time_code_field_follows_the_frame_number():
    # 24 fps: frame 24 -> 1 s, frame 1440 -> 1 min, frame 86400 -> 1 h (RTL:2685-2698); marker bit and closed_gop set
    assert bench.gop_time_code(0) == bytes([0x00, 0x08, 0x00, 0x40])
    assert bench.gop_time_code(23) == (0x00080000 | (23 << 7) | 0x40).to_bytes(4, "big")
    assert bench.gop_time_code(24) == (0x00080000 | (1 << 13) | 0x40).to_bytes(4, "big")
    assert bench.gop_time_code(1440) == (0x00080000 | (1 << 20) | 0x40).to_bytes(4, "big")
    assert bench.gop_time_code(86400) == (0x00080000 | (1 << 26) | 0x40).to_bytes(4, "big")
