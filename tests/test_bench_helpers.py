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


def test_traffic_carries_the_tree_it_was_measured_on(tmp_path, monkeypatch):
    """roofline.traffic comes from separate PMC passes (profiles/pmc_traffic.json); the line says which tree they measured and
    whether the kernel source has changed since (`traffic_stale`)."""
    import hashlib
    import json
    import os
    head, ksha = bench.source_shas()
    src = os.path.join(bench.ROOT, "fpga-mpeg2-encoder_amd", "csrc", "m2v_kernels.hpp")
    assert ksha == bench.kernel_source_sha(src)
    # comments and white space do not count, code does
    text = open(src, encoding="utf-8").read()
    (tmp_path / "a.hpp").write_text(text + "\n// a reworded comment\n   /* and\n another */\n")
    (tmp_path / "b.hpp").write_text(text + "\nint x;\n")
    assert bench.kernel_source_sha(str(tmp_path / "a.hpp")) == ksha != bench.kernel_source_sha(str(tmp_path / "b.hpp"))
    (tmp_path / "profiles").mkdir()
    import bench_util
    monkeypatch.setattr(bench_util, "ROOT", str(tmp_path))
    assert bench.pmc_traffic("k_mb_p_bytes_per_launch") == {"traffic": None}            # no file: null, nothing invented
    monkeypatch.setattr(bench_util, "source_shas", lambda: (head, ksha))
    for recorded, stale in ((ksha, False), ("0" * 64, True), (None, True)):
        (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps({"k_mb_p_bytes_per_launch": 136489472, "head": "abc", "kernel_sha": recorded}))
        t = bench.pmc_traffic("k_mb_p_bytes_per_launch")
        assert t["traffic"] == 136489472 and t["traffic_stale"] is stale and t["traffic_measured_at"]["head"] == "abc"
        assert bench.pmc_traffic("k_mb_i_c2_bytes_per_launch") == {"traffic": None}     # a key the passes did not measure
    # the vector ALU's share of the launch's SIMD-cycles rides in the same file (tools/make_pmc_traffic.py)
    assert bench.pmc_valu_busy("k_mb_p_bytes_per_launch") == {}
    (tmp_path / "profiles" / "pmc_traffic.json").write_text(json.dumps({"kernel_sha": ksha, "valu_busy": {"k_mb_p_bytes_per_launch": 0.88, "source": "x", "kernel_sha": ksha}}))
    assert bench.pmc_valu_busy("k_mb_p_bytes_per_launch") == {"valu_busy": 0.88, "valu_busy_source": "x", "valu_busy_stale": False}
