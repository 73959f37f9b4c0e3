#!/usr/bin/env python3
"""Regenerates tests/golden/*.  The reference ships no golden vectors, so these pins are:

  kat_gray_64x64.m2v / kat_black_64x64.m2v
      hand-derived bitstreams: assembled by hand from the RTL's constants, NOT
      produced by running any code; the sha256 values are fixed in tests/test_oracle_golden.py.
  oracle_hashes.json
      sha256 of the oracle's output on seeded synthetic clips: regression pins of the oracle itself
      (they freeze today's reading of the RTL; they are not evidence of parity with the RTL).

Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import m2v_load  # noqa: E402
from oracle import m2v_oracle_ctypes as orc  # noqa: E402

M = m2v_load.load()
HERE = os.path.dirname(os.path.abspath(__file__))

KAT_GRAY = (
    "000001b30400401209c42000000001b5144200010000000001b5230505050102"
    "0200000001b8000800400000010000080000000001b581111bc0000000010123"
    "94a5222e529488b94a5222e5294888000001022394a5222e529488b94a5222e5"
    "294888000001032394a5222e529488b94a5222e5294888000001042394a5222e"
    "529488b94a5222e5294888000001b70000000000000000000000000000000000")
KAT_BLACK = (
    "000001b30400401209c42000000001b5144200010000000001b5230505050102"
    "0200000001b8000800400000010000080000000001b581111bc0000000010123"
    "ff3ff4a5222e529488b94a5222e52948880000010223ff3ff4a5222e529488b9"
    "4a5222e52948880000010323ff3ff4a5222e529488b94a5222e5294888000001"
    "0423ff3ff4a5222e529488b94a5222e5294888000001b7000000000000000000")

CASES = {
    # name: (W, H, frames, clip_index, pframes, XL, YL, VL, Q, scene_len)
    "intra_640x480_q2": (640, 480, 2, 1, 0, 6, 5, 3, 2, 23),
    "ip_128x96_vl3_q2": (128, 96, 9, 2, 8, 7, 7, 3, 2, 23),
    "ip_160x128_vl1_q1": (160, 128, 5, 4, 3, 5, 5, 1, 1, 23),
    "ip_160x128_vl2_q3": (160, 128, 5, 5, 3, 5, 5, 2, 3, 23),
    "ip_160x128_vl3_q4": (160, 128, 5, 6, 3, 5, 5, 3, 4, 23),
    "ip_96x64_scenecuts": (96, 64, 14, 9, 3, 6, 6, 3, 2, 5),
}


def main():
    open(os.path.join(HERE, "kat_gray_64x64.m2v"), "wb").write(bytes.fromhex(KAT_GRAY))
    open(os.path.join(HERE, "kat_black_64x64.m2v"), "wb").write(bytes.fromhex(KAT_BLACK))
    hashes = {}
    for name, (W, H, n, ci, pf, XL, YL, VL, Q, sl) in CASES.items():
        clip = M.synth.clip(W, H, n, clip_index=ci, scene_len=sl)
        data = orc.encode(clip, W // 16, H // 16, pf, XL, YL, VL, Q)
        hashes[name] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data),
                        "input_sha256": hashlib.sha256(clip.tobytes()).hexdigest()}
    json.dump(hashes, open(os.path.join(HERE, "oracle_hashes.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(hashes, indent=1))


if __name__ == "__main__":
    main()
