#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the HBM counter passes of tools/profile_round.sh: HBM bytes per launch of the dominant kernels
(FETCH_SIZE x 2 + WRITE_SIZE, KiB per dispatch -> bytes; the x 2 is the gfx950 correction of MI355X_MICROARCH.md, calibrated
on the I-frame kernel, which reads exactly its 3 B/px of input) TOGETHER WITH the tree they were measured on: the git HEAD (passed
in: the GPU box has no .git) and the sha256 of the kernel source.  bench.py compares that sha with the running tree and says
"traffic_stale" when they differ.
    python tools/make_pmc_traffic.py <c3 pmc_hbm.json> <c2 pmc_hbm.json or -> <git head or -> > pmc_traffic.json"""
import hashlib
import re
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel(d, needle):
    """the instantiation that did the work: every k_mb<VL, P> also exists as a one-wavefront FILL instantiation (it writes the lane table)"""
    hits = [v for name, v in d.items() if needle in name and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    return max(hits, key=lambda v: (v.get("dispatches", 0), v["FETCH_SIZE"])) if hits else None


c3 = json.load(open(sys.argv[1]))
c2 = json.load(open(sys.argv[2])) if len(sys.argv) > 2 and sys.argv[2] != "-" and os.path.exists(sys.argv[2]) else {}
head = sys.argv[3] if len(sys.argv) > 3 and sys.argv[3] != "-" else None
def kernel_source_sha(path):
    """the same as bench.py's: comments dropped, white space collapsed"""
    text = open(path, encoding="utf-8").read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    return hashlib.sha256(" ".join(text.split()).encode()).hexdigest()


ksha = kernel_source_sha(os.path.join(ROOT, "fpga-mpeg2-encoder_amd", "csrc", "m2v_kernels.hpp"))
out = {"head": head, "kernel_sha": ksha,
       "source": "tools/profile_round.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `bench.py --split 1 --steps 1 "
                 "--warmup 1` (c3: one launch = 10 P frames of 1920x1152; c2: `--gops 128`, one launch = 128 I frames of 640x480, doubled for bench.py's 256), KiB per dispatch; "
                 "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"}
p = kernel(c3, "k_mb<3, true")
if p:
    out.update({"k_mb_p_bytes_per_launch": round((2 * p["FETCH_SIZE"] + p["WRITE_SIZE"]) * 1024), "fetch_kib": p["FETCH_SIZE"], "write_kib": p["WRITE_SIZE"],
                "algorithmic_bytes_per_launch": round((7 * 6.0 + 4.5) / 8 * 10 * 1920 * 1152)})      # 8 P launches per step: 7 with 10 referenced frames, 1 with the GOPs' last frames
i = kernel(c2, "k_mb<1, false")
if i:
    # measured on a 128-frame launch (rocprofv3 --pmc segfaults in this image with the 256-frame clip), stated for bench.py's 256-frame launch
    out.update({"k_mb_i_c2_bytes_per_launch": round((2 * i["FETCH_SIZE"] + i["WRITE_SIZE"]) * 1024 * 2), "c2_fetch_kib_128_frames": i["FETCH_SIZE"],
                "c2_write_kib_128_frames": i["WRITE_SIZE"], "c2_algorithmic_bytes_per_launch": round(256 * 3.0 * 640 * 480)})
print(json.dumps(out, indent=1))
