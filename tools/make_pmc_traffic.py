#!/usr/bin/env python3
"""profiles/pmc_traffic.json from the HBM counter passes of tools/profile_round.sh: HBM bytes per launch of the dominant kernels
(FETCH_SIZE x 2 + WRITE_SIZE, KiB per dispatch -> bytes; the x 2 is the gfx950 correction of MI355X_MICROARCH.md, calibrated
on the I-frame kernel, which reads exactly its 3 B/px of input) TOGETHER WITH the tree they were measured on: the git HEAD (passed
in: the GPU box has no .git) and the sha256 of the kernel source.  bench.py compares that sha with the running tree and says
"traffic_stale" when they differ.
`valu_busy` (from the SQ counter pass, tools/pmc_sq.sh): the vector ALU's share of the dominant kernel's SIMD-cycles - the roof that
binds it.  Two denominators, both from counters of the same file, no clock assumed:
    valu_busy                  4 x SQ_ACTIVE_INST_VALU (quad-cycles -> cycles, summed over all SIMDs) / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs):
                               of the whole dispatch window, the profiler's serialised start and drain included
    valu_busy_while_resident   ... / (1024 x SQ_BUSY_CYCLES / 32 shader engines): of the cycles in which the shader engines held waves
    python tools/make_pmc_traffic.py <c3 pmc_hbm.json> <c2 pmc_hbm.json or -> <git head or -> [<c3 pmc_sq.json>] > pmc_traffic.json"""
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
sq = json.load(open(sys.argv[4])) if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) else {}


def sq_kernel(needle):
    hits = [v for name, v in sq.items() if needle in name and "SQ_ACTIVE_INST_VALU" in v and "SQ_BUSY_CYCLES" in v]
    return max(hits, key=lambda v: v.get("SQ_WAVES", 0)) if hits else None


vb = {}
for key, needle in (("k_mb_p_bytes_per_launch", "k_mb<3, true"), ("k_mb_i_c2_bytes_per_launch", "k_mb<1, false")):    # (I frames run the VL-independent instantiation: config c2's kernel, counted here on c3's I launches)
    k = sq_kernel(needle)
    if not k:
        continue
    simd_cycles_valu = 4.0 * k["SQ_ACTIVE_INST_VALU"]
    if k.get("GRBM_GUI_ACTIVE"):
        vb[key] = round(simd_cycles_valu / (1024.0 * k["GRBM_GUI_ACTIVE"] / 8.0), 4)
    vb[key + "_while_resident"] = round(simd_cycles_valu / (1024.0 * k["SQ_BUSY_CYCLES"] / 32.0), 4)
    vb[key + "_per_wave"] = {"valu_quad_cycles": round(k["SQ_ACTIVE_INST_VALU"] / k["SQ_WAVES"], 1), "vector_instructions": round(k["SQ_INSTS_VALU"] / k["SQ_WAVES"], 1),
                             "scalar_instructions": round(k["SQ_INSTS_SALU"] / k["SQ_WAVES"], 1), "lds_instructions": round(k["SQ_INSTS_LDS"] / k["SQ_WAVES"], 1)}
if vb:
    vb.update({"kernel_sha": ksha, "source": "tools/pmc_sq.sh: rocprofv3 --pmc SQ_* / GRBM_* passes of `bench.py --inflight 1 --split 1 --steps 1 --warmup 1`; "
                                            "4 x SQ_ACTIVE_INST_VALU / (1024 x GRBM_GUI_ACTIVE / 8), `_while_resident`: / (1024 x SQ_BUSY_CYCLES / 32)"})
    out["valu_busy"] = vb
print(json.dumps(out, indent=1))
