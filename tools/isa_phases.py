#!/usr/bin/env python3
"""Static instruction mix of k_mb<3,true> per source phase (needs hipcc; compiles with line tables).
usage: python tools/isa_phases.py [mangled-substring]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fpga-mpeg2-encoder_amd", "csrc")
want = sys.argv[1] if len(sys.argv) > 1 else "k_mbILi3ELb1"
tmp = tempfile.mkdtemp()
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fwrapv", "-fPIC",
                       "-gline-tables-only", "-c", os.path.join(SRC, "m2v_launch.hip"), "-save-temps", "-o", "x.o"],
                      cwd=tmp, stderr=subprocess.DEVNULL)
s = open(os.path.join(tmp, "m2v_launch-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
name = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if want in m][0]
i = s.index(name + ":")
body = s[i:s.index("s_endpgm", i)].split("\n")
files = {int(m.group(1)): (m.group(3) or m.group(2)) for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
src = open(os.path.join(SRC, "m2v_kernels.hpp")).read().split("\n")
kstart = next(n for n, l in enumerate(src) if "void k_mb(" in l) + 1
kend = next(n for n, l in enumerate(src) if "k_slice_scan: one block" in l)
marks = [(n + 1, l.strip()[:80]) for n, l in enumerate(src) if kstart <= n < kend and l.strip().startswith("// ----")]
cur = (0, 0)
agg = collections.defaultdict(collections.Counter)
helpers = collections.Counter()
ops = collections.Counter()
for line in body:
    t = line.strip()
    m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if not t or t.startswith((".", ";", "_")) or t.endswith(":"):
        continue
    op = t.split()[0]
    ops[op] += 1
    k = "v" if op.startswith("v_") else "s" if op.startswith("s_") else "ds" if op.startswith("ds_") else "mem"
    f = files.get(cur[0], "?")
    if "m2v_kernels" not in f:
        ph = "(hip headers)"
    elif cur[1] < kstart:
        ph = "(helpers above k_mb)"
        if k == "v":
            helpers[src[cur[1] - 1].strip()[:60]] += 1
    else:
        ph = "pre"
        for n, txt in marks:
            if cur[1] >= n:
                ph = "%d %s" % (n, txt)
    agg[ph][k] += 1
print(name)
for ph in sorted(agg, key=lambda x: (not x[0].isdigit(), int(x.split()[0]) if x[0].isdigit() else 0)):
    c = agg[ph]
    print("%-95s v=%4d s=%4d ds=%3d mem=%3d" % (ph, c["v"], c["s"], c["ds"], c["mem"]))
print("total", sum(ops.values()), "valu", sum(v for k, v in ops.items() if k.startswith("v_")))
print("helper lines (VALU):", helpers.most_common(14))
print("top VALU ops:", [(k, v) for k, v in ops.most_common(60) if k.startswith("v_")][:28])
