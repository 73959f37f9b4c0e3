#!/usr/bin/env python3
"""GPU timeline of ONE m2v_strip_encode sequence from a rocprofv3 --kernel-trace run of tools/strip_solo.py: every kernel (the
encoder's, RCCL's, the runtime's copy kernels) with start / end relative to the sequence, its queue, and the gaps in which nothing ran.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/strip_solo.py --world 8 --graph 0 --steps 3
    python tools/strip_timeline.py DIR [sequence counted from the end, default 3]"""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        short = n.split("(")[0].replace("void ", "").replace("m2v::", "")[:44]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
rows.sort()
ends = [i for i, r in enumerate(rows) if r[2].startswith("k_strip_assemble") or (r[2].startswith("k_assemble"))]
# a sequence ends with k_assemble (ranks that do not own the output) or k_strip_assemble (the output rank): cut at the LAST of either per call
cuts = []
for i in ends:
    if cuts and i - cuts[-1] < 6 and rows[i][2].startswith("k_strip_assemble"):
        cuts[-1] = i
    else:
        cuts.append(i)
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = cuts[-back - 1] + 1, cuts[-back] + 1
seq = rows[a:b]
t0 = seq[0][0]
print("one sequence: %d kernels, %.1f us from first start to last end" % (len(seq), (seq[-1][1] - t0) / 1e3))
busy, cur_end = 0, t0
for s, e, n, q in seq:
    gap = (s - cur_end) / 1e3
    print("  %8.1f .. %8.1f us  %6.1f us  q%-3s %s%s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n, "   <- %.1f us idle before" % gap if gap > 0.5 else ""))
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
print("at least one kernel running: %.1f us; idle: %.1f us" % (busy / 1e3, (seq[-1][1] - t0 - busy) / 1e3))
