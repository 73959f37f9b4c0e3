#!/usr/bin/env python3
"""GPU timeline of a WINDOW of the in-flight strip loop (tools/strip_solo_turns.py under rocprofv3 --kernel-trace): every kernel with
start / duration / queue, and how much of the window had at least one kernel running - what K sequences in flight from one thread look like
on the GPU (which handle's launches fill which other handle's tail).
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/strip_solo_turns.py --handles 3 --split 1 --ranks 4 --seconds 0.3
    python tools/strip_timeline_window.py DIR [window us, default 800] [us before the end of the trace, default 20000]"""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        n = r["Kernel_Name"]
        short = n.split("(")[0].replace("void ", "").replace("m2v::", "")[:40]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
rows.sort()
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 800e3
back = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 20000e3
# the loop's kernels only (the trace also holds the clip generator's and the warm-up's)
t1 = rows[-1][1] - back
t0 = t1 - win
seq = [r for r in rows if r[1] > t0 and r[0] < t1]
print("window of %.0f us, %d kernels" % (win / 1e3, len(seq)))
busy, cur_end = 0, t0
for s, e, n, q in seq:
    print("  %8.1f  %6.1f us  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
    s2, e2 = max(s, t0), min(e, t1)
    if e2 > cur_end:
        busy += e2 - max(s2, cur_end)
        cur_end = e2
print("at least one kernel running: %.1f of %.1f us (%.1f %%)" % (busy / 1e3, win / 1e3, 100.0 * busy / win))
# per 10 ms of the whole loop: sequences (k_assemble launches) and the busy share, to show the window is typical
asm = [r for r in rows if r[2].startswith("k_assemble")]
if len(asm) > 20:
    span = (asm[-1][1] - asm[10][0]) / 1e3
    print("k_assemble launches %d over %.1f ms = %.4f ms per sequence" % (len(asm) - 10, span / 1e3, span / 1e3 / (len(asm) - 11)))
