#!/usr/bin/env python3
"""Timeline of the encoder's kernels inside the last steps of a rocprofv3 --kernel-trace run of bench.py: when each launch
starts and ends relative to the step, on which queue, and how much of the step no encoder kernel was running.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline
    python tools/step_timeline.py DIR"""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "m2v::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("m2v::", "")[:28], r.get("Queue_Id", "?")))
rows.sort()
# steps are separated by the k_assemble launches
ends = [i for i, r in enumerate(rows) if r[2].startswith("k_assemble")]
if len(ends) < 3:
    sys.exit("not enough steps in the trace")
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2     # which step, counted from the end (bench.py's last steps are the profiled single-stream pass)
a, b = ends[-back - 1] + 1, ends[-back] + 1
step = rows[a:b]
t0 = step[0][0]
print("one step: %d launches, %.1f us from first start to last end" % (len(step), (step[-1][1] - t0) / 1e3))
busy, cur_end = 0, t0
for s, e, n, q in step:
    print("  %8.1f .. %8.1f us  %6.1f us  queue %s  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, q, n))
    if e > cur_end:
        busy += e - max(s, cur_end)
        cur_end = e
print("time with at least one encoder kernel running: %.1f us; gaps: %.1f us" % (busy / 1e3, (step[-1][1] - t0 - busy) / 1e3))
if b < len(rows):
    print("from this step's last kernel to the next step's first: %.1f us (host: end-of-call wait, the caller, the next call's plan)" % ((rows[b][0] - step[-1][1]) / 1e3))
    other = [r for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]))
             if step[-1][1] <= int(r["Start_Timestamp"]) <= rows[b][0] and "m2v::" not in r["Kernel_Name"]]
    for r in other:
        print("    in between: %s %.1f us" % (r["Kernel_Name"][:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
