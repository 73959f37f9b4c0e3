#!/usr/bin/env python3
"""How much of bench.py's timed loop (two sequences in flight) runs with 0 / 1 / 2+ encoder kernels on the GPU, from a rocprofv3
--kernel-trace run.  Under the tracer kernels of different queues still overlap (it serialises nothing), but every dispatch is a little
slower: read the shares, not the absolute time.
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-e2e
    python tools/overlap_timeline.py DIR"""
import csv
import glob
import sys

rows = []
for path in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "m2v::" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("m2v::", "")[:12], r.get("Queue_Id", "?")))
rows.sort()
# the timed loop = the longest stretch in which kernels come from two queues alternately; take the middle third of all k_mb launches
# issued while two queues were active
queues = sorted({r[3] for r in rows})
print("queues seen:", queues)
ev = []
for s, e, n, q in rows:
    ev.append((s, 1, n, q))
    ev.append((e, -1, n, q))
ev.sort()
# windows of 20 ms: report those where two queues are active
t_first = rows[0][0]
win = 5_000_000
stats = {}
active = 0
last = ev[0][0]
for t, d, n, q in ev:
    w = (last - t_first) // win
    st = stats.setdefault(w, [0, 0, 0, set()])
    st[min(active, 2)] += t - last
    st[3].add(q)
    active += d
    last = t
print("per 5 ms window: share of time with 0 / 1 / 2+ encoder kernels running   (queues active)")
for w in sorted(stats):
    a = stats[w]
    tot = a[0] + a[1] + a[2]
    if tot > 0.5 * win:
        print("  window %3d: %5.1f %% / %5.1f %% / %5.1f %%   %s" % (w, 100 * a[0] / tot, 100 * a[1] / tot, 100 * a[2] / tot, sorted(a[3])))
