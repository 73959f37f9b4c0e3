#!/usr/bin/env python3
"""Per-kernel, per-dispatch averages of rocprofv3 --pmc runs (encoder kernels only).
    python tools/summarize_pmc.py DIR [DIR ...]      every *counter_collection.csv below the directories
FETCH_SIZE / WRITE_SIZE come out in KiB per dispatch; on gfx950 FETCH_SIZE counts 64 B per 128-B request on wide streaming
reads (x2 there, MI355X_MICROARCH.md HBM section; tools/profile_round.sh applies it and records what it measured)."""
import collections
import csv
import glob
import json
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "m2v::" not in r["Kernel_Name"]:
                continue
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            a = acc[name][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
out = {k: {c: round(v[1] / v[0], 1) for c, v in sorted(cs.items())} | {"dispatches": max(v[0] for v in cs.values())}
       for k, cs in acc.items()}
print(json.dumps(out, indent=1))
