#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv, encoder kernels only.
FETCH_SIZE / WRITE_SIZE are in KiB per dispatch (rocprofv3); see MI355X_MICROARCH.md HBM section for
the gfx950 caveats (FETCH_SIZE counts 64 B per 128-B request on wide streaming reads: x2 there;
other access widths are uncalibrated)."""
import csv
import collections
import json
import sys

out = collections.OrderedDict()
for path in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if "m2v::" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = acc[(name, r["Counter_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    for (name, ctr), (n, tot) in sorted(acc.items()):
        out.setdefault(name, {})[ctr] = {"dispatches": n, "avg_KiB_per_dispatch": round(tot / n, 1)}
print(json.dumps(out, indent=1))
