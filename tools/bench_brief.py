#!/usr/bin/env python3
"""Prints the headline numbers of a bench.py JSON line read from stdin (helper for quick GPU iterations):
    python bench.py | python tools/bench_brief.py      or      python tools/bench_brief.py bench.json"""
import json
import sys

text = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line = [l for l in text.splitlines() if l.startswith("{")]
if not line:
    print("no JSON line on stdin")
    sys.exit(1)
d = json.loads(line[-1])
pc = d.get("parity_check") or {}
print("MPixels/s", d["value"], "ms/step", d["ms_per_step"], "| profiled pass", (d.get("profiled_pass") or {}).get("value"),
      d.get("kernel_ms_per_step"), "| parity", pc.get("identical_to_oracle"), pc.get("gops_compared"), pc.get("problems"),
      "| roofline", d.get("roofline", {}).get("achieved"), d.get("roofline", {}).get("frac"),
      "| e2e", (d.get("end_to_end") or {}).get("value"), (d.get("end_to_end") or {}).get("identical_to_resident_stream"))
