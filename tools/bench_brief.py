#!/usr/bin/env python3
"""Runs bench.py and prints the headline numbers (helper for quick GPU iterations)."""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py"] + sys.argv[1:], capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:])
    sys.exit(1)
d = json.loads(line[-1])
print("MPixels/s", d["value"], "ms/step", d["ms_per_step"], d["kernel_ms_per_step"], d.get("parity_check"),
      "roofline", d["roofline"]["achieved"], d["roofline"]["frac"])
