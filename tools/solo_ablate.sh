#!/bin/bash
# tools/solo_ablate.sh OUTDIR [ablate values...]: one rank of 8 alone (tools/strip_solo.py --peer 1) with parts of the peer hand-off switched off
# (-DM2V_DEBUG library, option ablate bits 22-25; results invalid, timing only).  0 = everything on, the shipped library.
out=${1:-gpurun_out/solo_ablate}; shift
mkdir -p $out
for ab in ${@:-0 16777216}; do
  echo "== ablate $ab" >> $out/solo_ablate.txt
  timeout 100 python tools/strip_solo.py --world 8 --peer 1 --graph 0 --ablate $ab 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['rank'], d['form'], d['ms_per_sequence'], d['kernel_ms'], d['peer'])" >> $out/solo_ablate.txt
done
