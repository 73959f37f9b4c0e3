#!/bin/sh
# TIME of k_mb truncated at its stop points (-DM2V_DEBUG library, option ablate = n << 8; results invalid): how long do the phases take, cumulatively?
#   1 loads + 4:2:0 + window staging   2 + full-pel search   3 + half-pel, decision, prediction   4 + forward transform   5 + quantisers   6 + IDCT / reconstruction
export TMPDIR=/tmp
CFG=${1:-c3}
for rep in 1 2; do
for n in ${STOPS:-1 2 3 4 5 6 4096}; do
  python3 bench.py --config $CFG --ablate $((n * 256)) --inflight 1 --split 1 --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$CFG stop %4d  ms/step %.3f  P %.3f  I %.3f' % ($n, d['ms_per_step'], k['k_mb_P'], k['k_mb_I']))"
done; done
