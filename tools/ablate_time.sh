#!/bin/sh
# Time (not instruction counts) of the macroblock kernels with phases switched off in the -DM2V_DEBUG library (option ablate; results invalid):
# which phase is the time in?   usage: sh tools/ablate_time.sh [c2|c3]
export TMPDIR=/tmp
CFG=${1:-c2}
for rep in 1 2; do
for A in 1048576 1048580 1048592 1048596 1048584 1048604; do   # bit 20 = debug library; +4 no VLC, +16 no DCT/quant, +8 no IDCT/recon
  python3 bench.py --config $CFG --ablate $A --inflight 1 --split 1 --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); k=d['kernel_ms_per_step']; print('$CFG ablate bits %2d  ms/step %.3f  P %.3f  I %.3f' % ($A - 1048576, d['ms_per_step'], k['k_mb_P'], k['k_mb_I']))"
done; done
