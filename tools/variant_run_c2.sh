#!/bin/sh
# The same round robin as tools/variant_run.sh on config c2 (640x480, I frames only): MPixels/s and the I launch.
N=${1:-3}
for i in $(seq $N); do
  for l in ab_libs/lib_*.so; do
    M2V_LIB=$PWD/$l python3 bench.py --config c2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$(basename $l .so) c2: %.0f MPix/s  %.4f ms/step  I launch %.4f ms' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']))"
  done
done
