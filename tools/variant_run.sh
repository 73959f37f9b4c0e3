#!/bin/sh
# Times every ab_libs/lib_*.so (tools/variant_build.sh) on this box, round robin: P-kernel ms per sequence in the profiled pass.
#   usage (GPU box): sh tools/variant_run.sh [rounds]
N=${1:-3}
for i in $(seq $N); do
  for l in ab_libs/lib_*.so; do
    n=$(basename $l .so)
    M2V_LIB=$PWD/$l python3 tools/diag_variants.py profile+stats+$n 2>&1 | grep -v amdgpu.ids
  done
done
