#!/bin/sh
# Same-box A/B of two builds of the library: box-to-box variation (+-1.5 %) is larger than most kernel tweaks.
#   A = ab_libs/base.so (a copy of an earlier build, made before editing: `mkdir -p ab_libs && cp fpga-mpeg2-encoder_amd/libm2v_mi355x.so ab_libs/base.so`; delete ab_libs/ after the session)
#   B = the in-tree library
# usage (on the GPU box): sh tools/ab.sh [rounds]
N=${1:-3}
for i in $(seq $N); do
  M2V_LIB=$PWD/ab_libs/base.so python tools/diag_variants.py profile+stats+A 2>&1 | grep -v amdgpu.ids
  python tools/diag_variants.py profile+stats+B 2>&1 | grep -v amdgpu.ids
done
