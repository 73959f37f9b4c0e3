#!/bin/sh
# Builds the library once per preprocessor variant of the device unit (m2v_launch.hip) into ab_libs/ (travels to the GPU box with the tree;
# *.so is git-ignored); tools/variant_run.sh times them on one box.
#   usage: sh tools/variant_build.sh name1=-DX=1 name2="-DX=2 -DY=3" ...
set -e
C=fpga-mpeg2-encoder_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fwrapv -fPIC -pthread -Wno-unused-function"
rm -rf ab_libs; mkdir -p ab_libs/obj
for u in m2v_core m2v_port m2v_resident m2v_strips; do /opt/rocm/bin/hipcc $F -c -o ab_libs/obj/$u.o $C/$u.hip & done; wait
for spec in "$@"; do
  n=${spec%%=*}; fl=${spec#*=}
  ( if /opt/rocm/bin/hipcc $F $fl -c -o ab_libs/obj/launch_$n.o $C/m2v_launch.hip 2> ab_libs/obj/$n.log; then
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ab_libs/lib_$n.so ab_libs/obj/launch_$n.o ab_libs/obj/m2v_core.o ab_libs/obj/m2v_port.o ab_libs/obj/m2v_resident.o ab_libs/obj/m2v_strips.o
      echo "built $n ($fl)"
    else echo "FAILED $n: $(grep -m1 error ab_libs/obj/$n.log)"; fi ) &
done; wait
rm -rf ab_libs/obj
