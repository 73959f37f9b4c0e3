#!/bin/sh
# Builds the library once per compiler-flag variant of the device unit (m2v_launch.hip) into ab_libs/ (travels to the GPU box with the tree;
# *.so is git-ignored).  tools/flag_sweep_run.sh times them there.
set -e
C=fpga-mpeg2-encoder_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fwrapv -fPIC -pthread -Wno-unused-function"
mkdir -p ab_libs/obj
for u in m2v_core m2v_port m2v_resident m2v_strips; do /opt/rocm/bin/hipcc $F -c -o ab_libs/obj/$u.o $C/$u.hip & done; wait
build() {  # name, extra flags...
  n=$1; shift
  if /opt/rocm/bin/hipcc $F "$@" -c -o ab_libs/obj/launch_$n.o $C/m2v_launch.hip 2> ab_libs/obj/$n.log; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ab_libs/lib_$n.so ab_libs/obj/launch_$n.o ab_libs/obj/m2v_core.o ab_libs/obj/m2v_port.o ab_libs/obj/m2v_resident.o ab_libs/obj/m2v_strips.o
    echo "built $n"
  else echo "FAILED $n: $(grep -m1 error ab_libs/obj/$n.log)"; fi
}
build base &
build maxilp -mllvm -amdgpu-sched-strategy=max-ilp &
build maxmem -mllvm -amdgpu-sched-strategy=max-memory-clause &
build bias100 -mllvm -amdgpu-schedule-metric-bias=100 &
wait
build bias0 -mllvm -amdgpu-schedule-metric-bias=0 &
build trackers -mllvm -amdgpu-use-amdgpu-trackers &
build nopost -mllvm -enable-post-misched=false &
build topdown -mllvm -misched-prera-direction=topdown &
wait
build o2 -O2 &
build nohighrp -mllvm -amdgpu-disable-unclustered-high-rp-reschedule &
build nocluster -mllvm -misched-cluster=false &
build nosink -mllvm -disable-machine-sink &
wait
ls -la ab_libs/*.so
