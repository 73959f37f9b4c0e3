#!/bin/sh
# Memory-pipe PMC passes for the macroblock kernels (vector L1 = TCP, L2 = TCC) on a SMALL workload (tools/pmc_small.py), every pass bounded.
# Usage: tools/pmc_mem.sh <outdir> [cu_pack]
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_mem}
mkdir -p $OUT
B="python3 tools/pmc_small.py $2"
pass() { n=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$n -o p -- $B > $OUT/$n.log 2>&1; echo "pass $n rc=$?"; }
pass p1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum
pass p2 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
pass p3 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TA_BUSY_avr
python3 tools/summarize_pmc.py /tmp/pmc_p1 /tmp/pmc_p2 /tmp/pmc_p3 > $OUT/summary.json
rm -rf /tmp/pmc_p1 /tmp/pmc_p2 /tmp/pmc_p3
