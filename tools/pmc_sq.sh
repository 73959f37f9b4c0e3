#!/bin/sh
# SQ-side PMC passes for the macroblock kernel (8 SQ counters per pass). Usage: tools/pmc_sq.sh <outdir>
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_sq}
RAW=/tmp/pmc_sq_raw_$$          # raw counter CSVs: large, kept out of gpurun_out
rm -rf $RAW; mkdir -p $OUT $RAW
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $RAW/p1 -o p -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > $OUT/p1.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM --output-format csv -d $RAW/p2 -o p -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > $OUT/p2.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU --output-format csv -d $RAW/p3 -o p -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > $OUT/p3.log 2>&1
timeout 420 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $RAW/p4 -o p -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > $OUT/p4.log 2>&1
python3 tools/summarize_pmc.py $RAW > $OUT/summary.json
rm -rf $RAW
