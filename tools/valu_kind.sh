#!/bin/sh
# What does one more vector instruction of a given kind cost k_mb<3,true>?  The -DM2V_DEBUG library adds 64 independent instructions of
# one kind per macroblock behind the full-pel search (option ablate bits 16-19); the P-kernel time of bench.py's profiled pass against kind 0.
#   kinds: 1 v_add_u32 (VOP2, registers)  2 v_lshlrev_b32  3 v_mad_i32_i24 (VOP3)  4 v_add_u32 with an SGPR operand  5 v_ashrrev_i32  6 v_perm_b32
#          11 = 64 s_add_u32 (scalar ALU)   7 = 16 ds_read_b64 (not vector ALU: + 64 LDS data cycles)  8 = 8 global_load_dword from the lane table (+ 24 % vector memory instructions)
#   usage: sh tools/valu_kind.sh ["kinds"]     default "1 2 3 4 5 6"
export TMPDIR=/tmp
for rep in 1 2 3; do
for k in 0 ${1:-1 2 3 4 5 6} 0; do
  if [ $k = 0 ]; then A=1048576; else A=$((k * 65536)); fi     # kind 0: bit 20 = the debug library, no padding, no stop point
  python3 bench.py --ablate $A --inflight 1 --split 1 --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('kind $k  P kernel ms/sequence', d['kernel_ms_per_step']['k_mb_P'], ' ms/step (profiled pass)', d['profiled_pass']['ms_per_step'])"
done; done
