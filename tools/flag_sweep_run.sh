#!/bin/sh
# On the GPU box: every library of ab_libs/ through the smoke check (bytes against the oracle) and two short bench runs (blocking single-handle form:
# the least noisy); the base build first and last.
export TMPDIR=/tmp
python3 -c "from oracle import m2v_oracle_ctypes as o; o.build()" > /dev/null 2>&1
for rep in 1 2; do
for n in base $(ls ab_libs/lib_*.so | sed 's#ab_libs/lib_##; s#\.so##' | grep -v '^base$') base; do
  ok=$(M2V_LIB=$PWD/ab_libs/lib_$n.so python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "smoke ok")
  M2V_LIB=$PWD/ab_libs/lib_$n.so python3 bench.py --inflight 1 --split 1 --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-10s smoke_ok=$ok  ms/step %.3f  P %.3f  I %.3f' % ('$n', d['ms_per_step'], d['kernel_ms_per_step']['k_mb_P'], d['kernel_ms_per_step']['k_mb_I']))"
done; done
