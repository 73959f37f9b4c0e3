#!/bin/sh
# One visit to the GPU box.  sh tools/gpu_visit.sh <tag> [what...]   -> gpurun_out/<tag>/
#   what: tests (whole -m gpu suite) | newtests (strips + bench launch only) | bench | strips | c2 | sq (SQ counters) | prof (full profile set)
export TMPDIR=/tmp
TAG=${1:-visit}
shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
for what in "$@"; do
  case $what in
    tests)
      timeout 2400 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -6 $OUT/pytest.log ;;
    newtests)
      timeout 1800 python3 -m pytest tests/test_gpu_strips.py tests/test_gpu_bench_launch.py -x -q -m gpu -s > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log; tail -12 $OUT/pytest_new.log ;;
    bench)
      python3 bench.py --no-e2e > $OUT/bench.json 2> $OUT/bench.err; python3 tools/bench_brief.py < $OUT/bench.json ;;
    benchfull)
      python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; python3 tools/bench_brief.py < $OUT/bench.json ;;
    strips)
      python3 bench.py --mode strips > $OUT/strips_bench_1gpu.json 2> $OUT/strips.err
      python3 -c "
import json; d=json.load(open('$OUT/strips_bench_1gpu.json'))
print('strips: %.0f MPix/s  %.3f ms/step  loop %s  exchange %s  kernels %s  parity %s' % (d['value'], d['ms_per_step'], d['config']['strip_loop'], d['exchange_ms_per_step'], d['kernel_ms_per_step'], d.get('parity_check',{}).get('identical_to_oracle')))" ;;
    c2)
      python3 bench.py --config c2 > $OUT/c2_bench.json 2> $OUT/c2.err
      python3 -c "
import json; d=json.load(open('$OUT/c2_bench.json'))
print('c2: %.0f MPix/s  %.3f ms/step  roofline %s  parity %s' % (d['value'], d['ms_per_step'], {k: d['roofline'][k] for k in ('achieved','frac','avg_launch_ms')}, d.get('parity_check',{}).get('identical_to_oracle')))" ;;
    sq)
      sh tools/pmc_sq.sh $OUT/sq > /dev/null 2>&1
      cp $OUT/sq/summary.json $OUT/pmc_sq.json; rm -rf $OUT/sq
      python3 -c "
import json
d=json.load(open('$OUT/pmc_sq.json'))
for k,v in d.items():
    if 'k_mb' in k:
        w=v['SQ_WAVES']
        print(k[:60], 'VALU/wave %.1f SALU/wave %.1f LDS/wave %.1f  cycles/VALU %.2f  busy %.0f' % (v['SQ_INSTS_VALU']/w, v['SQ_INSTS_SALU']/w, v['SQ_INSTS_LDS']/w, 4*v['SQ_ACTIVE_INST_VALU']/v['SQ_INSTS_VALU'], v['SQ_BUSY_CYCLES']))
" ;;
    prof)
      sh tools/profile_round.sh $TAG ;;
    *) echo "unknown step $what" ;;
  esac
done
ls $OUT
