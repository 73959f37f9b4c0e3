#!/bin/sh
# kernel iteration visit: parity suite, bench line, SQ counters (-> VALU instructions per macroblock), optional ubench
export TMPDIR=/tmp
TAG=${1:-r02_c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 1800 python3 -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
python3 bench.py --no-e2e > $OUT/bench.json 2> $OUT/bench.err
python3 tools/bench_brief.py < $OUT/bench.json
sh tools/pmc_sq.sh $OUT/sq > /dev/null 2>&1
cp $OUT/sq/summary.json $OUT/pmc_sq.json; rm -rf $OUT/sq
python3 -c "
import json
d=json.load(open('$OUT/pmc_sq.json'))
for k,v in d.items():
    if 'k_mb' in k:
        w=v['SQ_WAVES']
        print(k, 'VALU/wave %.1f SALU/wave %.1f LDS/wave %.1f  cycles/VALU %.2f  busy %.0f' % (v['SQ_INSTS_VALU']/w, v['SQ_INSTS_SALU']/w, v['SQ_INSTS_LDS']/w, 4*v['SQ_ACTIVE_INST_VALU']/v['SQ_INSTS_VALU'], v['SQ_BUSY_CYCLES']))
"
if [ "$2" = "ubench" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_cost.hip -o /tmp/issue_cost > /dev/null 2>&1 && /tmp/issue_cost > $OUT/issue_cost.txt 2>&1
  cat $OUT/issue_cost.txt
fi
