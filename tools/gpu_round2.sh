#!/bin/sh
# issue-cost table, macroblock statistics of the bench clip, SQ / HBM counters of the current build
export TMPDIR=/tmp
TAG=${1:-r02_b}; export TAG
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_cost.hip -o /tmp/issue_cost > /dev/null 2>&1 && /tmp/issue_cost > $OUT/issue_cost.txt 2>&1
cat $OUT/issue_cost.txt
python3 tools/mb_stats.py 2>/dev/null | tee $OUT/mb_stats.txt
sh tools/pmc_sq.sh $OUT/sq > /dev/null 2>&1
cp $OUT/sq/summary.json $OUT/pmc_sq.json; rm -rf $OUT/sq
python3 -c "import json; d=json.load(open('$OUT/pmc_sq.json')); [print(k, v) for k, v in d.items() if 'k_mb' in k]"
