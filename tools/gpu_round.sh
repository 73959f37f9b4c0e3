#!/bin/sh
# One GPU-box visit of a development round: parity suite, the bench line, stream-count sweep, the VALU cost table.
#   sh tools/gpu_round.sh <tag> [pytest-args]     -> gpurun_out/<tag>/
export TMPDIR=/tmp
TAG=${1:-round}
OUT=gpurun_out/$TAG
mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
timeout 2400 python3 -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -25 $OUT/pytest.log
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
python3 tools/bench_brief.py < $OUT/bench.json
for S in 1 2 3 5; do
  python3 bench.py --split $S --no-cpu-baseline --steps 60 2>/dev/null | python3 tools/bench_brief.py | sed "s/^/split=$S /" | tee -a $OUT/split_sweep.txt
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_cost.hip -o /tmp/valu_cost && /tmp/valu_cost > $OUT/valu_cost.txt 2>&1
tail -30 $OUT/valu_cost.txt
