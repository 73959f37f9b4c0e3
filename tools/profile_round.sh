#!/bin/sh
# Full profiling pass on the GPU box: rocprofv3 kernel stats + HBM PMC (separate passes) + SQ PMC, summarised on the
# box (raw CSVs are large because torch's clip generator launches thousands of tiny kernels).
#   sh tools/profile_round.sh <tag>      -> gpurun_out/<tag>/{kernel_stats.csv,bench.json,pmc_hbm.json,pmc_sq.json}
export TMPDIR=/tmp
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw_stats -o s -- python3 bench.py --split 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep -h '^{' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
python3 tools/summarize_rocprof.py $OUT/raw_stats/s_kernel_stats.csv $OUT/kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/raw_f -o f -- python3 bench.py --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/raw_w -o w -- python3 bench.py --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline > /dev/null 2>&1
python3 tools/summarize_pmc2.py $OUT/raw_f $OUT/raw_w > $OUT/pmc_hbm.json
sh tools/pmc_sq.sh $OUT/sq > /dev/null 2>&1
cp $OUT/sq/summary.json $OUT/pmc_sq.json
rm -rf $OUT/raw_stats $OUT/raw_f $OUT/raw_w $OUT/sq
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
ls -la $OUT
