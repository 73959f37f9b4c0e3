#!/bin/sh
# Full profiling pass on the GPU box: rocprofv3 kernel stats + HBM PMC (separate passes) + SQ PMC, summarised on the
# box (raw CSVs are large because torch's clip generator launches thousands of tiny kernels).
#   sh tools/profile_round.sh <tag> [git head]   -> gpurun_out/<tag>/{kernel_stats.csv,bench.json,pmc_hbm.json,pmc_sq.json,pmc_traffic.json,c2_*}
# (from the build container:  gpurun -- "sh tools/profile_round.sh r03_x $(git rev-parse HEAD)"; copy the files to profiles/<tag>_*
#  and pmc_traffic.json to profiles/pmc_traffic.json)
export TMPDIR=/tmp
TAG=${1:-prof}
HEAD=${2:--}
OUT=gpurun_out/$TAG
RAW=/tmp/profile_round_$TAG          # raw rocprofv3 output: large, stays out of gpurun_out (64 MiB come back from a visit)
rm -rf $RAW; mkdir -p $OUT $RAW
step() { echo "$(date +%T) $*" >> $OUT/progress.log; }
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1
step raw_stats
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/raw_stats -o s -- python3 bench.py --inflight 1 --split 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep -h '^{' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
python3 tools/summarize_rocprof.py $RAW/raw_stats/s_kernel_stats.csv $OUT/kernel_stats.csv
step raw_f
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/raw_f -o f -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > /dev/null 2>&1
step raw_w
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/raw_w -o w -- python3 bench.py --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > /dev/null 2>&1
python3 tools/summarize_pmc.py $RAW/raw_f $RAW/raw_w > $OUT/pmc_hbm.json
# config c2 (640x480, I frames only): kernel stats and HBM counters of the I-frame kernel
step raw_c2
timeout 420 rocprofv3 --kernel-trace --stats --output-format csv -d $RAW/raw_c2 -o s -- python3 bench.py --config c2 --inflight 1 --split 1 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/c2_bench_under_rocprof.log 2>&1
python3 tools/summarize_rocprof.py $RAW/raw_c2/s_kernel_stats.csv $OUT/c2_kernel_stats.csv
step raw_c2f
timeout 420 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $RAW/raw_c2f -o f -- python3 bench.py --config c2 --gops 128 --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > /dev/null 2>&1
step raw_c2w
timeout 420 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $RAW/raw_c2w -o w -- python3 bench.py --config c2 --gops 128 --inflight 1 --split 1 --steps 1 --warmup 1 --prewarm 0 --no-cpu-baseline --no-e2e --sustain 0 > /dev/null 2>&1
python3 tools/summarize_pmc.py $RAW/raw_c2f $RAW/raw_c2w > $OUT/c2_pmc_hbm.json
step sq
timeout 600 sh tools/pmc_sq.sh $OUT/sq > /dev/null 2>&1
cp $OUT/sq/summary.json $OUT/pmc_sq.json
python3 tools/make_pmc_traffic.py $OUT/pmc_hbm.json $OUT/c2_pmc_hbm.json $HEAD $OUT/pmc_sq.json > $OUT/pmc_traffic.json
rm -rf $RAW $OUT/sq
cp $OUT/pmc_traffic.json profiles/pmc_traffic.json
step bench
timeout 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
step bench_c2
timeout 600 python3 bench.py --config c2 > $OUT/c2_bench.json 2> $OUT/c2_bench.err
ls -la $OUT
