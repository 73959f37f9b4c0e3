export TMPDIR=/tmp
OUT=gpurun_out/r04_d; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
timeout 1500 python3 -m pytest tests/test_gpu_strip_graph.py tests/test_gpu_multidevice.py tests/test_gpu_strips.py tests/test_gpu_bench_launch.py tests/test_gpu_errors.py -x -q -m gpu > $OUT/pytest_new.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_new.log; tail -4 $OUT/pytest_new.log
for p in 0 1 0 1; do
M2V_STREAM_PRIORITY=$p timeout 600 python3 tools/strip_solo.py --world 8 --rccl 1 --graph 0 > $OUT/solo_prio$p.jsonl 2>/dev/null
python3 -c "
import json
for l in open('$OUT/solo_prio$p.jsonl'):
    d=json.loads(l); print('prio $p', d['world'], d['rank'], d['transport'], 'graph', d['strip_graph'], 'ms', d['ms_per_sequence'], 'host/step', d['host_us_per_gop_step_timed'])"
done
