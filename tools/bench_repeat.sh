# the default bench line several times on one box: which submission form the warm-up probe picked, and both figures
for i in 1 2 3; do
  python3 bench.py --no-e2e --no-cpu-baseline $* 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value %.0f  %.3f ms/step  [%s]  in flight %.3f  blocking %.3f' % (d['value'], d['ms_per_step'], d['config']['submission'][:10], d['sequences_in_flight_loop']['ms_per_step'], d['one_synchronous_call_per_step']['ms_per_step']))"
done
