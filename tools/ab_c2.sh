for i in 1 2 3 4; do
for L in A B; do
  if [ $L = A ]; then export M2V_LIB=$PWD/ab_libs/base.so; else unset M2V_LIB; fi
  python3 bench.py --config c2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$L c2: %.0f MPix/s  %.4f ms/step  I launch %.4f ms  parity %s' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d.get('parity_check',{}).get('identical_to_oracle')))"
done; done
