# the step timeline of bench.py's own timed loop at the default step count
export M2V_BENCH_TIMELINE=1
python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-900
python bench.py --no-cpu-baseline 2>&1 | grep -v amdgpu.ids | cut -c1-900
