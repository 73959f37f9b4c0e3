# cold start: is the prewarm needed?  (first GPU work on a fresh box)
python bench.py --no-cpu-baseline --prewarm 0 --steps 5 --warmup 2 2>/dev/null | python tools/bench_brief.py
sleep 20
python bench.py --no-cpu-baseline --prewarm 0 --steps 5 --warmup 2 2>/dev/null | python tools/bench_brief.py
sleep 20
python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | python tools/bench_brief.py
python bench.py --no-cpu-baseline 2>/dev/null | python tools/bench_brief.py
