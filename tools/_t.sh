python3 tools/bench_lightglue.py --steps 100 --warmup 10 2>/dev/null | tail -1 | cut -c1-330
python3 tools/bench_lightglue.py --pairs 1 --steps 200 --warmup 20 2>/dev/null | tail -1 | cut -c1-330
python3 tools/bench_lightglue.py --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-330
python3 tools/bench_lightglue.py --pairs 1 --steps 20 --warmup 3 2>/dev/null | tail -1 | cut -c1-330
python3 bench.py --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | cut -c1-200
