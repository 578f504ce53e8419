python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
python3 tools/layer_profile.py 2>/dev/null | grep -E "confBb|convs.8 |forward wall"
python3 tools/layer_profile.py 2>/dev/null | grep -E "confBb|convs.8 |forward wall"
for i in 1 2 3; do python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 --warmup 5 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])'; done
