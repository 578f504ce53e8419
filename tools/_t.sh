python3 -m pytest tests -x -q -m gpu -k "topk or select or keypoint or frontend or post" 2>&1 | tail -4 || exit 1
python3 tools/graph_latency.py --config S --iters 300 2>&1 | tail -2
python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 --warmup 5 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])'
python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 --warmup 5 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])'
