python3 -m pytest tests -x -q -m gpu 2>&1 | tail -3 || exit 1
for i in 1 2 3; do for v in 1 0; do
echo "FLAT=$v $(KP2D_FLAT=$v python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 --warmup 5 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"])')"
done; done
