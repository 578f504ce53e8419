python3 -m pytest tests -x -q -m gpu -k "lightglue or lg_ or two_view or matcher" 2>&1 | tail -4 || exit 1
for p in 1 8 1 8; do
echo "pairs=$p $(python3 tools/bench_lightglue.py --pairs $p --steps 30 --warmup 5 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["matcher_ms_per_step"], d["value"])')"
done
