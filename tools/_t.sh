python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 || exit 1
for d in 1 0 1 0; do
echo "DEEP=$d $(KP2D_DEEP=$d python3 tools/graph_latency.py --config S --iters 300 2>&1 | tail -1)"
done
KP2D_DEEP=1 python3 tools/bench_frontend.py 2>/dev/null | tail -2
KP2D_DEEP=0 python3 tools/bench_frontend.py 2>/dev/null | tail -2
