set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/t9.log 2>&1 || { tail -30 gpurun_out/t9.log; exit 1; }
tail -2 gpurun_out/t9.log
for r in 1 2; do
for v in 0 1; do
KP2D_SHORT=$v python3 bench.py --batch 1 --steps 300 --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('SHORT=$v batch1', d['value'], d['ms_per_step'])"
KP2D_SHORT=$v python3 tools/bench_frontend.py --batch 1 --steps 2000 2>/dev/null | tail -1 | cut -c100-160
KP2D_SHORT=$v python3 bench.py --batch 2 --steps 200 --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('SHORT=$v batch2', d['value'], d['ms_per_step'])"
KP2D_SHORT=$v python3 bench.py --batch 1 --steps 300 --config S_A --v3 --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('SHORT=$v V3 S_A batch1', d['value'], d['ms_per_step'])"
done; done
