set -e
python -m pytest tests -m gpu -x -q > gpurun_out/t8.log 2>&1 || { tail -30 gpurun_out/t8.log; exit 1; }
tail -2 gpurun_out/t8.log
for r in 1 2; do
for v in 0 1; do
KP2D_PAR_HEADS=$v python3 bench.py --batch 1 --steps 300 --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PAR_HEADS=$v batch1', d['value'], d['ms_per_step'])"
KP2D_PAR_HEADS=$v python3 tools/bench_frontend.py --batch 1 --steps 2000 2>/dev/null | tail -1 | cut -c100-220
KP2D_PAR_HEADS=$v python3 bench.py --batch 4 --steps 100 --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('PAR_HEADS=$v batch4', d['value'], d['ms_per_step'])"
done; done
KP2D_PAR_HEADS=1 python3 bench.py --batch 1 --steps 300 --config S_A --v3 --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('V3 S_A batch1 par', d['value'], d['ms_per_step'])"
KP2D_PAR_HEADS=0 python3 bench.py --batch 1 --steps 300 --config S_A --v3 --no-cpu-baseline --no-precision-modes 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('V3 S_A batch1 serial', d['value'], d['ms_per_step'])"
