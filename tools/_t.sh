set -e
python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/t7.log 2>&1 || { tail -30 gpurun_out/t7.log; exit 1; }
tail -2 gpurun_out/t7.log
for r in 1 2 3; do
for v in 0 1; do
KP2D_WIDE=$v python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('WIDE=$v', d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'])"
done; done
KP2D_WIDE=0 python3 tools/layer_profile.py 2>/dev/null | grep -E "conv1b|conv2a|conv2b|conv3a|confBb|convs.8|forward wall"
KP2D_WIDE=1 python3 tools/layer_profile.py 2>/dev/null | grep -E "conv1b|conv2a|conv2b|conv3a|confBb|convs.8|forward wall"
