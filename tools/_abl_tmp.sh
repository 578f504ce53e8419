set -e
L=$PWD/nano-vs-slam_amd/csrc/build_exp/skew.so
for r in 1 2; do
for d in 0 2000 5000 10000 15000 20000 30000; do
KP2D_LIB=$L KP2D_SKEW=$d python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('SKEW=$d', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_step'].get('conv3x3_f16x3'))"
done; done
