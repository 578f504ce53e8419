set -e
L=$PWD/nano-vs-slam_amd/csrc/build_exp/ablate.so
for r in 1 2; do
for d in 0 1 2 4 8 16 32 9 11 15 ; do
KP2D_LIB=$L KP2D_DBG=$d python3 bench.py --no-cpu-baseline --no-precision-modes --steps 40 2>/dev/null | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('DBG=$d', d['ms_per_step'], d['roofline']['kernel_ms_per_step'].get('conv3x3_f16x3'))"
done; done
