python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "other_shapes or out_of_bounds" 2>&1 | tail -4
python3 -c "
import __graft_entry__ as g
g.smoke(); print('smoke ok')" 2>&1 | tail -2
