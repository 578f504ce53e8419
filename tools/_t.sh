python3 -m pytest tests/test_lightglue_gpu.py -x -q -m gpu 2>&1 | tail -4
