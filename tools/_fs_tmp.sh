set -e
python -m pytest tests/test_gpu_parity.py tests/test_lightglue_gpu.py -m gpu -x -q -k "SA or NA or DA or attention or lightglue or taps" > gpurun_out/t6.log 2>&1 || { tail -30 gpurun_out/t6.log; exit 1; }
tail -3 gpurun_out/t6.log
bash tools/ab_variants.sh gpurun_out/ab_att_tr.jsonl --config S_A --v3 --n-classes 19 --height 480 --width 640 --batch 32 -- nano-vs-slam_amd/csrc/build_exp/headdot.so nano-vs-slam_amd/csrc/build_exp/att_tr.so
