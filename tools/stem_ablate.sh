#!/bin/bash
# Phase ablations of the fused first layer (conv3x3_f16x3_ws_kernel<*, STEM>, conv3x3_f16.hip):  tools/stem_ablate.sh OUT.txt [DBG ...]
# Needs the ablation build (python3 nano-vs-slam_amd/csrc/build.py --ablate).  KP2D_DBG bits: 8 conv1b's MFMAs off, 1 conv1b's epilogue off,
# 32 no window loads, 512 no window split / LDS writes, 128 conv1a's MFMAs off, 1024 conv1a's epilogue + image writes off, 2048 the whole
# conv1a M-tile loop off.  Results are wrong by design; only the layer's time means anything.
set -eu -o pipefail
OUT=$1; shift
BITS=("$@"); [ ${#BITS[@]} -gt 0 ] || BITS=(0 8 9 2048 2056 2057 2601 128 1024 512 32)
export KP2D_LIB="$PWD/nano-vs-slam_amd/csrc/build_exp/libkp2d_ablate.so"
: > "$OUT"
for d in "${BITS[@]}"; do
  KP2D_DBG=$d timeout -k 10 120 python3 tools/layer_profile.py --reps 3 > "${OUT%.txt}_dbg$d.log" 2>&1
  ms=$(awk '$1 == "backbone.conv1b" {print $3}' "${OUT%.txt}_dbg$d.log")
  echo "DBG=$d conv1b(fused) $ms ms" | tee -a "$OUT"
done
