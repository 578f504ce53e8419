#!/bin/bash
# Phase ablations of conv3x3_f16x3_s16_kernel (conv3x3_s16.hip) on one box:  tools/s16_ablate.sh OUT.txt [DBG ...]
# Needs the ablation build (python3 nano-vs-slam_amd/csrc/build.py --ablate).  KP2D_DBG bits: 1 no epilogue, 64 epilogue without
# its global stores, 32 no image copies (LDS-DMA of out-of-range addresses: zeros, no traffic), 8 no MFMAs.
# Results are wrong by design; only the per-layer times mean anything.
set -eu -o pipefail
OUT=$1; shift
BITS=("$@"); [ ${#BITS[@]} -gt 0 ] || BITS=(0 8 32 40 1 64 41)
export KP2D_LIB="$PWD/nano-vs-slam_amd/csrc/build_exp/libkp2d_ablate.so"
: > "$OUT"
for d in "${BITS[@]}"; do
  KP2D_DBG=$d timeout -k 10 120 python3 tools/layer_profile.py --reps 3 > "${OUT%.txt}_dbg$d.log" 2>&1
  python3 - "$d" "${OUT%.txt}_dbg$d.log" >> "$OUT" <<'PY'
import sys
d, f = sys.argv[1], sys.argv[2]
want = ["backbone.conv1b", "backbone.conv2a", "backbone.conv2b", "backbone.conv3a", "backbone.conv3b"]
ms = {}
for ln in open(f):
    p = ln.split()
    if len(p) > 3 and p[0] in want:
        ms[p[0]] = (float(p[2]), p[1])
print(f"DBG={d:>3s} | " + "  ".join(f"{k.split('.')[-1]} {ms.get(k, (float('nan'), ''))[0]:.3f}" for k in want), flush=True)
PY
done
cat "$OUT"
