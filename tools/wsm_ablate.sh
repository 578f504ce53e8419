#!/bin/bash
# Phase ablations of conv3x3_f16x3_wsm_kernel (conv3x3_wsm.hip) on one box:  tools/wsm_ablate.sh OUT.txt [DBG ...]
# Needs the ablation build (python3 nano-vs-slam_amd/csrc/build.py --ablate).  KP2D_DBG bits: 1 no epilogue, 64 epilogue
# without its global stores, 2 no LDS commit of the input image, 32 no input loads, 16 no weight copies (LDS-DMA),
# 8 no MFMAs.  Results are wrong by design; only the per-layer times mean anything.
set -eu -o pipefail
OUT=$1; shift
BITS=("$@"); [ ${#BITS[@]} -gt 0 ] || BITS=(0 8 16 32 48 56 1 64 2 57 59)
export KP2D_LIB="$PWD/nano-vs-slam_amd/csrc/build_exp/libkp2d_ablate.so"
: > "$OUT"
for d in "${BITS[@]}"; do
  KP2D_DBG=$d timeout -k 10 120 python3 tools/layer_profile.py --reps 3 > "${OUT%.txt}_dbg$d.log" 2>&1
  python3 - "$d" "${OUT%.txt}_dbg$d.log" >> "$OUT" <<'PY'
import sys
d, f = sys.argv[1], sys.argv[2]
want = ["backbone.conv3b", "backbone.conv4b", "desc_head.convB", "desc_head.confAa", "seg_head.convs.1", "seg_head.convs.2", "seg_head.convs.4", "seg_head.convs.5"]
ms, tot = {}, 0.0
for ln in open(f):
    p = ln.split()
    if len(p) > 3 and p[0] in want:
        ms[p[0]] = float(p[2])
    if len(p) > 3 and "<wsm>" in p[1]:
        tot += float(p[2])
if d == "0" or not hasattr(sys, "_hdr"):
    pass
print(f"DBG={d:>3s}  wsm layers total {tot:.3f} ms | " + "  ".join(f"{k.split('.')[-1]} {ms.get(k, float('nan')):.3f}" for k in want), flush=True)
PY
done
cat "$OUT"
