#!/usr/bin/env python3
"""Diagnostic (GPU): is a forward with the warp-specialised multi-chunk conv form deterministic, and equal to the general kernels?"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import product_model  # noqa: E402
from oracle.weights import synthetic_frames  # noqa: E402

DEV = "cuda:0"
cfg, v3, B, H, W = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
minitems = int(sys.argv[6]) if len(sys.argv) > 6 else 8
model, _ = product_model(cfg, v3, 28)
x = torch.from_numpy(synthetic_frames(B, H, W, seed=21)).to(DEV)
with torch.no_grad():
    model(x[:1])
    eng = model._engine
    eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", -1)
    ref = {k: v.clone() for k, v in model(x).items()}
    eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", minitems)
    runs = [{k: v.clone() for k, v in model(x).items()} for _ in range(6)]
    torch.cuda.synchronize()
for k in ref:
    d = [float((r[k].float() - ref[k].float()).abs().max()) for r in runs]
    n = [int(((r[k] != ref[k])).sum()) for r in runs]
    print(f"lanes={os.environ.get('KP2D_LANES', 'default')} {k:6s} max|diff| per run {['%.2e' % v for v in d]} differing {n}", flush=True)
