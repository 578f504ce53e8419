#!/usr/bin/env python3
"""Diagnostic (GPU; produced profiles/r4_wsm_store_hazard.txt): pattern of the values that differ in one layer (tap) between the two conv forms."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import product_model  # noqa: E402
from oracle.weights import synthetic_frames  # noqa: E402

DEV = "cuda:0"
layer, Cc, Hh, Ww = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B, H, W = 64, 240, 320
model, _ = product_model("S", False, 28)
x = torch.from_numpy(synthetic_frames(B, H, W, seed=21)).to(DEV)
with torch.no_grad():
    model(x[:1])
eng = model._engine


def tap():
    buf = torch.zeros(B * Cc * Hh * Ww, device=DEV)
    assert eng.lib.kp2d_set_tap(eng.handle, layer.encode(), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel())) == 0
    with torch.no_grad():
        model(x)
    torch.cuda.synchronize()
    return buf.view(B, Cc, Hh, Ww).clone()


eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", -1)
ref = tap()
eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", 256)
for rep in range(6):
    got = tap()
    idx = torch.nonzero(got != ref)
    print(f"rep {rep}: {idx.shape[0]} differing", flush=True)
    if idx.shape[0]:
        for b in torch.unique(idx[:, 0]).tolist()[:4]:
            sel = idx[idx[:, 0] == b]
            cs, ys, xs = (torch.unique(sel[:, k]).tolist() for k in (1, 2, 3))
            print(f"   frame {b}: {sel.shape[0]} values; channels {cs[:20]}{'...' if len(cs) > 20 else ''} rows {ys} cols {xs}", flush=True)
