#!/usr/bin/env python3
"""Diagnostic (GPU): where do the warp-specialised multi-chunk conv layers differ from the general kernels?
Reads every CBR output through kp2d_set_tap with the form off and forced on."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import product_model  # noqa: E402
from oracle.weights import synthetic_frames  # noqa: E402

DEV = "cuda:0"


def taps(model, x, layers):
    eng = model._engine
    out = {}
    for name in layers:
        buf = torch.zeros(x.shape[0] * 256 * (x.shape[2] // 2) * (x.shape[3] // 2), device=DEV)
        assert eng.lib.kp2d_set_tap(eng.handle, name.encode(), C.c_void_p(buf.data_ptr()), C.c_size_t(buf.numel())) == 0
        with torch.no_grad():
            model(x)
        torch.cuda.synchronize()
        out[name] = buf.clone()
    eng.lib.kp2d_set_tap(eng.handle, None, None, C.c_size_t(0))
    return out


def main():
    cfg, v3, B, H, W = sys.argv[1], sys.argv[2] == "1", int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    model, _ = product_model(cfg, v3, 28)
    x = torch.from_numpy(synthetic_frames(B, H, W, seed=21)).to(DEV)
    with torch.no_grad():
        model(x[:1])
    eng = model._engine
    layers = ["backbone.conv3a", "backbone.conv3b", "backbone.conv4a", "backbone.conv4b", "score_head.convDa", "desc_head.convA",
              "desc_head.confAa", "seg_head.convs.0", "seg_head.convs.1", "seg_head.convs.2", "seg_head.convs.4", "seg_head.convs.5",
              "seg_head.convs.6", "seg_head.convs.7", "vlad_head.convlad1"]
    eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", -1)
    ref = taps(model, x, layers)
    for rep in range(2):
        eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", int(sys.argv[6]) if len(sys.argv) > 6 else 8)
        got = taps(model, x, layers)
        for name in layers:
            d = (ref[name] - got[name]).abs()
            nz = int((d > 0).sum())
            print(f"rep {rep} {name:22s} max|diff| {float(d.max()):.3e}  differing values {nz}  first at {int(torch.nonzero(d > 0)[0]) if nz else -1}", flush=True)
    eng.lib.kp2d_set_option(eng.handle, b"wsm_min_items", 0)


if __name__ == "__main__":
    main()
