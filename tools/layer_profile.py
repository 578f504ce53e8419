#!/usr/bin/env python3
"""Per-layer HIP-event profile of one forward (uses the engine's kp2d_profile_* hooks).

    python3 tools/layer_profile.py [--config S] [--v3] [--batch 64] [--height 240] [--width 320] [--chunk 0]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="S")
    ap.add_argument("--v3", action="store_true")
    ap.add_argument("--n-classes", type=int, default=28)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()
    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from nano_vs_slam_amd.synthetic import spread_state_dict
    model = tiny_factory(a.config, a.n_classes, v3=a.v3)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to("cuda:0").eval()
    model.set_precision(a.precision)
    model.training = False
    x = torch.rand(a.batch, 3, a.height, a.width, device="cuda:0") * 2 - 1
    with torch.no_grad():
        model(x)
        eng = model._engine
        lib = eng.lib
        if a.chunk:
            lib.kp2d_set_chunk_frames(eng.handle, a.chunk)
            eng._ws = None
        for _ in range(2):
            model(x)
        torch.cuda.synchronize()
        import time
        t0 = time.perf_counter()
        for _ in range(a.reps):
            model(x)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / a.reps * 1e3
        lib.kp2d_set_profiling(eng.handle, 1)
        agg = {}
        order = []
        for _ in range(a.reps):
            model(x)
            n = lib.kp2d_profile_count(eng.handle)
            layer, kern = C.c_char_p(), C.c_char_p()
            ms, fl, by = C.c_float(), C.c_double(), C.c_double()
            for i in range(n):
                lib.kp2d_profile_get(eng.handle, i, C.byref(layer), C.byref(kern), C.byref(ms), C.byref(fl), C.byref(by))
                key = (layer.value.decode(), kern.value.decode())
                if key not in agg:
                    agg[key] = [0.0, 0.0, 0.0, 0]
                    order.append(key)
                r = agg[key]
                r[0] += ms.value
                r[1] += fl.value
                r[2] += by.value
                r[3] += 1
    tot = sum(r[0] for r in agg.values()) / a.reps
    print(f"forward wall {wall:.3f} ms/batch ({a.batch / wall * 1e3:.0f} frames/s); sum of kernel events {tot:.3f} ms")
    print(f"{'layer':38s} {'kernel':20s} {'ms':>8s} {'%':>6s} {'TFLOP/s':>8s} {'GB/s':>8s} {'launches':>8s}")
    for key in order:
        r = agg[key]
        ms = r[0] / a.reps
        print(f"{key[0]:38s} {key[1]:20s} {ms:8.3f} {100 * ms / tot:6.1f} {r[1] / r[0] / 1e9:8.1f} {r[2] / r[0] / 1e6:8.0f} {r[3] // a.reps:8d}")


if __name__ == "__main__":
    main()
