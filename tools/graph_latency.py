#!/usr/bin/env python3
"""Single-frame latency of the VO front-end path, eager vs replayed as a HIP graph (torch.cuda.CUDAGraph).

    python3 tools/graph_latency.py [--config S_A --v3] [--height 240 --width 320] [--batch 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="S_A")
    ap.add_argument("--v3", action="store_true")
    ap.add_argument("--n-classes", type=int, default=28)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--height", type=int, default=240)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from nano_vs_slam_amd.selectors import select_and_gather
    from nano_vs_slam_amd.synthetic import spread_state_dict
    model = tiny_factory(a.config, a.n_classes, v3=a.v3)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    model = model.to("cuda:0").eval()
    model.training = False
    x = torch.rand(a.batch, 3, a.height, a.width, device="cuda:0") * 2 - 1

    def step():
        out = model.post_processing(model(x), a.height, a.width)
        idx, val, cnt, pts, desc = select_and_gather(out["score"], out["coord"], out["feat"], 1000, 0.7)
        return out, pts, desc, cnt

    with torch.no_grad():
        for _ in range(5):
            ref = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            step()
        torch.cuda.synchronize()
        eager = (time.perf_counter() - t0) / a.iters * 1e3
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                step()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g):
            got = step()
        g.replay()
        torch.cuda.synchronize()
        same = torch.equal(got[0]["score"], ref[0]["score"]) and torch.equal(got[1], ref[1])
        t0 = time.perf_counter()
        for _ in range(a.iters):
            g.replay()
        torch.cuda.synchronize()
        graph = (time.perf_counter() - t0) / a.iters * 1e3
    print(f"{a.config}{' V3' if a.v3 else ''} {a.height}x{a.width} B={a.batch}: eager {eager:.3f} ms/step ({a.batch / eager * 1e3:.0f} frames/s), "
          f"graph replay {graph:.3f} ms/step ({a.batch / graph * 1e3:.0f} frames/s), outputs identical: {same}")


if __name__ == "__main__":
    main()
