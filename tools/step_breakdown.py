#!/usr/bin/env python3
"""Wall time of the bench step's parts in the bench setting (two lanes, 64 frames 240x320): forward only,
forward + post_processing, and the full step (+ top-k selection and gather).

    python3 tools/step_breakdown.py [--batch B]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
from nano_vs_slam_amd.selectors import select_and_gather
from nano_vs_slam_amd.synthetic import spread_state_dict
m = tiny_factory("S", 28)
sd = spread_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
m = m.to("cuda:0").eval(); m.training = False
_ap = argparse.ArgumentParser()
_ap.add_argument("--batch", type=int, default=64)
B = _ap.parse_args().batch
x = torch.rand(B, 3, 240, 320, device="cuda:0") * 2 - 1
def fwd(): return m(x)
def full():
    out = m.post_processing(m(x), 240, 320)
    return select_and_gather(out["score"], out["coord"], out["feat"], 1000, 0.7)[3:]
def fwd_post():
    return m.post_processing(m(x), 240, 320)
with torch.no_grad():
    for name, f in (("forward", fwd), ("forward+post", fwd_post), ("full step", full), ("forward", fwd), ("full step", full)):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): f()
        torch.cuda.synchronize(); print(name, round((time.perf_counter() - t0) / 40 * 1e3, 3), "ms")
