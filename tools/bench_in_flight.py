#!/usr/bin/env python3
"""Batches in flight: throughput of forward + post_processing + selection at 64 x 240 x 320 with 1, 2, 3, ... independent
(model replica, HIP stream) slots fed alternately — the experiment behind pipeline.BatchStream (which shares ONE engine
handle and gives each slot its own workspace instead of a replica).

    SLOTS=1,2,3,4 KP2D_LANES=1 python3 tools/bench_in_flight.py       # one engine lane per forward
    SLOTS=1,2 python3 tools/bench_in_flight.py                        # the engine's default two lanes per forward
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
from nano_vs_slam_amd.selectors import select_and_gather
from nano_vs_slam_amd.synthetic import spread_state_dict
dev = torch.device("cuda:0")
def mk():
    m = tiny_factory("S", 28)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in m.state_dict().items()})
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m = m.to(dev).eval(); m.training = False
    return m
B, H, W = 64, 240, 320
x = torch.rand(B, 3, H, W, device=dev) * 2 - 1
def step(m):
    out = m(x); out = m.post_processing(out, H, W)
    return select_and_gather(out["score"], out["coord"], out["feat"], 1000, 0.7)
for nslots in [int(v) for v in os.environ.get('SLOTS','1,2,3,1,2').split(',')]:
    ms_ = [mk() for _ in range(nslots)]
    ss = [torch.cuda.Stream(dev) for _ in range(nslots)]
    keep = [None] * nslots
    with torch.no_grad():
        for k in range(6):
            with torch.cuda.stream(ss[k % nslots]): keep[k % nslots] = step(ms_[k % nslots])
        torch.cuda.synchronize()
        K = 60
        t0 = time.perf_counter()
        for k in range(K):
            with torch.cuda.stream(ss[k % nslots]): keep[k % nslots] = step(ms_[k % nslots])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"slots {nslots}: {K * B / dt:9.1f} frames/s  {dt / K * 1e3:.3f} ms/step", flush=True)
    del ms_
