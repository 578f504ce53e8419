#!/usr/bin/env python3
"""BASELINE config 5: KP2DTiny-S keypoints + LightGlue on 480x640 image pairs (synthetic frames, seeded weights).

    python3 tools/bench_lightglue.py [--pairs 8] [--steps 20] [--kpts 1024] [--sweep 8,16,32]

Prints one JSON line per pairs-per-step value: image pairs/s for extractor (both images) + K3 top-k + matcher, the matcher
alone, and the `roofline` object of the step's dominant kernel family (HIP events around every launch of the extractor's
forward on the 2 x pairs frames, as bench.py does; the matcher's ~25 launches are timed as a whole beside it).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kpts", type=int, default=1024)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--sweep", default="", help="comma-separated pairs-per-step values (one line each) instead of --pairs")
    ap.add_argument("--filter-threshold", type=float, default=0.0,
                    help="LightGlue filter_threshold (seeded random weights give matching scores near 0: with the reference configs' "
                         "0.1 the filter and the compaction would see no match at all)")
    a = ap.parse_args()
    for pairs in ([int(v) for v in a.sweep.split(",")] if a.sweep else [a.pairs]):
        a.pairs = pairs
        run(a)


def run(a):
    from lightglue.lightglue import LightGlue
    from lightglue.lightglue_configs import get_light_glue_config
    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from nano_vs_slam_amd.pipeline import two_view_match
    from nano_vs_slam_amd.synthetic import seeded_linear_state_dict
    from nano_vs_slam_amd.synthetic import spread_state_dict
    dev = torch.device("cuda:0")
    net = tiny_factory("S", 28)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net = net.to(dev).eval()
    net.training = False
    conf = dict(get_light_glue_config("S"), filter_threshold=a.filter_threshold)
    lg = LightGlue(conf)
    shapes = {k: tuple(v.shape) for k, v in lg.state_dict().items()}
    lg.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_linear_state_dict(shapes).items()})
    lg = lg.to(dev).eval()
    g = torch.Generator(device=dev).manual_seed(0)
    im0 = torch.rand(a.pairs, 3, a.height, a.width, device=dev, generator=g)
    im1 = torch.rand(a.pairs, 3, a.height, a.width, device=dev, generator=g)

    def timed(fn):
        for _ in range(a.warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    with torch.no_grad():
        p0, p1, m = two_view_match(net, lg, im0, im1, a.kpts)
        data = {"keypoints0": p0["keypoints"], "keypoints1": p1["keypoints"], "descriptors0": p0["descriptors"],
                "descriptors1": p1["descriptors"], "view0": {"image_size": p0["image_size"]},
                "view1": {"image_size": p1["image_size"]}}
        t_all = timed(lambda: two_view_match(net, lg, im0, im1, a.kpts))
        t_lg = timed(lambda: lg(data))
        # roofline of the dominant kernel family: the extractor's forward on the 2 x pairs frames, one lane, HIP events per launch
        import bench
        H, W = a.height - a.height % 8, a.width - a.width % 8
        x = ((torch.cat([im0[:, :, :H, :W], im1[:, :, :H, :W]], 0) - 0.5) * 2.0).contiguous()
        agg = bench.kernel_profile(net, x, H, W, 3)
    dom = max(agg, key=lambda k: agg[k]["ms"])
    d = agg[dom]
    split = "f16x3" in dom
    peak = bench.PEAK_F16_MFMA_TFLOPS / 3.0 if split else bench.PEAK_F32_MFMA_TFLOPS
    achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12
    roof = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": None,
            "avg_launch_ms": round(d["ms"] / d["launches"], 4), "launches_per_step": d["launches"] // 3,
            "share_of_extractor_kernel_time": round(d["ms"] / sum(v["ms"] for v in agg.values()), 4),
            "extractor_kernel_ms_sum": round(sum(v["ms"] for v in agg.values()) / 3, 3),
            "note": "extractor forward of the step's 2 x pairs frames, single lane; the matcher (fp32, ~25 small launches) is matcher_ms_per_step"}
    print(json.dumps({
        "metric": "image pairs/sec KP2DTiny-S + LightGlue 480x640", "value": round(a.pairs / t_all * 1e3, 1),
        "unit": "pairs/s", "pairs_per_step": a.pairs, "keypoints": a.kpts, "ms_per_step": round(t_all, 3),
        "matcher_ms_per_step": round(t_lg, 3), "matcher_pairs_per_s": round(a.pairs / t_lg * 1e3, 1),
        "matched_fraction": round(float((m["matches0"] >= 0).float().mean()), 4), "data": "synthetic",
        "filter_threshold": a.filter_threshold, "roofline": roof,
        "dtype": "f16x3 extractor, fp32 matcher"}), flush=True)


if __name__ == "__main__":
    main()
