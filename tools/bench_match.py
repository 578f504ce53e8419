#!/usr/bin/env python3
"""Timing of the on-device matcher (kp2d_match_descriptors_ex) on the shapes the VO loop produces.

    python3 tools/bench_match.py [--reps 50]

One JSON line per case: microseconds per call (all kernels of the call, HIP events on the launch stream), pair-evaluations
per second and the VALU fraction (3 C lane-operations per (query, train) evaluation against 256 CUs x 128 lanes x clock).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nano_vs_slam_amd.matching import match_descriptors, match_pairs  # noqa: E402

CASES = [("1 pair 4000 x 4000 x 32", 1, 4000, 4000, 32, {}), ("1 pair 1000 x 1000 x 32", 1, 1000, 1000, 32, {}),
         ("64 pairs 1000 x 1000 x 32", 64, 1000, 1000, 32, {}), ("64 pairs 1000 x 1000 x 128", 64, 1000, 1000, 128, {}),
         ("64 pairs 1000 x 1000 x 32, 28 classes", 64, 1000, 1000, 32, {"classes": 28}),
         ("64 pairs 1000 x 1000 x 32, mutual", 64, 1000, 1000, 32, {"mutual": True}),
         ("1 pair 1000 x 1000 x 32, 28 classes", 1, 1000, 1000, 32, {"classes": 28})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    a = ap.parse_args()
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(3)
    for name, B, k0, k1, C, kw in CASES:
        d0 = torch.nn.functional.normalize(torch.randn(B, k0, C, device=dev, generator=g), dim=-1)
        d1 = torch.nn.functional.normalize(torch.randn(B, k1, C, device=dev, generator=g), dim=-1)
        n0 = torch.full((B,), k0, dtype=torch.int32, device=dev)
        n1 = torch.full((B,), k1, dtype=torch.int32, device=dev)
        extra = {}
        if "classes" in kw:
            extra = {"cls0": torch.randint(0, kw["classes"], (B, k0), device=dev, generator=g, dtype=torch.int32),
                     "cls1": torch.randint(0, kw["classes"], (B, k1), device=dev, generator=g, dtype=torch.int32)}
        if kw.get("mutual"):
            extra["mutual"] = True
        p0 = torch.rand(B, k0, 2, device=dev)
        p1 = torch.rand(B, k1, 2, device=dev)
        out = match_descriptors(d0, n0, d1, n1, 0.7, **extra)
        po = match_pairs(out, p0, p1)
        torch.cuda.synchronize()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        for _ in range(a.reps):
            match_descriptors(d0, n0, d1, n1, 0.7, out=out, **extra)
        e1.record()
        for _ in range(a.reps):
            match_pairs(out, p0, p1, out=po)
        e2.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / a.reps
        evals = B * k0 * k1 * (2 if kw.get("mutual") else 1)
        lane_ops = evals * 3 * C
        mfma = os.environ.get("KP2D_MATCH_MFMA", "1") != "0" and k1 >= 256
        rec = {"case": name, "form": "matrix cores + exact recheck" if mfma else "VALU", "us_per_call": round(us, 2),
               "pairs_kernel_us": round(e1.elapsed_time(e2) * 1e3 / a.reps, 2), "gevals_per_s": round(evals / us * 1e-3, 1),
               "matches_pair0": int(po["count"][0])}
        if mfma:     # 3 C / 16 MFMAs of 32 x 32 x 16 per 1024 evaluations against the dense fp16 peak
            rec["mfma_frac_of_2516_TFLOPs"] = round(evals * 3 * C * 2 / (us * 1e-6) / 2516.6e12, 3)
        else:        # 3 C lane-operations per evaluation against 256 CUs x 128 lanes x 2.4 GHz
            rec["valu_frac_at_2.4GHz"] = round(lane_ops / (us * 1e-6) / (256 * 128 * 2.4e9), 3)
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
