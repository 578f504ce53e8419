#!/usr/bin/env python3
"""demo.py equivalent (reference demo.py:1-28): tiny_factory("S_A", 28, v3=True) at 240x320, frame by frame.

The reference's demo needs ./demo_data/V3_S_A_p_best.ckpt, an mp4 and OpenCV, none of which are
distributed.  This entry point keeps its model-facing sequence (factory -> load_state_dict(strict=False)
-> to(device) -> eval() -> training=False -> per-frame inference) and runs it on a synthetic frame
sequence when no video/weights are present; tracking / pose estimation (cv2) are out of scope.
"""
import argparse
import os
import time

import numpy as np
import torch

from src.kp2dtiny.models.kp2dtiny import tiny_factory
from nano_vs_slam_amd.pipeline import FrameStream, inference


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--weights", default="./demo_data/V3_S_A_p_best.ckpt")
    ap.add_argument("--frames", type=int, default=100)
    ap.add_argument("--device", default="cuda" if torch.cuda.is_available() else "cpu")
    ap.add_argument("--no-match", action="store_true", help="extract only (keypoints and descriptors to the host), as in round 3")
    a = ap.parse_args()
    model = tiny_factory("S_A", 28, v3=True).cpu()
    if os.path.exists(a.weights):
        model.load_state_dict(torch.load(a.weights, map_location=torch.device("cpu"), weights_only=True)["state_dict"], strict=False)
    else:
        from nano_vs_slam_amd.synthetic import spread_state_dict   # seeded stand-in weights (test infrastructure)
        sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
        model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=False)
        print(f"{a.weights} not found: using seeded synthetic weights")
    model = model.to(a.device)
    model.eval()
    model.training = False

    new_size = (240, 320)
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (new_size[0], new_size[1], 3), dtype=np.uint8)
    n_kp = []
    pts, feat, out = inference(model, frame, new_size, device=a.device)      # warm-up
    torch.cuda.synchronize()
    # the reference calls inference() once per video frame (evaluation/visual_odometry.py:409-495); FrameStream is that
    # same step replayed as a HIP graph, with the next frame's upload overlapping the current frame's kernels
    def video(n, f):
        for _ in range(n):
            f = np.roll(f, 3, axis=1)                                         # a panning "video"
            yield f
    if a.no_match:
        fs = FrameStream(model, frame.shape[:2], new_size, device=a.device)
        t0 = time.perf_counter()
        for pts, feat, out in fs.map(video(a.frames, frame)):
            n_kp.append(len(pts))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{a.frames} frames, {a.frames / dt:.1f} frames/s (single-frame latency path), "
              f"{np.mean(n_kp):.0f} keypoints/frame, descriptor dim {feat.shape[1] if len(feat) else 0}, "
              f"seg classes seen {int(out['seg'].unique().numel())}")
        return
    # the VO loop's next step on the device too (visual_odometry.py:193-284: matcher.match(prev_descriptors, feat_cur)):
    # every frame is matched against its predecessor inside the replayed graphs; the host receives the matched coordinate
    # pairs (what estimatePose takes), never the descriptors
    fs = FrameStream(model, frame.shape[:2], new_size, device=a.device, match=True)
    n_m = []
    t0 = time.perf_counter()
    for kps0, kps1, dist, out in fs.map(video(a.frames, frame)):
        n_m.append(len(kps0))
        n_kp.append(int(out["rows"]["cnt"][0]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    shift = np.median(kps1[:, 0] - kps0[:, 0]) if len(kps0) else float("nan")
    print(f"{a.frames} frames, {a.frames / dt:.1f} frames/s (extract + select + match, single-frame latency path), "
          f"{np.mean(n_kp):.0f} keypoints/frame, {np.mean(n_m[1:]) if len(n_m) > 1 else 0:.0f} matches/frame, "
          f"median x displacement of the last frame's matches {shift:.2f} px (the synthetic video pans by 3), "
          f"seg classes seen {int(out['seg'].unique().numel())}")


if __name__ == "__main__":
    main()
