#!/usr/bin/env python3
"""PCIe-inclusive rate of the frame front-end: uint8 HWC frames in HOST memory -> device -> kp2d_preprocess -> forward
-> post_processing -> threshold/top-k selection -> selected rows back to the host (pipeline.inference).  Not the
headline metric (bench.py starts with inputs resident in HBM); DESIGN.md §5 quotes this number beside it."""
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
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--pinned", action="store_true")
    ap.add_argument("--eager", action="store_true", help="batch 1 only: call inference() per frame instead of FrameStream")
    ap.add_argument("--slots", type=int, default=7, help="batch 1 only: FrameStream slots (frames in flight)")
    ap.add_argument("--in-flight", type=int, default=1, help="batch > 1: batches in flight (pipeline.BatchStream.submit_frames); 1 = inference() per batch")
    ap.add_argument("--match", action="store_true", help="batch 1 only: match every frame against its predecessor on the device")
    ap.add_argument("--semantic", action="store_true", help="with --match: per-class matching (match_semantic)")
    ap.add_argument("--lightglue", action="store_true", help="batch 1 only: LightGlue (config S, seeded weights) as the loop's matcher instead of brute force")
    ap.add_argument("--top-k-matches", type=int, default=0, help="with --match / --lightglue: the loop's cap on the device (0: every match)")
    a = ap.parse_args()
    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from nano_vs_slam_amd.pipeline import BatchStream, FrameStream, inference
    from nano_vs_slam_amd.synthetic import spread_state_dict
    net = tiny_factory("S", 28)
    sd = spread_state_dict({k: tuple(v.shape) for k, v in net.state_dict().items()})
    net.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    net = net.to("cuda:0").eval()
    net.training = False
    frames = np.random.default_rng(0).integers(0, 256, (a.batch, 240, 320, 3), dtype=np.uint8)
    src = torch.from_numpy(frames).pin_memory() if a.pinned else frames
    if a.batch == 1 and not a.eager:
        # the VO loop: one frame per call, replayed HIP graph + overlapped upload (pipeline.FrameStream)
        if a.semantic:
            net.sample_segmentation = True
        matcher = None
        if a.lightglue:
            from lightglue.lightglue import LightGlue
            from lightglue.lightglue_configs import get_light_glue_config
            from nano_vs_slam_amd.synthetic import seeded_linear_state_dict
            matcher = LightGlue(dict(get_light_glue_config("S"), filter_threshold=0.0))
            shapes = {k: tuple(v.shape) for k, v in matcher.state_dict().items()}
            matcher.load_state_dict({k: torch.from_numpy(v) for k, v in seeded_linear_state_dict(shapes).items()})
            matcher = matcher.to("cuda:0").eval()
        matching = a.match or a.lightglue
        fs = FrameStream(net, (240, 320), None, 0.7, 1000, "cuda:0", slots=a.slots, match="lightglue" if a.lightglue else a.match,
                         semantic=a.semantic, matcher=matcher, top_k_matches=a.top_k_matches)
        # matching: consecutive frames of one scene shifted by whole cells (4 pixels) plus a little noise, as the tests' _vo_frames:
        # most keypoints have their correspondence in the next frame and survive the ratio test (hundreds of matches per frame;
        # round 4's frames were shifted by single pixels: 1-3 matches per frame)
        rng = np.random.default_rng(1)
        if matching:
            seq = []
            for i in range(a.steps):
                f = np.roll(frames[0], (4 * ((i // 2) % 6), 4 * (i % 12)), axis=(0, 1)).astype(np.int16) + rng.integers(-2, 3, frames[0].shape)
                seq.append(np.clip(f, 0, 255).astype(np.uint8))
        else:
            seq = [frames[0]] * a.steps
        for _ in fs.map(seq[:5]):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n, nm = 0, 0
        for r in fs.map(seq):
            n += 1
            nm += len(r[0])
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        how = "LightGlue (config S, padded sets: kp2d_lg_forward_counts) + get_matches_scores" if a.lightglue else \
              (("per class, " if a.semantic else "") + "BF k-NN(2) + ratio + one-to-one")
        cap = f", top_k_matches {a.top_k_matches}" if a.top_k_matches > 0 else ""
        what = (f"extract + select + match against the previous frame ({how}{cap}) on the device, D2H of the matched coordinate "
                "pairs only") if matching else "H2D of uint8 frames and D2H of keypoints"
        print(json.dumps({"metric": "frames/sec KP2DTiny-S 240x320 front-end incl. " + what,
                          "value": round(1e3 / ms, 1), "ms_per_step": round(ms, 3), "batch": 1,
                          "rows_per_frame": round(nm / n, 1),
                          "mode": f"FrameStream (HIP graph replay, {a.slots} pinned slots, "
                                  + ("ONE shared compute stream" if os.environ.get("KP2D_FS_SHARED_STREAM") == "1"
                                     else "a compute stream and workspace per slot") + ")"}))
        return
    if a.in_flight > 1:
        # whole batches, several in flight: the upload of batch n + 1 (the preprocess kernel reading pinned host memory) and the
        # download of batch n - 1's selected rows run behind batch n's kernels
        bs = BatchStream(net, slots=a.in_flight, top_k=1000, nn_thresh=0.7, device="cuda:0")
        for _ in bs.map_frames([src] * 4):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for pts, feat, out in bs.map_frames([src] * a.steps):
            n += len(pts)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        bs.close()
        print(json.dumps({"metric": "frames/sec KP2DTiny-S 240x320 front-end incl. H2D of uint8 frames and D2H of keypoints",
                          "value": round(a.batch / ms * 1e3, 1), "ms_per_step": round(ms, 3), "batch": a.batch,
                          "host_memory": "pinned" if a.pinned else "pageable",
                          "mode": f"BatchStream.submit_frames, {a.in_flight} batches in flight"}))
        return
    for _ in range(3):
        inference(net, src, None, 0.7, 1000, "cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        inference(net, src, None, 0.7, 1000, "cuda:0")
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    print(json.dumps({"metric": "frames/sec KP2DTiny-S 240x320 front-end incl. H2D of uint8 frames and D2H of keypoints",
                      "value": round(a.batch / ms * 1e3, 1), "ms_per_step": round(ms, 3), "batch": a.batch,
                      "host_memory": "pinned" if a.pinned else "pageable"}))


if __name__ == "__main__":
    main()
