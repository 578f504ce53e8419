#!/usr/bin/env python3
"""Host-to-device rate of one 64-frame uint8 batch (14.7 MB): pageable (torch's staged synchronous copy) vs pinned
(hipMemcpyAsync, an SDMA engine) vs the preprocess kernel reading pinned memory in place.  Why the pinned front-end of
round 2 measured slower than the pageable one (profiles/r2_frontend_pcie.json)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    from nano_vs_slam_amd.pipeline import frames_to_input
    frames = np.random.default_rng(0).integers(0, 256, (64, 240, 320, 3), dtype=np.uint8)
    pageable = torch.from_numpy(frames)
    pinned = torch.from_numpy(frames).pin_memory()
    mb = frames.nbytes / 1e6
    res = {"batch_MB": round(mb, 2)}
    res["pageable_to_ms"] = round(timeit(lambda: pageable.to("cuda:0", non_blocking=True)), 3)
    res["pinned_to_async_ms"] = round(timeit(lambda: pinned.to("cuda:0", non_blocking=True)), 3)
    dev = pinned.to("cuda:0")
    res["preprocess_from_device_ms"] = round(timeit(lambda: frames_to_input(dev, "cuda:0")), 3)
    os.environ["KP2D_PINNED_ZERO_COPY"] = "1"
    res["preprocess_reading_pinned_in_place_ms"] = round(timeit(lambda: frames_to_input(pinned, "cuda:0")), 3)
    os.environ["KP2D_PINNED_ZERO_COPY"] = "0"
    res["pinned_copy_then_preprocess_ms"] = round(timeit(lambda: frames_to_input(pinned, "cuda:0")), 3)
    res["pageable_copy_then_preprocess_ms"] = round(timeit(lambda: frames_to_input(pageable, "cuda:0")), 3)
    for k in list(res):
        if k.endswith("_ms") and "preprocess_from_device" not in k:
            res[k.replace("_ms", "_GBps")] = round(mb / res[k], 2)
    res["HSA_ENABLE_SDMA"] = os.environ.get("HSA_ENABLE_SDMA", "(default)")
    print(json.dumps(res))


if __name__ == "__main__":
    main()
