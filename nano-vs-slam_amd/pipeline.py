"""Frame -> keypoints front-end, the host mirror of the reference's per-frame ``inference()``
(src/evaluation/visual_odometry.py:74-122) and ``KP2DtinyFrontend.run`` (src/visual_odometry/frontend.py:78-129).

Differences, all on purpose:
  * the image is converted and normalised on the device (``/255``, ``*2-1``), forward + post_processing +
    threshold/top-k selection all run as HIP kernels; only the selected rows are copied to the host
    (the reference copies every cell's score/coord/descriptor and selects with numpy);
  * whole batches are accepted (the reference's ``.view(3, -1)`` idiom assumes B == 1);
  * the uint8 frame is resized (bilinear, align_corners=False, as kornia's resize) and normalised by one HIP kernel
    (kp2d_preprocess); keypoints are scaled back to the original frame as the reference does.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from .selectors import select_keypoints, select_keypoints_host


def frames_to_input(frames, device, size=None) -> torch.Tensor:
    """uint8 [Hs,Ws,3] / [B,Hs,Ws,3] (numpy or torch) -> float32 [B,3,H,W] in [-1,1] on ``device``;
    ``size`` = (H, W) resizes (visual_odometry.py:77-87: /255, kornia resize, .sub(0.5).mul(2))."""
    t = torch.as_tensor(np.asarray(frames) if not torch.is_tensor(frames) else frames)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or t.shape[-1] != 3:
        raise ValueError("expected uint8 frames of shape [H,W,3] or [B,H,W,3]")
    t = t.to(device, non_blocking=True).contiguous()
    if t.device.type != "cuda":
        raise RuntimeError("the frame front-end runs on the HIP device only")
    B, Hs, Ws, _ = t.shape
    H, W = (Hs, Ws) if size is None else (int(size[0]), int(size[1]))
    x = torch.empty(B, 3, H, W, device=t.device)
    stream = torch.cuda.current_stream(t.device).cuda_stream
    _lib.check(_lib.load().kp2d_preprocess(C.c_void_p(t.data_ptr()), B, Hs, Ws, C.c_void_p(x.data_ptr()), H, W,
                                           C.c_void_p(stream)))
    return x


@torch.no_grad()
def inference(net, image, new_size=None, nn_thresh=0.7, top_k=4000, device="cuda"):
    """Returns (pts, feat, out) like the reference: for a single frame ``pts`` [n,2] and ``feat`` [n,C] numpy
    arrays; for a batch, lists of them.  ``out`` is the post-processed dict (device tensors)."""
    src_hw = tuple(image.shape[-3:-1])
    x = frames_to_input(image, device, new_size)
    _, _, H, W = x.shape
    out = net(x)
    out = net.post_processing(out, H, W)
    scale = None
    if new_size is not None and (H, W) != src_hw:     # pts / scale: visual_odometry.py:81-83, 119-121
        scale = (W / float(src_hw[1]), H / float(src_hw[0]))
    sel = select_keypoints_host(out, nn_thresh, top_k, scale)
    pts = [p for p, _ in sel]
    feat = [d for _, d in sel]
    single = (not torch.is_tensor(image) and np.asarray(image).ndim == 3) or (torch.is_tensor(image) and image.dim() == 3)
    if single:
        return pts[0], feat[0], out
    return pts, feat, out


@torch.no_grad()
def two_view_match(net, matcher, image0: torch.Tensor, image1: torch.Tensor, max_num_keypoints: int = 1024):
    """Extractor + LightGlue on batches of image pairs, everything on the device — the model-facing sequence of
    gluefactory's ``two_view_pipeline`` with the ``kp2dtiny`` extractor and the ``lightglue`` matcher
    (gluefactory/models/extractors/kp2dtiny.py:24-58, gluefactory/configs/kp2dtiny_S+lightglue_homography.yaml).

    image0 / image1: float32 [B,3,H,W] in [0,1] (the extractor applies ``.sub(0.5).mul(2)`` itself, :25), H and W are
    cropped to multiples of 8 (:30-33).  Returns (pred0, pred1, matches) with the extractor's and the matcher's dicts.
    """
    from .selectors import extract_topk
    if image0.shape != image1.shape:
        raise ValueError("two_view_match expects both views at the same size")
    B = image0.shape[0]
    H, W = image0.shape[-2] - image0.shape[-2] % 8, image0.shape[-1] - image0.shape[-1] % 8
    # both views go through the extractor as ONE batch (frames are independent): twice the work per launch
    x = (torch.cat([image0[:, :, :H, :W], image1[:, :, :H, :W]], 0) - 0.5) * 2.0
    out = net.post_processing(net(x.contiguous()), H, W)
    both = extract_topk(out, max_num_keypoints)
    size = torch.tensor([float(W), float(H)], device=x.device).expand(B, 2)
    preds = []
    for v in range(2):
        p = {k: t[v * B:(v + 1) * B] for k, t in both.items()}
        p["image_size"] = size
        preds.append(p)
    data = {"keypoints0": preds[0]["keypoints"], "keypoints1": preds[1]["keypoints"],
            "descriptors0": preds[0]["descriptors"], "descriptors1": preds[1]["descriptors"],
            "view0": {"image_size": preds[0]["image_size"]}, "view1": {"image_size": preds[1]["image_size"]}}
    return preds[0], preds[1], matcher(data)
