"""Frame -> keypoints front-end, the host mirror of the reference's per-frame ``inference()``
(src/evaluation/visual_odometry.py:74-122) and ``KP2DtinyFrontend.run`` (src/visual_odometry/frontend.py:78-129).

Differences, all on purpose:
  * the image is converted and normalised on the device (``/255``, ``*2-1``), forward + post_processing +
    threshold/top-k selection all run as HIP kernels; only the selected rows are copied to the host
    (the reference copies every cell's score/coord/descriptor and selects with numpy);
  * whole batches are accepted (the reference's ``.view(3, -1)`` idiom assumes B == 1);
  * resizing (kornia bilinear in the reference) is not built yet: pass frames at the
    network resolution.
"""
from __future__ import annotations

import numpy as np
import torch

from .selectors import select_keypoints


def frames_to_input(frames, device) -> torch.Tensor:
    """uint8 [H,W,3] / [B,H,W,3] (numpy or torch) -> float32 [B,3,H,W] in [-1,1] on ``device``
    (kornia.image_to_tensor(image).float() / 255 then .sub(0.5).mul(2): visual_odometry.py:77,85)."""
    t = torch.as_tensor(np.asarray(frames) if not torch.is_tensor(frames) else frames)
    if t.dim() == 3:
        t = t.unsqueeze(0)
    if t.dtype != torch.uint8 or t.shape[-1] != 3:
        raise ValueError("expected uint8 frames of shape [H,W,3] or [B,H,W,3]")
    t = t.to(device, non_blocking=True).permute(0, 3, 1, 2).float()
    return t.div_(255.0).sub_(0.5).mul_(2.0).contiguous()


@torch.no_grad()
def inference(net, image, new_size=None, nn_thresh=0.7, top_k=4000, device="cuda"):
    """Returns (pts, feat, out) like the reference: for a single frame ``pts`` [n,2] and ``feat`` [n,C] numpy
    arrays; for a batch, lists of them.  ``out`` is the post-processed dict (device tensors)."""
    x = frames_to_input(image, device)
    _, _, H, W = x.shape
    if new_size is not None and tuple(new_size) != (H, W):
        raise NotImplementedError("on-device resize is not built yet: feed frames at the network resolution")
    out = net(x)
    out = net.post_processing(out, H, W)
    sel = select_keypoints(out, nn_thresh, top_k)
    pts = [p.cpu().numpy().copy() for p, _, _ in sel]
    feat = [d.cpu().numpy().copy() for _, d, _ in sel]
    single = (not torch.is_tensor(image) and np.asarray(image).ndim == 3) or (torch.is_tensor(image) and image.dim() == 3)
    if single:
        return pts[0], feat[0], out
    return pts, feat, out
