"""Descriptor matching on the device — host mirror of ``BfFeatureMatcher.match``
(src/visual_odometry/feature_matcher.py:89-98: cv2.BFMatcher(NORM_L2).knnMatch(des1, des2, k=2), then
``goodMatchesOneToOne`` :179-209: ratio test 0.7 and one query per train index).

The reference copies every frame's keypoints to the host and matches with OpenCV; here whole batches of
frame pairs are matched by HIP kernels (kp2d_match_descriptors) straight from the [B,k,C] tensors that
``selectors.gather_keypoints`` produces, so the VO loop's per-frame device->host copy disappears
(SURVEY.md §8f rank 1).  Pose estimation (cv2.findEssentialMat) stays out of scope.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def match_descriptors(desc0: torch.Tensor, cnt0: torch.Tensor, desc1: torch.Tensor, cnt1: torch.Tensor,
                      ratio: float = 0.7):
    """desc0 [B,k0,C] (query), cnt0 [B] int32, desc1 [B,k1,C] (train), cnt1 [B] -> dict of device tensors:
    nn_idx/nn_dist/nn_dist2 [B,k0] (k=2 neighbours of every query), match_q/match_d [B,k1] (query kept for
    each train row after ratio test + one-to-one, -1 = unmatched)."""
    if desc0.device.type != "cuda":
        raise RuntimeError("match_descriptors runs on the HIP device only")
    lib = _lib.load()
    desc0, desc1 = desc0.contiguous().float(), desc1.contiguous().float()
    cnt0, cnt1 = cnt0.contiguous().to(torch.int32), cnt1.contiguous().to(torch.int32)
    B, k0, Cd = desc0.shape
    k1 = desc1.shape[1]
    dev = desc0.device
    nn_idx = torch.empty(B, k0, dtype=torch.int32, device=dev)
    nn_d = torch.empty(B, k0, device=dev)
    nn_d2 = torch.empty(B, k0, device=dev)
    mq = torch.empty(B, k1, dtype=torch.int32, device=dev)
    md = torch.empty(B, k1, device=dev)
    scratch = torch.empty(B * k1, dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.kp2d_match_descriptors(_ptr(desc0), _ptr(cnt0), _ptr(desc1), _ptr(cnt1), B, k0, k1, Cd, float(ratio),
                                          _ptr(nn_idx), _ptr(nn_d), _ptr(nn_d2), _ptr(mq), _ptr(md), _ptr(scratch),
                                          C.c_void_p(stream)))
    return {"nn_idx": nn_idx, "nn_dist": nn_d, "nn_dist2": nn_d2, "match_q": mq, "match_d": md}


def bf_match(des1, des2, ratio_test: float = 0.7):
    """Single pair, reference signature: match(des1 = query, des2 = train) -> (idx1, idx2, score) lists
    (feature_matcher.py:89-98).  Order: ascending train index (the reference's order is insertion order;
    callers index both lists jointly, so only the pairing matters)."""
    d1 = torch.as_tensor(des1, dtype=torch.float32, device="cuda").unsqueeze(0)
    d2 = torch.as_tensor(des2, dtype=torch.float32, device="cuda").unsqueeze(0)
    c1 = torch.tensor([d1.shape[1]], dtype=torch.int32, device="cuda")
    c2 = torch.tensor([d2.shape[1]], dtype=torch.int32, device="cuda")
    r = match_descriptors(d1, c1, d2, c2, ratio_test)
    mq, md = r["match_q"][0], r["match_d"][0]
    idx2 = torch.nonzero(mq >= 0).squeeze(1)
    return mq[idx2].tolist(), idx2.tolist(), md[idx2].tolist()
