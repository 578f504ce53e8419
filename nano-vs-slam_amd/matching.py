"""Descriptor matching on the device — host mirror of the reference's brute-force matchers:

* ``BfFeatureMatcher.match`` (src/visual_odometry/feature_matcher.py:89-98: cv2.BFMatcher(NORM_L2).knnMatch(des1,
  des2, k=2), then ``goodMatchesOneToOne`` :179-209: ratio test 0.7 and one query per train index) — the VO loop's
  matcher (src/visual_odometry/visual_odometry.py:270-284);
* ``VisualOdometry.match_semantic`` (visual_odometry.py:347-380: the same per semantic class) — ``cls0 / cls1``;
* ``cv2.BFMatcher(NORM_L2, crossCheck=False).match`` (src/evaluation/descriptor.py:132-134) — ``nn_idx``;
* ``cv2.BFMatcher(NORM_L2, crossCheck=True).match`` (descriptor.py:221-222) — ``mutual=True``.

The reference copies every frame's keypoints to the host and matches with OpenCV; here whole batches of frame pairs are
matched by HIP kernels (kp2d_match_descriptors_ex) straight from the [B,k,C] tensors that
``selectors.gather_keypoints`` produces, so the VO loop's per-frame device->host copy of every descriptor disappears
(SURVEY.md §8f rank 1).  Pose estimation (cv2.findEssentialMat) stays out of scope.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib

MATCH_MUTUAL = 1


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def match_descriptors(desc0: torch.Tensor, cnt0: torch.Tensor, desc1: torch.Tensor, cnt1: torch.Tensor,
                      ratio: float = 0.7, cls0: torch.Tensor | None = None, cls1: torch.Tensor | None = None,
                      mutual: bool = False, out: dict | None = None):
    """desc0 [B,k0,C] (query), cnt0 [B] int32, desc1 [B,k1,C] (train), cnt1 [B] -> dict of device tensors:
    nn_idx/nn_dist/nn_dist2 [B,k0] (k=2 neighbours of every query), match_q/match_d [B,k1] (query kept for each train
    row, -1 = unmatched: after ratio test + one-to-one, or — ``mutual`` — the mutual nearest neighbour).
    ``cls0`` [B,k0] / ``cls1`` [B,k1] int32 class ids restrict every query to train rows of its own class.
    ``out``: a dict from an earlier call with the same shapes, to reuse its tensors (static buffers for graph capture)."""
    if desc0.device.type != "cuda":
        raise RuntimeError("match_descriptors runs on the HIP device only")
    if (cls0 is None) != (cls1 is None):
        raise ValueError("class ids must be given for both sides or neither")
    lib = _lib.load()
    desc0, desc1 = desc0.contiguous().float(), desc1.contiguous().float()
    cnt0, cnt1 = cnt0.contiguous().to(torch.int32), cnt1.contiguous().to(torch.int32)
    B, k0, Cd = desc0.shape
    k1 = desc1.shape[1]
    dev = desc0.device
    if cls0 is not None:
        cls0, cls1 = cls0.contiguous().to(torch.int32), cls1.contiguous().to(torch.int32)
        if tuple(cls0.shape) != (B, k0) or tuple(cls1.shape) != (B, k1):
            raise ValueError("class ids must be [B,k0] and [B,k1]")
    if out is None:
        nbytes = int(lib.kp2d_match_scratch_bytes(B, k0, k1))
        out = {"nn_idx": torch.empty(B, k0, dtype=torch.int32, device=dev), "nn_dist": torch.empty(B, k0, device=dev),
               "nn_dist2": torch.empty(B, k0, device=dev), "match_q": torch.empty(B, k1, dtype=torch.int32, device=dev),
               "match_d": torch.empty(B, k1, device=dev),
               "_scratch": torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)}
    scratch = out["_scratch"]
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.kp2d_match_descriptors_ex(_ptr(desc0), _ptr(cnt0), _ptr(desc1), _ptr(cnt1), B, k0, k1, Cd, float(ratio),
                                             _ptr(cls0), _ptr(cls1), MATCH_MUTUAL if mutual else 0,
                                             _ptr(out["nn_idx"]), _ptr(out["nn_dist"]), _ptr(out["nn_dist2"]),
                                             _ptr(out["match_q"]), _ptr(out["match_d"]), _ptr(scratch),
                                             scratch.numel() * 8, C.c_void_p(stream)))
    return out


def match_pairs(match: dict, pts0: torch.Tensor | None = None, pts1: torch.Tensor | None = None, out: dict | None = None):
    """Compact the matched rows of every pair (train order): ``count`` [B], ``idx`` [B,k1,2] (query row, train row),
    ``dist`` [B,k1] and — with ``pts0`` [B,k0,2] / ``pts1`` [B,k1,2] — ``pairs`` [B,k1,4] = (x0, y0, x1, y1)."""
    lib = _lib.load()
    mq, md = match["match_q"], match["match_d"]
    B, k1 = mq.shape
    k0 = match["nn_idx"].shape[1]
    dev = mq.device
    if out is None:
        out = {"count": torch.empty(B, dtype=torch.int32, device=dev), "idx": torch.empty(B, k1, 2, dtype=torch.int32, device=dev),
               "dist": torch.empty(B, k1, device=dev)}
        if pts0 is not None:
            out["pairs"] = torch.empty(B, k1, 4, device=dev)
    if pts0 is not None:
        pts0, pts1 = pts0.contiguous().float(), pts1.contiguous().float()
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.kp2d_match_pairs(_ptr(mq), _ptr(md), _ptr(pts0), _ptr(pts1), B, k0, k1, _ptr(out.get("pairs")), _ptr(out["idx"]),
                                    _ptr(out["dist"]), _ptr(out["count"]), C.c_void_p(stream)))
    return out


TOPK_BF, TOPK_LG = 0, 1


def match_topk_pairs(k: int, pts0: torch.Tensor | None, pts1: torch.Tensor | None, match: dict | None = None,
                     matches0: torch.Tensor | None = None, scores0: torch.Tensor | None = None, out: dict | None = None):
    """The VO loop's ``top_k_matches`` cap on the device, fused with the compaction (kp2d_match_topk_pairs; replaces
    visual_odometry.py:272-283 for the brute-force branch — the k smallest distances — and :26-32 + :260-266 for the
    LightGlue branch — ``get_matches_scores`` then ``scores.topk(k)``).  Either ``match`` (the dict of
    ``match_descriptors``) or ``matches0`` [B,M] int64 + ``scores0`` [B,M] (``matches0`` / ``matching_scores0`` of
    ``LightGlue.forward``).  ``k <= 0``: every match.  Returns ``count`` [B], ``idx`` [B,kcap,2] (row in set 0, row in set
    1), ``val`` [B,kcap] (distance / score) and — with the keypoints — ``pairs`` [B,kcap,4]; best first.
    ``out``: the dict of an earlier call with the same shapes (static buffers for graph capture)."""
    lib = _lib.load()
    if (match is None) == (matches0 is None):
        raise ValueError("pass either match= (brute force) or matches0= / scores0= (LightGlue)")
    if match is not None:
        mode, src_i, src_l, val = TOPK_BF, match["match_q"], None, match["match_d"]
        B, max1 = src_i.shape
        max0 = match["nn_idx"].shape[1]
        n = max1
    else:
        mode, src_i, src_l, val = TOPK_LG, None, matches0.contiguous(), scores0.contiguous().float()
        if src_l.dtype != torch.int64:
            raise TypeError("matches0 must be int64 (LightGlue's output)")
        B, max0 = src_l.shape
        if pts1 is None:
            raise ValueError("the LightGlue form needs the keypoints of set 1 for its row count")
        max1 = pts1.shape[1]
        n = max0
    dev = val.device
    if dev.type != "cuda":
        raise RuntimeError("match_topk_pairs runs on the HIP device only")
    kcap = n if (k <= 0 or k > n) else int(k)
    if out is None:
        nbytes = int(lib.kp2d_match_topk_scratch_bytes(B, max0, max1))
        out = {"count": torch.empty(B, dtype=torch.int32, device=dev), "idx": torch.empty(B, kcap, 2, dtype=torch.int32, device=dev),
               "val": torch.empty(B, kcap, device=dev), "_scratch": torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)}
        if pts0 is not None:
            out["pairs"] = torch.empty(B, kcap, 4, device=dev)
    if pts0 is not None:
        pts0, pts1 = pts0.contiguous().float(), pts1.contiguous().float()
    stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(lib.kp2d_match_topk_pairs(mode, _ptr(src_i), _ptr(src_l), _ptr(val), _ptr(pts0), _ptr(pts1), B, max0, max1, int(k),
                                         _ptr(out.get("pairs")), _ptr(out["idx"]), _ptr(out["val"]), _ptr(out["count"]),
                                         _ptr(out["_scratch"]), out["_scratch"].numel() * 8, C.c_void_p(stream)))
    return out


def _single(des1, des2, **kw):
    d1 = torch.as_tensor(des1, dtype=torch.float32, device="cuda").unsqueeze(0)
    d2 = torch.as_tensor(des2, dtype=torch.float32, device="cuda").unsqueeze(0)
    c1 = torch.tensor([d1.shape[1]], dtype=torch.int32, device="cuda")
    c2 = torch.tensor([d2.shape[1]], dtype=torch.int32, device="cuda")
    return match_descriptors(d1, c1, d2, c2, **kw)


def bf_match(des1, des2, ratio_test: float = 0.7):
    """Single pair, reference signature: match(des1 = query, des2 = train) -> (idx1, idx2, score) lists
    (feature_matcher.py:89-98).  Order: ascending train index (the reference's order is insertion order;
    callers index both lists jointly, so only the pairing matters)."""
    if len(des1) == 0 or len(des2) == 0:
        return [], [], []
    r = _single(des1, des2, ratio=ratio_test)
    mq, md = r["match_q"][0], r["match_d"][0]
    idx2 = torch.nonzero(mq >= 0).squeeze(1)
    return mq[idx2].tolist(), idx2.tolist(), md[idx2].tolist()


def bf_match_semantic(des1, cls1, des2, cls2, ratio_test: float = 0.7):
    """``match_semantic`` (visual_odometry.py:347-380) for one pair: des1 / cls1 = previous frame's descriptors and
    per-keypoint class ids (query side), des2 / cls2 = current frame (train side) -> (idx1, idx2, score), indices into
    the FULL arrays (the reference concatenates per-class index lists; callers only use the pairing)."""
    if len(des1) == 0 or len(des2) == 0:
        return [], [], []
    k1 = torch.as_tensor(cls1, dtype=torch.int32, device="cuda").reshape(1, -1)
    k2 = torch.as_tensor(cls2, dtype=torch.int32, device="cuda").reshape(1, -1)
    r = _single(des1, des2, ratio=ratio_test, cls0=k1, cls1=k2)
    mq, md = r["match_q"][0], r["match_d"][0]
    idx2 = torch.nonzero(mq >= 0).squeeze(1)
    return mq[idx2].tolist(), idx2.tolist(), md[idx2].tolist()


def bf_match_nn(des1, des2):
    """cv2.BFMatcher(NORM_L2, crossCheck=False).match(des1, des2) (src/evaluation/descriptor.py:132-134): every query's
    nearest train row -> (queryIdx, trainIdx, distance) lists in query order."""
    if len(des1) == 0 or len(des2) == 0:
        return [], [], []
    r = _single(des1, des2)
    n = len(des1)
    return list(range(n)), r["nn_idx"][0, :n].tolist(), r["nn_dist"][0, :n].tolist()


def bf_match_crosscheck(des1, des2):
    """cv2.BFMatcher(NORM_L2, crossCheck=True).match(des1, des2) (descriptor.py:221-222): mutual nearest neighbours ->
    (queryIdx, trainIdx, distance) lists, ascending train index."""
    if len(des1) == 0 or len(des2) == 0:
        return [], [], []
    r = _single(des1, des2, mutual=True)
    mq, md = r["match_q"][0], r["match_d"][0]
    idx2 = torch.nonzero(mq >= 0).squeeze(1)
    return mq[idx2].tolist(), idx2.tolist(), md[idx2].tolist()
