"""Keypoint selection on the device — host mirror of the reference's three selectors:

  K1  VO selector      src/evaluation/visual_odometry.py:93-122, src/visual_odometry/frontend.py:94-129
  K2  eval selector    src/evaluation/keypoints.py:113-128 + src/evaluation/descriptor.py:12-36
  K3  gluefactory      gluefactory/models/extractors/kp2dtiny.py:38-52  (batched torch.topk)

All three reduce to "score > thr, then the k best cells"; the reference does K1/K2 on the CPU after
copying every cell's score/coord/descriptor to the host (and assumes B == 1).  Here the threshold,
the exact top-k and the gather run as HIP kernels (kp2d_select_topk / kp2d_gather_keypoints) for the
whole batch; only the selected rows ever leave the device.  Order is (score desc, flat index asc) —
torch.topk's order; K1/K2 callers only depend on the selected SET.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()


def select_topk(score: torch.Tensor, k: int, thr: float = float("-inf")):
    """score [B,1,Hc,Wc] or [B,n] (device) -> (idx int32 [B,k] padded with -1, val [B,k], count int32 [B])."""
    if score.device.type != "cuda":
        raise RuntimeError("select_topk runs on the HIP device only")
    lib = _lib.load()
    B = score.shape[0]
    s = score.reshape(B, -1).contiguous().float()
    n = s.shape[1]
    if k < 1:
        raise ValueError("k must be >= 1 (use k = number of cells for 'every cell above the threshold')")
    k = int(min(k, n))
    idx = torch.empty(B, k, dtype=torch.int32, device=s.device)
    val = torch.empty(B, k, dtype=torch.float32, device=s.device)
    cnt = torch.empty(B, dtype=torch.int32, device=s.device)
    stream = torch.cuda.current_stream(s.device).cuda_stream
    _lib.check(lib.kp2d_select_topk(_ptr(s), B, n, k, float(thr), _ptr(idx), _ptr(val), _ptr(cnt), C.c_void_p(stream)))
    return idx, val, cnt


def gather_keypoints(coord: torch.Tensor, desc: torch.Tensor, idx: torch.Tensor):
    """coord [B,2,Hc,Wc], desc [B,C,Hc,Wc], idx [B,k] -> pts [B,k,2] (x,y), descriptors [B,k,C]."""
    lib = _lib.load()
    B, Cd = desc.shape[0], desc.shape[1]
    n = desc.shape[2] * desc.shape[3]
    k = idx.shape[1]
    coord, desc, idx = coord.contiguous(), desc.contiguous(), idx.contiguous()
    pts = torch.empty(B, k, 2, device=desc.device)
    dsel = torch.empty(B, k, Cd, device=desc.device)
    stream = torch.cuda.current_stream(desc.device).cuda_stream
    _lib.check(lib.kp2d_gather_keypoints(_ptr(coord), _ptr(desc), _ptr(idx), B, Cd, n, k, _ptr(pts), _ptr(dsel),
                                         C.c_void_p(stream)))
    return pts, dsel


def select_and_gather(score: torch.Tensor, coord: torch.Tensor, desc: torch.Tensor, k: int, thr: float = float("-inf")):
    """``select_topk`` + ``gather_keypoints`` as one library call (bit-identical results; one trip through ctypes).

    Returns (idx [B,k], val [B,k], count [B], pts [B,k,2], descriptors [B,k,C])."""
    if score.device.type != "cuda":
        raise RuntimeError("select_and_gather runs on the HIP device only")
    lib = _lib.load()
    B, Cd = desc.shape[0], desc.shape[1]
    s = score.reshape(B, -1).contiguous().float()
    n = s.shape[1]
    if n != desc.shape[2] * desc.shape[3]:
        raise ValueError("score and descriptor grids differ")
    if k < 1:
        raise ValueError("k must be >= 1 (use k = number of cells for 'every cell above the threshold')")
    k = int(min(k, n))
    coord, desc = coord.contiguous(), desc.contiguous()
    idx = torch.empty(B, k, dtype=torch.int32, device=s.device)
    val = torch.empty(B, k, dtype=torch.float32, device=s.device)
    cnt = torch.empty(B, dtype=torch.int32, device=s.device)
    pts = torch.empty(B, k, 2, device=s.device)
    dsel = torch.empty(B, k, Cd, device=s.device)
    stream = torch.cuda.current_stream(s.device).cuda_stream
    _lib.check(lib.kp2d_select_keypoints(_ptr(s), _ptr(coord), _ptr(desc), B, Cd, n, k, float(thr), _ptr(idx), _ptr(val),
                                         _ptr(cnt), _ptr(pts), _ptr(dsel), C.c_void_p(stream)))
    return idx, val, cnt, pts, dsel


def _cap(top_k: int, score: torch.Tensor) -> int:
    """The reference's cap semantics (``if len(score) > top_k and top_k > 0``): top_k <= 0 means no cap."""
    n = score[0].numel()
    return n if top_k <= 0 else min(int(top_k), n)


def select_keypoints(out: dict, nn_thresh: float = 0.7, top_k: int = 4000, scale=None):
    """K1/K2 for a whole batch.  ``out`` is the dict returned by ``post_processing``.

    Returns a list (one entry per frame) of (pts [n,2], desc [n,C], idx [n]) device tensors, n <= top_k.
    ``top_k <= 0`` keeps every cell above the threshold, as the reference does (visual_odometry.py:112).
    ``scale`` = (scale_x, scale_y) divides the coordinates as the VO front-end does when it resized
    the frame (evaluation/visual_odometry.py:119-121).
    """
    score, coord, feat = out["score"], out["coord"], out["feat"]
    idx, _val, cnt, pts, dsel = select_and_gather(score, coord, feat, _cap(top_k, score), nn_thresh)
    counts = cnt.tolist()  # the only host sync: one int per frame
    res = []
    for b, n in enumerate(counts):
        p = pts[b, :n]
        if scale is not None:
            p = p / torch.tensor([scale[0], scale[1]], device=p.device, dtype=p.dtype)
        res.append((p, dsel[b, :n], idx[b, :n]))
    return res


def select_keypoints_host(out: dict, nn_thresh: float = 0.7, top_k: int = 4000, scale=None):
    """K1/K2 delivered to the host as the reference's ``inference()`` returns them: per frame (pts [n,2], desc [n,C])
    numpy arrays.  Three device-to-host copies for the whole batch (counts, padded points, padded descriptors) instead
    of two per frame."""
    score, coord, feat = out["score"], out["coord"], out["feat"]
    idx, _val, cnt, pts, dsel = select_and_gather(score, coord, feat, _cap(top_k, score), nn_thresh)
    if scale is not None:
        pts = pts / torch.tensor([scale[0], scale[1]], device=pts.device, dtype=pts.dtype)
    counts = cnt.tolist()
    kmax = max(counts) if counts else 0
    hp, hd = pts[:, :kmax].cpu().numpy(), dsel[:, :kmax].cpu().numpy()
    return [(hp[b, :n].copy(), hd[b, :n].copy()) for b, n in enumerate(counts)]


def extract_topk(out: dict, max_num_keypoints: int = 1024):
    """K3: batched top-k without threshold -> dict like the gluefactory extractor's ``pred``."""
    idx, val, _, pts, dsel = select_and_gather(out["score"], out["coord"], out["feat"], max_num_keypoints)
    return {"keypoints": pts, "keypoint_scores": val, "descriptors": dsel, "indices": idx}
