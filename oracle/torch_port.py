"""CPU baseline port of the kp2dtiny path in plain torch ops (TEST / MEASUREMENT INFRASTRUCTURE ONLY).

This is what ``bench.py``'s ``cpu_baseline`` leg times on the GPU box's host cores: the same dataflow
the reference executes with ``--device cpu`` (torch conv2d / batch_norm / pixel_shuffle / grid_sample on
oneDNN), written from this repo's own restatement (``oracle/kp2d_oracle.py``) because the reference's
files cannot travel to the GPU box.  It is checked against the numpy oracle in
``tests/test_torch_port.py``.  The product path never imports it.

Reference lines restated: see the matching functions in oracle/kp2d_oracle.py (same names).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _t(p, k):
    return p[k]


def cbr(x, p, prefix, leaky=True):
    x = F.conv2d(x.contiguous(), _t(p, f"{prefix}.conv.weight"), None, 1, 1)
    x = F.batch_norm(x, _t(p, f"{prefix}.bn.running_mean"), _t(p, f"{prefix}.bn.running_var"),
                     _t(p, f"{prefix}.bn.weight"), _t(p, f"{prefix}.bn.bias"), False, 0.1, 1e-5)
    return F.leaky_relu(x, 0.01) if leaky else F.relu(x)


def conv_b(x, p, prefix):
    return F.conv2d(x, _t(p, f"{prefix}.weight"), p.get(f"{prefix}.bias"), 1, 1)


def backbone(x, p, cfg):
    lk, ds = cfg["leaky_relu"], cfg["downsample"]
    x = cbr(cbr(x, p, "backbone.conv1a", lk), p, "backbone.conv1b", lk)
    if ds >= 2:
        x = F.max_pool2d(x, 2, 2)
    x = cbr(cbr(x, p, "backbone.conv2a", lk), p, "backbone.conv2b", lk)
    if ds >= 3:
        x = F.max_pool2d(x, 2, 2)
    x = cbr(x, p, "backbone.conv3a", lk)
    skip = cbr(x, p, "backbone.conv3b", lk)
    x = F.max_pool2d(skip, 2, 2) if ds >= 1 else skip
    x = cbr(cbr(x, p, "backbone.conv4a", lk), p, "backbone.conv4b", lk)
    return x, skip


def channel_layernorm(x, g, b, eps=1e-5):
    std = torch.var(x, dim=1, unbiased=False, keepdim=True).sqrt()
    mean = torch.mean(x, dim=1, keepdim=True)
    return (x - mean) / (std + eps) * g + b


def attention_module(x, p, prefix, heads=4):
    pa, pm = f"{prefix}.att", f"{prefix}.mff"
    y = channel_layernorm(x, p[f"{pa}.norm.g"], p[f"{pa}.norm.b"])
    B, C, H, W = y.shape
    d = C // heads
    q = F.conv2d(y, p[f"{pa}.fn.to_q.weight"])
    kv = F.conv2d(y, p[f"{pa}.fn.to_kv.weight"], stride=2)
    k, v = kv[:, :C], kv[:, C:]
    q = q.reshape(B * heads, d, H * W).transpose(1, 2)
    k = k.reshape(B * heads, d, -1)
    v = v.reshape(B * heads, d, -1).transpose(1, 2)
    att = torch.softmax(torch.matmul(q, k) * (d ** -0.5), dim=-1)
    o = torch.matmul(att, v).transpose(1, 2).reshape(B, C, H, W)
    x = F.conv2d(o, p[f"{pa}.fn.to_out.weight"])
    y = channel_layernorm(x, p[f"{pm}.norm.g"], p[f"{pm}.norm.b"])
    y = F.conv2d(y, p[f"{pm}.fn.net.0.weight"], p[f"{pm}.fn.net.0.bias"])
    y = F.conv2d(y, p[f"{pm}.fn.net.1.net.0.weight"], p[f"{pm}.fn.net.1.net.0.bias"], padding=1, groups=y.shape[1])
    y = F.conv2d(y, p[f"{pm}.fn.net.1.net.1.weight"], p[f"{pm}.fn.net.1.net.1.bias"])
    return F.conv2d(F.gelu(y), p[f"{pm}.fn.net.3.weight"], p[f"{pm}.fn.net.3.bias"])


def seg_trunk(x, skip, p, cfg):
    lk, P = cfg["leaky_relu"], "seg_head.convs"
    if cfg["use_attention"]:
        s = cbr(x, p, f"{P}.0", lk)
        s = attention_module(s, p, f"{P}.1")
        s = attention_module(F.max_pool2d(s, 2, 2), p, f"{P}.2")
        s = cbr(s, p, f"{P}.3", lk)
        i = 4
    else:
        s = cbr(cbr(x, p, f"{P}.0", lk), p, f"{P}.1", lk)
        s = F.max_pool2d(s, 2, 2)
        s = cbr(cbr(cbr(s, p, f"{P}.2", lk), p, f"{P}.3", lk), p, f"{P}.4", lk)
        i = 5
    s = torch.cat([F.pixel_shuffle(s, 2), x], 1)
    s = cbr(cbr(s, p, f"{P}.{i}", lk), p, f"{P}.{i + 1}", lk)
    s = torch.cat([F.pixel_shuffle(s, 2), skip], 1)
    return cbr(s, p, f"{P}.{i + 2}", lk), f"{P}.{i + 3}"


def netvlad(x, p, prefix="vlad_head.netvlad"):
    """Literal reference dataflow (materialises [B,K,C,S]) — this is what the CPU path pays for."""
    B, C = x.shape[:2]
    x = F.normalize(x, p=2.0, dim=1)
    w, cent = p[f"{prefix}.conv.weight"], p[f"{prefix}.centroids"]
    K = w.shape[0]
    a = F.softmax(F.conv2d(x, w).view(B, K, -1), dim=1)
    xf = x.view(B, C, -1)
    resid = xf.expand(K, -1, -1, -1).permute(1, 0, 2, 3) - cent.expand(xf.size(-1), -1, -1).permute(1, 2, 0).unsqueeze(0)
    resid = resid * a.unsqueeze(2)
    v = F.normalize(resid.sum(dim=-1), p=2.0, dim=2)
    return F.normalize(v.view(B, -1), p=2.0, dim=1)


def forward(x, p, cfg, eval_mode=True):
    lk = cfg["leaky_relu"]
    xb, skip = backbone(x, p, cfg)
    if cfg["v3"]:
        sl = conv_b(cbr(xb, p, "score_loc_head.convDa", lk), p, "score_loc_head.convDb")
        score, shift = sl[:, 0:1].sigmoid(), sl[:, 1:3].tanh()
        s, last = seg_trunk(xb, skip, p, cfg)
        half = s.shape[1] // 2
        feat = conv_b(s[:, :half], p, "seg_head.featB")
        seg = conv_b(s[:, -half:], p, last)
        if eval_mode:
            seg = torch.softmax(seg, 1)
    else:
        score = conv_b(cbr(xb, p, "score_head.convDa", lk), p, "score_head.convDb").sigmoid()
        shift = conv_b(cbr(xb, p, "loc_head.convDa", lk), p, "loc_head.convDb").tanh()
        d = conv_b(cbr(xb, p, "desc_head.convA", lk), p, "desc_head.convB")
        d = torch.cat([F.pixel_shuffle(d, 2), skip], 1)
        feat = conv_b(cbr(d, p, "desc_head.confAa", lk), p, "desc_head.confBb")
        s, last = seg_trunk(xb, skip, p, cfg)
        seg = conv_b(s, p, last)
    v = cbr(cbr(cbr(xb, p, "vlad_head.convlad1", lk), p, "vlad_head.convlad2", lk), p, "vlad_head.convlad3", lk)
    return {"score": score, "coord": shift, "feat": feat, "vlad": netvlad(v, p), "seg": seg}


def post_processing(out, H, W, cfg):
    score, shift, feat = out["score"], out["coord"], out["feat"]
    B, _, Hc, Wc = score.shape
    mask = torch.ones(B, Hc, Wc)
    mask[:, 0] = 0
    mask[:, Hc - 1] = 0
    mask[:, :, 0] = 0
    mask[:, :, Wc - 1] = 0
    score = score * mask.unsqueeze(1)
    cell = 2 ** cfg["downsample"]
    step = (cell - 1) / 2.0
    ys, xs = torch.meshgrid(torch.arange(Hc, dtype=score.dtype), torch.arange(Wc, dtype=score.dtype), indexing="ij")
    base = torch.stack([xs, ys])[None] * cell + step
    coord = base + shift * (2.0 * step)
    coord = torch.stack([coord[:, 0].clamp(0, W - 1), coord[:, 1].clamp(0, H - 1)], 1)
    grid = torch.stack([coord[:, 0] / ((W - 1) / 2.0) - 1.0, coord[:, 1] / ((H - 1) / 2.0) - 1.0], -1)
    f = F.grid_sample(feat, grid, align_corners=True)
    f = f / f.norm(p=2, dim=1, keepdim=True)
    return {"score": score, "coord": coord, "feat": f, "seg": out["seg"].argmax(1).unsqueeze(1), "vlad": out["vlad"]}


def select(post, thr=0.7, top_k=4000):
    """K1 selection per frame on the host, as the reference's callers do."""
    res = []
    B = post["score"].shape[0]
    for b in range(B):
        s = post["score"][b].reshape(-1)
        keep = torch.nonzero(s > thr).squeeze(1)
        if keep.numel() > top_k:
            keep = keep[s[keep].topk(top_k).indices]
        res.append((post["coord"][b].reshape(2, -1).t()[keep], post["feat"][b].reshape(32, -1).t()[keep]))
    return res


def to_torch(sd):
    import numpy as np
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}
