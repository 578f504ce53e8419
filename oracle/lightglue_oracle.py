"""CPU restatement (numpy) of the reference's LightGlue matcher, inference path only.

TEST INFRASTRUCTURE — not product code.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.

PARITY UNPINNED: /root/reference/lightglue/lightglue.py cannot be imported in the build container (its first lines
import ``omegaconf``, which is not installed) and the reference ships neither weights nor
test vectors for it.  This file restates the source text function by function (citations below); tests cross-check it
against an independent torch.nn.functional formulation (tests/test_lightglue_oracle.py), nothing more.

Everything cites /root/reference/lightglue/lightglue.py unless stated otherwise.  Only the configuration the
reference's callers use is restated: eval mode, flash=False, depth_confidence = width_confidence = -1 (no early
stopping, no point pruning: lightglue.py:420-431 defaults; gluefactory/configs/kp2dtiny_S+lightglue_homography.yaml).
"""
from __future__ import annotations

import math

import numpy as np

# lightglue/lightglue_configs.py:1-22
LIGHT_GLUE_CONFIGS = {
    "S": dict(input_dim=32, descriptor_dim=32, n_layers=4),
    "F": dict(input_dim=64, descriptor_dim=64, n_layers=4),
    "A": dict(input_dim=32, descriptor_dim=32, n_layers=4),
}
DEFAULTS = dict(input_dim=256, descriptor_dim=256, n_layers=9, num_heads=4, add_scale_ori=False,
                filter_threshold=0.0)   # lightglue.py:419-438


def get_config(name_or_conf) -> dict:
    conf = dict(DEFAULTS)
    if isinstance(name_or_conf, str):
        if name_or_conf not in LIGHT_GLUE_CONFIGS:
            raise ValueError("Config not supported")        # lightglue_configs.py:27-28
        conf.update(LIGHT_GLUE_CONFIGS[name_or_conf])
    else:
        conf.update({k: v for k, v in name_or_conf.items() if k in DEFAULTS})
    return conf


def state_dict_shapes(conf: dict) -> dict:
    """{key: shape} in registration order of LightGlue.__init__ (lightglue.py:444-470)."""
    d, din, h, n = conf["descriptor_dim"], conf["input_dim"], conf["num_heads"], conf["n_layers"]
    hd = d // h
    s = {}
    if din != d:
        s["input_proj.weight"], s["input_proj.bias"] = (d, din), (d,)
    s["posenc.Wr.weight"] = (hd // 2, 2 + 2 * int(conf["add_scale_ori"]))        # :161-166

    def ffn(p):                                                                   # :240-245 / :289-294
        s[f"{p}.0.weight"], s[f"{p}.0.bias"] = (2 * d, 2 * d), (2 * d,)
        s[f"{p}.1.weight"], s[f"{p}.1.bias"] = (2 * d,), (2 * d,)
        s[f"{p}.3.weight"], s[f"{p}.3.bias"] = (d, 2 * d), (d,)

    for i in range(n):
        p = f"transformers.{i}.self_attn"                                        # SelfBlock :228-245
        s[f"{p}.Wqkv.weight"], s[f"{p}.Wqkv.bias"] = (3 * d, d), (3 * d,)
        s[f"{p}.out_proj.weight"], s[f"{p}.out_proj.bias"] = (d, d), (d,)
        ffn(f"{p}.ffn")
        p = f"transformers.{i}.cross_attn"                                       # CrossBlock :273-298
        for nm in ("to_qk", "to_v", "to_out"):
            s[f"{p}.{nm}.weight"], s[f"{p}.{nm}.bias"] = (d, d), (d,)
        ffn(f"{p}.ffn")
    for i in range(n):                                                           # MatchAssignment :379-384
        s[f"log_assignment.{i}.matchability.weight"], s[f"log_assignment.{i}.matchability.bias"] = (1, d), (1,)
        s[f"log_assignment.{i}.final_proj.weight"], s[f"log_assignment.{i}.final_proj.bias"] = (d, d), (d,)
    for i in range(n - 1):                                                       # TokenConfidence :181-184
        s[f"token_confidence.{i}.token.0.weight"], s[f"token_confidence.{i}.token.0.bias"] = (1, d), (1,)
    return s


def seeded_state_dict(conf: dict, seed: int = 4321) -> dict:
    """Seeded spread weights for parity work (generator shared with the tools: nano-vs-slam_amd/synthetic.py)."""
    from oracle.weights import seeded_linear_state_dict
    return seeded_linear_state_dict(state_dict_shapes(conf), seed)


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def linear(x, p, prefix):
    w = p[f"{prefix}.weight"].astype(x.dtype)
    y = x @ w.T
    b = p.get(f"{prefix}.bias")
    return y if b is None else y + b.astype(x.dtype)


def normalize_keypoints(kpts, size=None):
    """:137-149.  size None: 1 + max - min over the keypoints; else the image size [.., 2] (w, h)."""
    if size is None:
        size = 1 + kpts.max(-2) - kpts.min(-2)
    size = np.asarray(size, kpts.dtype)
    if size.ndim == 1:
        size = np.broadcast_to(size, kpts.shape[:-2] + (2,))
    shift = size / 2
    scale = size.max(-1) / 2
    return (kpts - shift[..., None, :]) / scale[..., None, None]


def posenc(kpts, p):
    """LearnableFourierPositionalEncoding.forward :168-173 -> [2, B, 1, M, head_dim] (cos, sin; each frequency
    repeated for the two members of a rotary pair)."""
    proj = kpts @ p["posenc.Wr.weight"].astype(kpts.dtype).T
    emb = np.stack([np.cos(proj), np.sin(proj)], 0)[:, :, None]
    return np.repeat(emb, 2, axis=-1)


def rotate_half(x):
    """:152-155: (x0, x1) -> (-x1, x0) on consecutive pairs."""
    x = x.reshape(x.shape[:-1] + (-1, 2))
    return np.stack((-x[..., 1], x[..., 0]), axis=-1).reshape(x.shape[:-2] + (-1,))


def apply_rotary(freqs, t):
    """:158-159."""
    return t * freqs[0] + rotate_half(t) * freqs[1]


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def attention(q, k, v):
    """Attention.forward :208-224 (all three branches compute softmax(q k^T / sqrt(d)) v without a mask)."""
    s = q.shape[-1] ** -0.5
    return softmax(np.einsum("...id,...jd->...ij", q, k) * s, -1) @ v


def layer_norm(x, g, b, eps=1e-5):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + x.dtype.type(eps)) * g.astype(x.dtype) + b.astype(x.dtype)


def gelu(x):
    erf = np.vectorize(math.erf, otypes=[np.float64])
    return (0.5 * x * (1.0 + erf(x.astype(np.float64) / math.sqrt(2.0)))).astype(x.dtype)


def ffn(x, p, prefix):
    """nn.Sequential(Linear(2d,2d), LayerNorm(2d), GELU, Linear(2d,d)) :240-245."""
    h = linear(x, p, f"{prefix}.0")
    h = gelu(layer_norm(h, p[f"{prefix}.1.weight"], p[f"{prefix}.1.bias"]))
    return linear(h, p, f"{prefix}.3")


def self_block(x, enc, p, prefix, heads):
    """SelfBlock.forward :247-261."""
    B, M, d = x.shape
    qkv = linear(x, p, f"{prefix}.Wqkv").reshape(B, M, heads, d // heads, 3).transpose(0, 2, 1, 3, 4)
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q, k = apply_rotary(enc, q), apply_rotary(enc, k)
    ctx = attention(q, k, v).transpose(0, 2, 1, 3).reshape(B, M, d)
    msg = linear(ctx, p, f"{prefix}.out_proj")
    return x + ffn(np.concatenate([x, msg], -1), p, f"{prefix}.ffn")


def cross_block(x0, x1, p, prefix, heads):
    """CrossBlock.forward :303-327, non-flash branch (flash=False in every reference config)."""
    B, M, d = x0.shape
    hd = d // heads
    split = lambda t: t.reshape(t.shape[0], t.shape[1], heads, hd).transpose(0, 2, 1, 3)
    qk0, qk1 = split(linear(x0, p, f"{prefix}.to_qk")), split(linear(x1, p, f"{prefix}.to_qk"))
    v0, v1 = split(linear(x0, p, f"{prefix}.to_v")), split(linear(x1, p, f"{prefix}.to_v"))
    sc = x0.dtype.type((hd ** -0.5) ** 0.5)
    sim = np.einsum("bhid,bhjd->bhij", qk0 * sc, qk1 * sc)
    m0 = softmax(sim, -1) @ v1
    m1 = softmax(sim.transpose(0, 1, 3, 2), -1) @ v0
    merge = lambda t: t.transpose(0, 2, 1, 3).reshape(t.shape[0], t.shape[2], d)
    m0, m1 = linear(merge(m0), p, f"{prefix}.to_out"), linear(merge(m1), p, f"{prefix}.to_out")
    x0 = x0 + ffn(np.concatenate([x0, m0], -1), p, f"{prefix}.ffn")
    x1 = x1 + ffn(np.concatenate([x1, m1], -1), p, f"{prefix}.ffn")
    return x0, x1


def log_sigmoid(x):
    return np.where(x >= 0, -np.log1p(np.exp(-np.abs(x))), x - np.log1p(np.exp(-np.abs(x))))


def log_softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    return x - m - np.log(np.exp(x - m).sum(axis=axis, keepdims=True))


def match_assignment(d0, d1, p, prefix):
    """MatchAssignment.forward :386-395 + sigmoid_log_double_softmax :363-376 -> (scores [B,M+1,N+1], sim)."""
    B, M, d = d0.shape
    N = d1.shape[1]
    m0 = linear(d0, p, f"{prefix}.final_proj") / d0.dtype.type(d ** 0.25)
    m1 = linear(d1, p, f"{prefix}.final_proj") / d0.dtype.type(d ** 0.25)
    sim = np.einsum("bmd,bnd->bmn", m0, m1)
    z0, z1 = linear(d0, p, f"{prefix}.matchability"), linear(d1, p, f"{prefix}.matchability")   # [B,M,1]
    cert = log_sigmoid(z0) + log_sigmoid(z1).transpose(0, 2, 1)
    scores = np.zeros((B, M + 1, N + 1), d0.dtype)
    scores[:, :M, :N] = log_softmax(sim, 2) + log_softmax(sim, 1) + cert
    scores[:, :M, N] = log_sigmoid(-z0[..., 0])
    scores[:, M, :N] = log_sigmoid(-z1[..., 0])
    return scores, sim


def filter_matches(scores, th):
    """:401-416."""
    inner = scores[:, :-1, :-1]
    m0, m1 = inner.argmax(2), inner.argmax(1)
    max0 = inner.max(2)
    B, M = m0.shape
    N = m1.shape[1]
    mutual0 = np.arange(M)[None] == np.take_along_axis(m1, m0, 1)
    mutual1 = np.arange(N)[None] == np.take_along_axis(m0, m1, 1)
    ms0 = np.where(mutual0, np.exp(max0), 0).astype(scores.dtype)
    ms1 = np.where(mutual1, np.take_along_axis(ms0, m1, 1), 0).astype(scores.dtype)
    valid0 = mutual0 & (ms0 > th)
    valid1 = mutual1 & np.take_along_axis(valid0, m1, 1)
    return np.where(valid0, m0, -1), np.where(valid1, m1, -1), ms0, ms1


def forward(data: dict, p: dict, conf: dict) -> dict:
    """LightGlue.forward :484-614, eval mode without early stopping / pruning.

    data: keypoints0 [B,M,2], keypoints1 [B,N,2], descriptors0 [B,M,Din], descriptors1 [B,N,Din] and (as every
    reference caller passes) view0/view1 = {"image_size": [B,2] or [2]}; image_size None -> keypoint extents.
    """
    k0, k1 = data["keypoints0"], data["keypoints1"]
    s0 = data.get("view0", {}).get("image_size")
    s1 = data.get("view1", {}).get("image_size")
    k0, k1 = normalize_keypoints(k0, s0), normalize_keypoints(k1, s1)
    d0, d1 = data["descriptors0"], data["descriptors1"]
    assert d0.shape[-1] == conf["input_dim"] and d1.shape[-1] == conf["input_dim"]
    if conf["input_dim"] != conf["descriptor_dim"]:
        d0, d1 = linear(d0, p, "input_proj"), linear(d1, p, "input_proj")
    e0, e1 = posenc(k0, p), posenc(k1, p)
    h = conf["num_heads"]
    for i in range(conf["n_layers"]):
        d0 = self_block(d0, e0, p, f"transformers.{i}.self_attn", h)
        d1 = self_block(d1, e1, p, f"transformers.{i}.self_attn", h)
        d0, d1 = cross_block(d0, d1, p, f"transformers.{i}.cross_attn", h)
    last = conf["n_layers"] - 1
    scores, _ = match_assignment(d0, d1, p, f"log_assignment.{last}")
    m0, m1, ms0, ms1 = filter_matches(scores, conf["filter_threshold"])
    return {
        "matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
        "ref_descriptors0": d0[:, None], "ref_descriptors1": d1[:, None],
        "log_assignment": scores,
        "prune0": np.ones_like(ms0) * conf["n_layers"], "prune1": np.ones_like(ms1) * conf["n_layers"],
    }
