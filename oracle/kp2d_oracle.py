"""CPU oracle for the kp2dtiny multi-task inference path (numpy restatement).

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is product code: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it, and only as the checker.  The product path
(``nano-vs-slam_amd``) never imports this module and raises if its HIP library
is missing.

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference
(``/root/reference``, this container only) with the seeded spread weights of
``oracle/weights.py`` and stores its outputs under ``tests/golden``;
``tests/test_oracle_golden.py`` checks every function below against them.
The reference itself ships no golden vectors or known-answer tests
(SURVEY.md §4), so those fixtures are the pin.

Every function cites the reference file:line it restates (paths relative to
``/root/reference``).  Arrays are NCHW; ``dtype`` selects float32 (the
reference's arithmetic) or float64 (used to size tolerances).
"""
from __future__ import annotations

import math

import numpy as np

# --------------------------------------------------------------------------
# configuration tables — restated from src/kp2dtiny/models/kp2dtiny.py:46-218
# --------------------------------------------------------------------------
_S = dict(nfeatures=32, channel_dims=[16, 32, 32, 64, 64, 128], downsample=2, leaky_relu=True, encoder_dim=64)
_N = dict(nfeatures=32, channel_dims=[16, 24, 24, 48, 48, 96], downsample=2, leaky_relu=True, encoder_dim=48,
          num_clusters=32)
_D = dict(nfeatures=128, channel_dims=[64, 128, 128, 256, 256, 512], downsample=2, leaky_relu=True, encoder_dim=128,
          global_descriptor_method="convap")
V2_CONFIGS = {
    "D": dict(_D, use_attention=True),                                                     # LARGE_D :168-176
    "F": dict(nfeatures=64, channel_dims=[16, 32, 64, 128, 128, 256], downsample=3,        # TINY_F :113-119
              use_attention=False, leaky_relu=True),
    "S": dict(_S, use_attention=False),
    "S_A": dict(_S, use_attention=True),
    "N": dict(_N, use_attention=False),
    "N_A": dict(_N, use_attention=True),
    "GEM_N": dict(_N, use_attention=False, global_descriptor_method="gem"),
    "GEM_S_A": dict(_S, use_attention=True, global_descriptor_method="gem"),
    "CONVAP_S_A": dict(_S, use_attention=True, global_descriptor_method="convap"),
}
V3_CONFIGS = {
    "D": dict(_D, use_attention=False),                                                    # LARGE_D_V3 :187-195
    "D_A": dict(_D, use_attention=True),                                                   # LARGE_D_A_V3 :177-185
    "S": dict(_S, use_attention=False),
    "S_A": dict(_S, use_attention=True),
    # V3_N / V3_N_A carry no num_clusters key -> constructor default 64 (kp2dtiny.py:151-166,690)
    "N": dict(nfeatures=32, channel_dims=[16, 24, 24, 48, 48, 96], downsample=2, leaky_relu=True,
              encoder_dim=48, use_attention=False),
    "N_A": dict(nfeatures=32, channel_dims=[16, 24, 24, 48, 48, 96], downsample=2, leaky_relu=True,
                encoder_dim=48, use_attention=True),
    "CONVAP_S_A": dict(_S, use_attention=True, global_descriptor_method="convap"),
}


def get_config(name: str, v3: bool = False) -> dict:
    """``name`` may carry test-fixture suffixes: "+depth" (constructor depth=True), "+mcu" (tiny_factory
    to_mcu=True: upscale_method="convtranspose", leaky_relu=False — kp2dtiny.py:271-273) and "+gray"
    (KP2DTinyV3(use_color=False): one input channel, kp2dtiny.py:718-721)."""
    mods = name.split("+")[1:]
    name = name.split("+")[0]
    table = V3_CONFIGS if v3 else V2_CONFIGS
    if name not in table:
        raise ValueError(f"Config {name} not supported by the oracle, choose from {list(table)}")
    cfg = dict(table[name])
    cfg.setdefault("num_clusters", 64)  # kp2dtiny.py:308 / :690
    cfg.setdefault("encoder_dim", cfg["channel_dims"][3])  # kp2dtiny.py:342-345 / :727-730 (default c4)
    cfg.setdefault("global_descriptor_method", "netvlad")
    cfg.setdefault("remove_netvlad", False)
    cfg.setdefault("depth", False)
    cfg.setdefault("upscale_method", "pixelshuffle")   # "convtranspose" under to_mcu (kp2dtiny.py:271-273)
    if "depth" in mods:
        cfg["depth"] = True
    if "mcu" in mods:
        cfg["upscale_method"], cfg["leaky_relu"] = "convtranspose", False
    cfg["in_channels"] = 1 if "gray" in mods else 3
    if "gray" in mods and not v3:
        raise ValueError("use_color is a KP2DTinyV3 argument (kp2dtiny.py:682); KP2DTinyV2 always reads RGB")
    cfg["v3"] = v3
    return cfg


# --------------------------------------------------------------------------
# state-dict key/shape layout — SURVEY.md App. C, read off the constructors
# --------------------------------------------------------------------------
def _cbr_shapes(prefix, ci, co):
    return {
        f"{prefix}.conv.weight": (co, ci, 3, 3),
        f"{prefix}.bn.weight": (co,),
        f"{prefix}.bn.bias": (co,),
        f"{prefix}.bn.running_mean": (co,),
        f"{prefix}.bn.running_var": (co,),
        f"{prefix}.bn.num_batches_tracked": (),
    }


def _conv_shapes(prefix, ci, co, k=3, bias=True, groups=1):
    d = {f"{prefix}.weight": (co, ci // groups, k, k)}
    if bias:
        d[f"{prefix}.bias"] = (co,)
    return d


def _tconv_shapes(prefix, c):
    """TransposedConvUpsampleModel(c): modules/base.py:80-117."""
    return {
        f"{prefix}.transposed_conv.weight": (c, c // 4, 3, 3),
        f"{prefix}.bn.weight": (c // 4,),
        f"{prefix}.bn.bias": (c // 4,),
        f"{prefix}.bn.running_mean": (c // 4,),
        f"{prefix}.bn.running_var": (c // 4,),
        f"{prefix}.bn.num_batches_tracked": (),
    }


def _attmod_shapes(prefix, c):
    """SegFormerAttentionModule(c): modules/segformer.py:209-220 (heads 4, reduction 2, expansion 2)."""
    d = {}
    # PreNorm registers ``fn`` before ``norm`` (modules/segformer.py:78-81)
    d.update(_conv_shapes(f"{prefix}.att.fn.to_q", c, c, 1, bias=False))
    d.update(_conv_shapes(f"{prefix}.att.fn.to_kv", c, 2 * c, 2, bias=False))
    d.update(_conv_shapes(f"{prefix}.att.fn.to_out", c, c, 1, bias=False))
    d[f"{prefix}.att.norm.g"] = (1, c, 1, 1)
    d[f"{prefix}.att.norm.b"] = (1, c, 1, 1)
    h = 2 * c
    d.update(_conv_shapes(f"{prefix}.mff.fn.net.0", c, h, 1))
    d.update(_conv_shapes(f"{prefix}.mff.fn.net.1.net.0", h, h, 3, groups=h))
    d.update(_conv_shapes(f"{prefix}.mff.fn.net.1.net.1", h, h, 1))
    d.update(_conv_shapes(f"{prefix}.mff.fn.net.3", h, c, 1))
    d[f"{prefix}.mff.norm.g"] = (1, c, 1, 1)
    d[f"{prefix}.mff.norm.b"] = (1, c, 1, 1)
    return d


def state_dict_shapes(cfg: dict, n_classes: int) -> dict:
    """{key: shape} in the reference's registration order (kp2dtiny.py:347-449 / :732-803)."""
    c1, c2, c3, c4, c5, d1 = cfg["channel_dims"]
    nf, K, enc = cfg["nfeatures"], cfg["num_clusters"], cfg["encoder_dim"]
    v3, att = cfg["v3"], cfg["use_attention"]
    s = {}
    for name, ci, co in [("conv1a", cfg.get("in_channels", 3), c1), ("conv1b", c1, c2), ("conv2a", c2, c2), ("conv2b", c2, c3),
                         ("conv3a", c3, c3), ("conv3b", c3, c4), ("conv4a", c4, c4), ("conv4b", c4, c4)]:
        s.update(_cbr_shapes(f"backbone.{name}", ci, co))
    if v3:
        s.update(_cbr_shapes("score_loc_head.convDa", c4, c4))
        s.update(_conv_shapes("score_loc_head.convDb", c4, 3))
    else:
        s.update(_cbr_shapes("score_head.convDa", c4, c4))
        s.update(_conv_shapes("score_head.convDb", c4, 1))
        s.update(_cbr_shapes("loc_head.convDa", c4, c4))
        s.update(_conv_shapes("loc_head.convDb", c4, 2))
        # UpscaleHead(c4, c4, c3*4, c3+c4, c4, nfeatures): kp2dtiny.py:377-388; upsample registered first (heads.py:53-58)
        if cfg.get("upscale_method", "pixelshuffle") == "convtranspose":
            s.update(_tconv_shapes("desc_head.upsample", c3 * 4))
        s.update(_cbr_shapes("desc_head.convA", c4, c4))
        s.update(_conv_shapes("desc_head.convB", c4, c3 * 4))
        s.update(_cbr_shapes("desc_head.confAa", c3 + c4, c4))
        s.update(_conv_shapes("desc_head.confBb", c4, nf))
    # seg head: (c_in=c4, c_hidden=c5, c_exp=c4+c3, c_out=nClasses, d1)
    ch, cexp = c5, c4 + c3
    depth = cfg.get("depth", False)
    tconv = cfg.get("upscale_method", "pixelshuffle") == "convtranspose"

    def upsamplers(prefix):      # registered after convs / featB / featD (segmentation.py:113-118, :290-297)
        if tconv:
            s.update(_tconv_shapes(f"{prefix}.upsample", d1))
            s.update(_tconv_shapes(f"{prefix}.upsample2", d1))

    def seg_like(prefix, c_out, width, last_in):
        P = f"{prefix}.convs"
        if att:
            s.update(_cbr_shapes(f"{P}.0", c4, ch))
            s.update(_attmod_shapes(f"{P}.1", ch))
            s.update(_attmod_shapes(f"{P}.2", ch))
            s.update(_cbr_shapes(f"{P}.3", ch, d1))
            s.update(_cbr_shapes(f"{P}.4", ch + d1 // 4, ch))
            s.update(_cbr_shapes(f"{P}.5", ch, d1))
            s.update(_cbr_shapes(f"{P}.6", cexp, width))
            s.update(_conv_shapes(f"{P}.7", last_in, c_out))
        else:
            s.update(_cbr_shapes(f"{P}.0", c4, ch))
            s.update(_cbr_shapes(f"{P}.1", ch, ch))
            s.update(_cbr_shapes(f"{P}.2", ch, ch))
            s.update(_cbr_shapes(f"{P}.3", ch, ch))
            s.update(_cbr_shapes(f"{P}.4", ch, d1))
            s.update(_cbr_shapes(f"{P}.5", ch + d1 // 4, ch))
            s.update(_cbr_shapes(f"{P}.6", ch, d1))
            s.update(_cbr_shapes(f"{P}.7", cexp, width))
            s.update(_conv_shapes(f"{P}.8", last_in, c_out))

    if v3:
        # depth=True widens the last CBR by c_hidden//2 and adds featD (segmentation.py:190-193, 281-284)
        seg_like("seg_head", n_classes, ch + ch // 2 if depth else ch, ch // 2)
        s.update(_conv_shapes("seg_head.featB", ch // 2, nf))
        if depth:
            s.update(_conv_shapes("seg_head.featD", ch // 2, 1, bias=False))
        upsamplers("seg_head")
    else:
        seg_like("seg_head", n_classes, ch, ch)
        upsamplers("seg_head")
        if depth:
            seg_like("depth_head", 1, ch, ch)            # kp2dtiny.py:402-437
            upsamplers("depth_head")
    for i in (1, 2, 3):
        s.update(_cbr_shapes(f"vlad_head.convlad{i}", c4 if i == 1 else enc, enc))
    method = cfg.get("global_descriptor_method", "netvlad")
    if method == "netvlad":
        if not cfg.get("remove_netvlad", False):
            s["vlad_head.netvlad.centroids"] = (K, enc)
            s.update(_conv_shapes("vlad_head.netvlad.conv", enc, K, 1, bias=False))
    elif method == "gem":
        s["vlad_head.netvlad.p"] = (1,)                                  # aggregators/gem.py:10
    elif method == "convap":
        s.update(_conv_shapes("vlad_head.netvlad.channel_pool", enc, enc, 1))   # aggregators/convap.py:23-25
    return s


# --------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------
def conv2d_3x3(x, w, b=None):
    """Conv2d(k=3, s=1, p=1) — torch.nn.Conv2d as used at modules/base.py:28-30."""
    B, C, H, W = x.shape
    Co = w.shape[0]
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros((B, Co, H * W), x.dtype)
    for dy in range(3):
        for dx in range(3):
            patch = np.ascontiguousarray(xp[:, :, dy:dy + H, dx:dx + W]).reshape(B, C, H * W)
            out += np.matmul(w[:, :, dy, dx].astype(x.dtype), patch)
    out = out.reshape(B, Co, H, W)
    if b is not None:
        out += b.astype(x.dtype)[None, :, None, None]
    return out


def conv2d_1x1(x, w, b=None):
    B, C, H, W = x.shape
    out = np.matmul(w.reshape(w.shape[0], C).astype(x.dtype), x.reshape(B, C, H * W)).reshape(B, -1, H, W)
    if b is not None:
        out = out + b.astype(x.dtype)[None, :, None, None]
    return out


def conv2d_2x2_s2(x, w):
    """Conv2d(k=2, s=2, no bias) — EfficientSelfAttention.to_kv, modules/segformer.py:93-95."""
    B, C, H, W = x.shape
    out = np.zeros((B, w.shape[0], (H // 2) * (W // 2)), x.dtype)
    for dy in range(2):
        for dx in range(2):
            patch = np.ascontiguousarray(x[:, :, dy:2 * (H // 2):2, dx:2 * (W // 2):2]).reshape(B, C, -1)
            out += np.matmul(w[:, :, dy, dx].astype(x.dtype), patch)
    return out.reshape(B, -1, H // 2, W // 2)


def dwconv2d_3x3(x, w, b):
    """Depthwise 3x3 (groups=C, p=1, bias) — DsConv2d.net[0], modules/segformer.py:47-56."""
    B, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)))
    out = np.zeros_like(x)
    for dy in range(3):
        for dx in range(3):
            out += xp[:, :, dy:dy + H, dx:dx + W] * w[:, 0, dy, dx].astype(x.dtype)[None, :, None, None]
    return out + b.astype(x.dtype)[None, :, None, None]


def batchnorm_eval(x, p, prefix, eps=1e-5):
    """BatchNorm2d in eval mode (running stats, eps 1e-5) — modules/base.py:31,43."""
    dt = x.dtype
    mean = p[f"{prefix}.running_mean"].astype(dt)[None, :, None, None]
    var = p[f"{prefix}.running_var"].astype(dt)[None, :, None, None]
    g = p[f"{prefix}.weight"].astype(dt)[None, :, None, None]
    b = p[f"{prefix}.bias"].astype(dt)[None, :, None, None]
    return (x - mean) / np.sqrt(var + dt.type(eps)) * g + b


def act(x, leaky=True):
    """LeakyReLU(0.01) or ReLU — modules/base.py:32-35."""
    return np.where(x >= 0, x, x * x.dtype.type(0.01)) if leaky else np.maximum(x, 0)


def cbr(x, p, prefix, leaky=True):
    """AnnotatedConvBnReLUModel.forward — modules/base.py:39-46 (Quant/DeQuant stubs are identities)."""
    return act(batchnorm_eval(conv2d_3x3(x, p[f"{prefix}.conv.weight"]), p, f"{prefix}.bn"), leaky)


def conv_b(x, p, prefix):
    """Plain Conv2d 3x3 with bias, no BN, no activation (heads.py:22,72-74,85; segmentation.py:108)."""
    return conv2d_3x3(x, p[f"{prefix}.weight"], p.get(f"{prefix}.bias"))


def maxpool2(x):
    """MaxPool2d(2,2), floor mode — encoders.py:100."""
    B, C, H, W = x.shape
    x = x[:, :, : H // 2 * 2, : W // 2 * 2].reshape(B, C, H // 2, 2, W // 2, 2)
    return x.max(axis=(3, 5))


def pixel_shuffle2(x):
    """PixelShuffle(2): out[c,2h+i,2w+j] = in[4c+2i+j,h,w] — heads.py:54."""
    B, C, H, W = x.shape
    x = x.reshape(B, C // 4, 2, 2, H, W).transpose(0, 1, 4, 2, 5, 3)
    return np.ascontiguousarray(x).reshape(B, C // 4, 2 * H, 2 * W)


def conv_transpose2d_3x3_s2(x, w):
    """ConvTranspose2d(k=3, stride=2, padding=1, output_padding=1, bias=False) — base.py:93-101, scatter form:
    out[2*iy - 1 + ky, 2*ix - 1 + kx] += x[iy, ix] * w[ci, co, ky, kx]; output is exactly 2H x 2W."""
    B, C, H, W = x.shape
    Co = w.shape[1]
    full = np.zeros((B, Co, 2 * H + 2, 2 * W + 2), x.dtype)      # index = output coordinate + 1
    xf = x.reshape(B, C, H * W)
    for ky in range(3):
        for kx in range(3):
            contrib = np.matmul(w[:, :, ky, kx].T.astype(x.dtype), xf).reshape(B, Co, H, W)
            full[:, :, ky:ky + 2 * H:2, kx:kx + 2 * W:2] += contrib
    return np.ascontiguousarray(full[:, :, 1:2 * H + 1, 1:2 * W + 1])


def upsample2(x, p, prefix, cfg):
    """heads.py:53-58 / segmentation.py:113-118: PixelShuffle(2), or TransposedConvUpsampleModel.forward
    (base.py:109-117: transposed conv -> BN -> (Leaky)ReLU) when upscale_method == "convtranspose"."""
    if cfg.get("upscale_method", "pixelshuffle") == "pixelshuffle":
        return pixel_shuffle2(x)
    y = conv_transpose2d_3x3_s2(x, p[f"{prefix}.transposed_conv.weight"])
    return act(batchnorm_eval(y, p, f"{prefix}.bn"), cfg["leaky_relu"])


def softmax(x, axis):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def l2_normalize(x, axis, eps=1e-12):
    """F.normalize(p=2): x / max(||x||, eps) — modules/base.py:11, aggregators/netvlad.py:83."""
    n = np.sqrt((x * x).sum(axis=axis, keepdims=True))
    return x / np.maximum(n, x.dtype.type(eps))


_erf = np.vectorize(math.erf, otypes=[np.float64])


def gelu_erf(x):
    """nn.GELU() default = exact erf form — modules/segformer.py:185."""
    y = 0.5 * x.astype(np.float64) * (1.0 + _erf(x.astype(np.float64) / math.sqrt(2.0)))
    return y.astype(x.dtype)


# --------------------------------------------------------------------------
# modules
# --------------------------------------------------------------------------
def backbone(x, p, cfg, taps=None):
    """BackBone.forward — modules/encoders.py:105-129 (Dropout2d inactive in eval)."""
    lk, ds = cfg["leaky_relu"], cfg["downsample"]
    x = cbr(x, p, "backbone.conv1a", lk)
    if taps is not None:
        taps["backbone.conv1a"] = x
    x = cbr(x, p, "backbone.conv1b", lk)
    if ds >= 2:
        x = maxpool2(x)
    x = cbr(x, p, "backbone.conv2a", lk)
    x = cbr(x, p, "backbone.conv2b", lk)
    if ds >= 3:
        x = maxpool2(x)
    x = cbr(x, p, "backbone.conv3a", lk)
    skip = cbr(x, p, "backbone.conv3b", lk)
    if ds >= 1:
        x = maxpool2(skip)
    x = cbr(x, p, "backbone.conv4a", lk)
    x = cbr(x, p, "backbone.conv4b", lk)
    if taps is not None:
        taps["backbone.skip"] = skip
        taps["backbone.x"] = x
    return x, skip


def simple_task_head(x, p, prefix, lk):
    """SimpleTaskHead.forward — modules/decoders/heads.py:28-35."""
    return conv_b(cbr(x, p, f"{prefix}.convDa", lk), p, f"{prefix}.convDb")


def upscale_head(x, skip, p, prefix, lk, cfg=None):
    """UpscaleHead.forward — modules/decoders/heads.py:91-104."""
    x = cbr(x, p, f"{prefix}.convA", lk)
    x = conv_b(x, p, f"{prefix}.convB")
    x = upsample2(x, p, f"{prefix}.upsample", cfg or {})
    x = np.concatenate([x, skip], axis=1)
    x = cbr(x, p, f"{prefix}.confAa", lk)
    return conv_b(x, p, f"{prefix}.confBb")


def channel_layernorm(x, g, b, eps=1e-5):
    """Custom LayerNorm: eps added to the STD, biased variance — modules/segformer.py:70-73."""
    dt = x.dtype
    mean = x.mean(axis=1, keepdims=True)
    std = np.sqrt(((x - mean) ** 2).mean(axis=1, keepdims=True))
    return (x - mean) / (std + dt.type(eps)) * g.astype(dt) + b.astype(dt)


def efficient_self_attention(x, p, prefix, heads=4, kv_block=None):
    """EfficientSelfAttention.forward — modules/segformer.py:100-138.

    q = 1x1 conv; kv = 2x2 stride-2 conv; k = first C channels, v = last C;
    head h owns channels [h*d, (h+1)*d); softmax(q k^T * d^-0.5) v; to_out 1x1.
    ``kv_block`` is ignored (kept for signature parity with the streaming variant).
    """
    B, C, H, W = x.shape
    d = C // heads
    scale = x.dtype.type(d ** -0.5)
    q = conv2d_1x1(x, p[f"{prefix}.to_q.weight"])
    kv = conv2d_2x2_s2(x, p[f"{prefix}.to_kv.weight"])
    k, v = kv[:, :C], kv[:, C:]
    qh = q.reshape(B, heads, d, H * W).transpose(0, 1, 3, 2)          # B,h,S,d
    kh = k.reshape(B, heads, d, -1)                                     # B,h,d,T
    vh = v.reshape(B, heads, d, -1).transpose(0, 1, 3, 2)              # B,h,T,d
    out = np.empty_like(qh)
    # row-blocked so the S x T score matrix is never held whole (1.47 GB/frame at 480x640)
    step = 4096
    for s0 in range(0, H * W, step):
        sim = np.matmul(qh[:, :, s0:s0 + step], kh) * scale
        out[:, :, s0:s0 + step] = np.matmul(softmax(sim, axis=-1), vh)
    out = out.transpose(0, 1, 3, 2).reshape(B, C, H, W)
    return conv2d_1x1(out, p[f"{prefix}.to_out.weight"])


def mix_feed_forward(x, p, prefix):
    """MixFeedForward.forward — modules/segformer.py:182-206 (1x1, dw3x3, 1x1, GELU, 1x1; all biased)."""
    x = conv2d_1x1(x, p[f"{prefix}.net.0.weight"], p[f"{prefix}.net.0.bias"])
    x = dwconv2d_3x3(x, p[f"{prefix}.net.1.net.0.weight"], p[f"{prefix}.net.1.net.0.bias"])
    x = conv2d_1x1(x, p[f"{prefix}.net.1.net.1.weight"], p[f"{prefix}.net.1.net.1.bias"])
    x = gelu_erf(x)
    return conv2d_1x1(x, p[f"{prefix}.net.3.weight"], p[f"{prefix}.net.3.bias"])


def attention_module(x, p, prefix, taps=None):
    """SegFormerAttentionModule.forward — modules/segformer.py:217-220 (PreNorm twice, NO residuals)."""
    x = efficient_self_attention(
        channel_layernorm(x, p[f"{prefix}.att.norm.g"], p[f"{prefix}.att.norm.b"]), p, f"{prefix}.att.fn")
    if taps is not None:
        taps[f"{prefix}.att"] = x
    x = mix_feed_forward(
        channel_layernorm(x, p[f"{prefix}.mff.norm.g"], p[f"{prefix}.mff.norm.b"]), p, f"{prefix}.mff.fn")
    if taps is not None:
        taps[f"{prefix}.mff"] = x
    return x


def seg_trunk(x, skip, p, cfg, taps=None, head="seg_head"):
    """Shared trunk of the four segmentation heads up to the last CBR(c_exp -> c_hidden).

    no-att: modules/decoders/segmentation.py:126-152 (V2) / :321-338 (V3)
    att:    :442-463 (V2) / :588-609 (V3)
    """
    lk = cfg["leaky_relu"]
    P = f"{head}.convs"
    if cfg["use_attention"]:
        seg = cbr(x, p, f"{P}.0", lk)
        seg = attention_module(seg, p, f"{P}.1", taps)
        seg = maxpool2(seg)
        seg = attention_module(seg, p, f"{P}.2", taps)
        seg = cbr(seg, p, f"{P}.3", lk)
        i = 4
    else:
        seg = cbr(x, p, f"{P}.0", lk)
        seg = cbr(seg, p, f"{P}.1", lk)
        seg = maxpool2(seg)
        seg = cbr(seg, p, f"{P}.2", lk)
        seg = cbr(seg, p, f"{P}.3", lk)
        seg = cbr(seg, p, f"{P}.4", lk)
        i = 5
    seg = np.concatenate([upsample2(seg, p, f"{head}.upsample", cfg), x], axis=1)
    seg = cbr(seg, p, f"{P}.{i}", lk)
    seg = cbr(seg, p, f"{P}.{i + 1}", lk)
    seg = np.concatenate([upsample2(seg, p, f"{head}.upsample2", cfg), skip], axis=1)
    seg = cbr(seg, p, f"{P}.{i + 2}", lk)
    if taps is not None:
        taps[f"{head}.trunk"] = seg
    return seg, f"{P}.{i + 3}"


def seg_head_v2(x, skip, p, cfg, taps=None, head="seg_head"):
    """SegmentationHead / SegmentationHeadATT — segmentation.py:153-157 / :464-466: logits."""
    seg, last = seg_trunk(x, skip, p, cfg, taps, head)
    return conv_b(seg, p, last)


def seg_feat_head_v3(x, skip, p, cfg, taps=None):
    """SegmentationFeatHeadLight(/ATT) — segmentation.py:339-347 / :611-619.

    feat = featB(seg[:, :c_hidden//2]); seg_out = convs[-1](seg[:, -c_hidden//2:]).
    """
    seg, last = seg_trunk(x, skip, p, cfg, taps)
    split = cfg["channel_dims"][4] // 2
    feat = conv_b(seg[:, :split], p, "seg_head.featB")
    seg_out = conv_b(seg[:, -split:], p, last)
    if cfg.get("depth", False):
        return seg_out, feat, conv_b(seg[:, split:2 * split], p, "seg_head.featD")
    return seg_out, feat


def netvlad(x, p, prefix="vlad_head.netvlad", literal=False):
    """NetVLAD.forward — modules/aggregators/netvlad.py:79-106 (vladv2=False: 1x1 conv without bias).

    Restated as V = A X^T - rowsum(A) * centroids; ``literal=True``
    materialises the [K,C,S] residual tensor exactly as the reference does (small inputs only).
    """
    B, C = x.shape[:2]
    xn = l2_normalize(x, axis=1)
    w = p[f"{prefix}.conv.weight"]
    cent = p[f"{prefix}.centroids"].astype(x.dtype)
    K = w.shape[0]
    a = softmax(conv2d_1x1(xn, w).reshape(B, K, -1), axis=1)        # B,K,S
    xf = xn.reshape(B, C, -1)                                         # B,C,S
    if literal:
        resid = xf[:, None, :, :] - cent[None, :, :, None]           # B,K,C,S
        v = (resid * a[:, :, None, :]).sum(axis=-1)
    else:
        v = np.matmul(a, xf.transpose(0, 2, 1)) - a.sum(axis=2)[:, :, None] * cent[None]
    v = l2_normalize(v, axis=2)
    return l2_normalize(v.reshape(B, -1), axis=1)


def vpr_head(x, p, cfg, taps=None, only_encoder=False):
    """VPRHead.forward — modules/decoders/vpr.py:78-89."""
    lk = cfg["leaky_relu"]
    v = cbr(x, p, "vlad_head.convlad1", lk)
    v = cbr(v, p, "vlad_head.convlad2", lk)
    v = cbr(v, p, "vlad_head.convlad3", lk)
    if taps is not None:
        taps["vlad_head.enc"] = v
    if cfg.get("remove_netvlad", False):
        return v                                    # vpr.py:84: the encoder map itself, NCHW (whatever the pooler)
    if only_encoder:
        return l2_normalize(v, axis=1)              # vpr.py:85-86, L2Norm base.py:5-11
    method = cfg.get("global_descriptor_method", "netvlad")
    if method == "gem":
        return gem(v, p)
    if method == "convap":
        return convap(v, p)
    return netvlad(v, p)


def only_encoder(x, p, cfg):
    """KP2DTinyV2/V3.only_encoder — kp2dtiny.py:515-518 / :869-872."""
    xb, _ = backbone(x, p, cfg)
    return vpr_head(xb, p, cfg, only_encoder=True)


def netvlad_init_params(clsts, traindescs):
    """NetVLAD.init_params, vladv2=False — aggregators/netvlad.py:51-63.  Returns (alpha, centroids, conv weight)."""
    unit = clsts / np.linalg.norm(clsts, axis=1, keepdims=True)
    dots = np.dot(unit, traindescs.T)
    dots.sort(0)
    dots = dots[::-1, :]
    alpha = float(-np.log(0.01) / np.mean(dots[0, :] - dots[1, :]))
    return alpha, clsts, (alpha * unit)[:, :, None, None]


def gem(x, p, prefix="vlad_head.netvlad", unshuffle=4, eps=1e-6):
    """GeM.forward — modules/aggregators/gem.py:21-31: PixelUnshuffle(4), clamp(min=eps)^p, global mean, ^(1/p)."""
    B, C, H, W = x.shape
    r = unshuffle
    x = x.reshape(B, C, H // r, r, W // r, r).transpose(0, 1, 3, 5, 2, 4).reshape(B, C * r * r, H // r, W // r)
    pw = p[f"{prefix}.p"].astype(x.dtype)[0]
    return (np.maximum(x, x.dtype.type(eps)) ** pw).mean(axis=(2, 3)) ** (x.dtype.type(1) / pw)


def convap(x, p, prefix="vlad_head.netvlad", s1=4, s2=4):
    """ConvAP.forward — modules/aggregators/convap.py:28-34: 1x1 conv + bias, AdaptiveAvgPool2d((4,4)), L2."""
    x = conv2d_1x1(x, p[f"{prefix}.channel_pool.weight"], p[f"{prefix}.channel_pool.bias"])
    B, C, H, W = x.shape
    out = np.zeros((B, C, s1, s2), x.dtype)
    for a in range(s1):
        y0, y1 = (a * H) // s1, -((-(a + 1) * H) // s1)
        for b in range(s2):
            x0, x1 = (b * W) // s2, -((-(b + 1) * W) // s2)
            out[:, :, a, b] = x[:, :, y0:y1, x0:x1].mean(axis=(2, 3))
    return l2_normalize(out.reshape(B, -1), axis=1)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


# --------------------------------------------------------------------------
# model forward / post-processing
# --------------------------------------------------------------------------
def forward(x, p, cfg, taps=None, eval_mode=True):
    """KP2DTinyV2.forward (kp2dtiny.py:552-591) / KP2DTinyV3.forward (:906-957).

    Returns the same dict: score (sigmoid, un-bordered), coord (= raw tanh shift),
    feat (dense, H/2), vlad, seg (V2: logits; V3 in eval: Softmax2d probabilities).
    """
    lk = cfg["leaky_relu"]
    xb, skip = backbone(x, p, cfg, taps)
    if cfg["v3"]:
        sl = simple_task_head(xb, p, "score_loc_head", lk)
        score, shift = sigmoid(sl[:, 0:1]), np.tanh(sl[:, 1:3])
        r3 = seg_feat_head_v3(xb, skip, p, cfg, taps)
        seg, feat = r3[0], r3[1]
        if eval_mode:
            seg = softmax(seg, axis=1)
    else:
        score = sigmoid(simple_task_head(xb, p, "score_head", lk))
        shift = np.tanh(simple_task_head(xb, p, "loc_head", lk))
        feat = upscale_head(xb, skip, p, "desc_head", lk, cfg)
        seg = seg_head_v2(xb, skip, p, cfg, taps)
    vlad = vpr_head(xb, p, cfg, taps)
    out = {"score": score, "coord": shift, "feat": feat, "vlad": vlad, "seg": seg}
    if cfg.get("depth", False):   # kp2dtiny.py:588-590 (V2: second head) / :955-956 (V3: featD slice); both sigmoid
        out["depth"] = sigmoid(r3[2]) if cfg["v3"] else sigmoid(seg_head_v2(xb, skip, p, cfg, taps, "depth_head"))
    return out


def grid_sample_bilinear(feat, gx, gy):
    """F.grid_sample(bilinear, padding zeros, align_corners=True) — kp2dtiny.py:628.

    gx, gy: normalised coords [B,Ho,Wo] in [-1,1]; feat [B,C,Hi,Wi].
    """
    B, C, Hi, Wi = feat.shape
    dt = feat.dtype
    ix = (gx + dt.type(1)) / dt.type(2) * dt.type(Wi - 1)
    iy = (gy + dt.type(1)) / dt.type(2) * dt.type(Hi - 1)
    x0 = np.floor(ix)
    y0 = np.floor(iy)
    out = np.zeros((B, C) + gx.shape[1:], dt)
    bi = np.arange(B)[:, None, None]
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wx = (ix - x0) if dx else (x0 + dt.type(1) - ix)
            wy = (iy - y0) if dy else (y0 + dt.type(1) - iy)
            ok = (xi >= 0) & (xi <= Wi - 1) & (yi >= 0) & (yi <= Hi - 1)
            xc = np.clip(xi, 0, Wi - 1).astype(np.int64)
            yc = np.clip(yi, 0, Hi - 1).astype(np.int64)
            val = feat[bi, :, yc, xc]                      # B,Ho,Wo,C
            out += (val * (wx * wy * ok)[..., None]).transpose(0, 3, 1, 2)
    return out


def grid_sample_nearest(t, gx, gy):
    """F.grid_sample(mode="nearest", align_corners=True, zeros) — kp2dtiny.py:635-637 (sample_segmentation)."""
    B, C, Hi, Wi = t.shape
    dt = t.dtype
    ix = np.rint((gx + dt.type(1)) / dt.type(2) * dt.type(Wi - 1)).astype(np.int64)   # nearbyint: half to even
    iy = np.rint((gy + dt.type(1)) / dt.type(2) * dt.type(Hi - 1)).astype(np.int64)
    ok = (ix >= 0) & (ix < Wi) & (iy >= 0) & (iy < Hi)
    bi = np.arange(B)[:, None, None]
    val = t[bi, :, np.clip(iy, 0, Hi - 1), np.clip(ix, 0, Wi - 1)] * ok[..., None]
    return val.transpose(0, 3, 1, 2)


def post_processing(out, H, W, cfg, training=False, sample_segmentation=False):
    """KP2DTinyV2.post_processing (kp2dtiny.py:593-625) / V3 (:959-993).

    score * border mask (:520-528); coord = grid*cell + (cell-1)/2 + shift*cross_ratio*(cell-1)/2,
    clamped (:597-614, grid from utils/image.py:44-75); when ``training is False``: descriptors
    bilinearly sampled at the predicted coords and divided by their norm with NO eps (:627-631);
    seg -> argmax over classes, int64, at H/2 x W/2 (:633-640 / :1001-1008).
    """
    score, shift, feat = out["score"], out["coord"], out["feat"]
    dt = score.dtype
    B, _, Hc, Wc = score.shape
    mask = np.ones((Hc, Wc), dt)
    mask[0], mask[-1], mask[:, 0], mask[:, -1] = 0, 0, 0, 0
    score = score * mask
    cell = 2 ** cfg["downsample"]
    step = dt.type((cell - 1) / 2.0)
    xs = np.broadcast_to(np.arange(Wc, dtype=dt)[None, :], (Hc, Wc))
    ys = np.broadcast_to(np.arange(Hc, dtype=dt)[:, None], (Hc, Wc))
    base = np.stack([xs, ys])[None] * dt.type(cell) + step
    coord = base + shift * (dt.type(2.0) * step)
    coord = coord.copy()
    coord[:, 0] = np.clip(coord[:, 0], 0, W - 1)
    coord[:, 1] = np.clip(coord[:, 1], 0, H - 1)
    res = dict(out)
    if training is False:
        gx = coord[:, 0] / dt.type((W - 1) / 2.0) - dt.type(1)
        gy = coord[:, 1] / dt.type((H - 1) / 2.0) - dt.type(1)
        f = grid_sample_bilinear(feat, gx, gy)
        feat = f / np.sqrt((f * f).sum(axis=1, keepdims=True))
        seg = out["seg"]
        if sample_segmentation:
            seg = grid_sample_nearest(seg, gx, gy)
        res["seg"] = seg.argmax(axis=1)[:, None].astype(np.int64)
    res["feat"], res["coord"], res["score"] = feat, coord, score
    return res


# --------------------------------------------------------------------------
# keypoint selectors (callers of the path) — restated from source text; these
# modules need cv2/kornia and cannot be imported here.
# --------------------------------------------------------------------------
def select_k1(score, coord, feat, thr=0.7, top_k=4000):
    """VO selector — src/evaluation/visual_odometry.py:93-117, visual_odometry/frontend.py:94-127.

    B must be 1.  Returns (flat cell indices sorted ascending, pts[n,2], desc[n,C]).
    The reference uses np.argpartition (unordered); the kept SET is what is pinned.
    """
    assert score.shape[0] == 1
    s = score.reshape(-1)
    pts = coord.reshape(2, -1).T
    d = feat.reshape(feat.shape[1], -1).T
    idx = np.nonzero(s > thr)[0]
    if len(idx) > top_k > 0:
        order = np.lexsort((idx, -s[idx]))            # score desc, index asc
        idx = np.sort(idx[order[:top_k]])
    return idx, pts[idx], d[idx]


def select_k2(score, coord, feat, thr=0.7, k=1000):
    """Eval selector — evaluation/keypoints.py:113-128 + descriptor.py:12-36 (argsort asc, keep last k)."""
    return select_k1(score, coord, feat, thr, k)


def select_k3(score, coord, feat, k=1024):
    """gluefactory selector — gluefactory/models/extractors/kp2dtiny.py:38-42: batched torch.topk.

    Returns (idx[B,k] by score desc / index asc on ties, scores, pts[B,k,2], desc[B,k,C]).
    """
    B = score.shape[0]
    s = score.reshape(B, -1)
    n = s.shape[1]
    order = np.stack([np.lexsort((np.arange(n), -s[b])) for b in range(B)])[:, :k]
    pts = coord.reshape(B, 2, -1).transpose(0, 2, 1)
    d = feat.reshape(B, feat.shape[1], -1).transpose(0, 2, 1)
    bi = np.arange(B)[:, None]
    return order, s[bi, order], pts[bi, order], d[bi, order]


def bf_match_one_to_one(des1, des2, ratio=0.7):
    """BfFeatureMatcher.match — src/visual_odometry/feature_matcher.py:89-98 (cv2.BFMatcher(NORM_L2).knnMatch
    k=2, restated: L2 distance = sqrt(sum (a-b)^2), neighbours sorted by distance) + goodMatchesOneToOne
    :179-209 (skip if m.distance > ratio * n.distance; each trainIdx keeps the smallest distance, the earlier
    query on ties).  Returns ({trainIdx: (queryIdx, distance)}, nn_idx, nn_dist, nn_dist2)."""
    des1 = np.asarray(des1, np.float32)
    des2 = np.asarray(des2, np.float32)
    d = np.sqrt(((des1[:, None, :] - des2[None, :, :]) ** 2).sum(-1, dtype=np.float32))
    order = np.argsort(d, axis=1, kind="stable")
    nn = order[:, 0]
    d1 = d[np.arange(len(des1)), nn]
    d2 = d[np.arange(len(des1)), order[:, 1]] if des2.shape[0] > 1 else np.full(len(des1), np.inf, np.float32)
    best = {}
    for q in range(len(des1)):
        if d1[q] > np.float32(ratio) * d2[q]:
            continue
        t = int(nn[q])
        if t not in best or d1[q] < best[t][1]:
            best[t] = (q, float(d1[q]))
    return best, nn, d1, d2


def bf_match_semantic(des1, cls1, des2, cls2, ratio=0.7, n_classes=28):
    """VisualOdometry.match_semantic as written — src/visual_odometry/visual_odometry.py:347-380: for every class id the
    BF one-to-one match between the previous frame's keypoints of that class (query) and the current frame's (train),
    index lists concatenated.  A class with fewer than two train rows has no second neighbour: the reference's
    knnMatch(k=2) returns one-element lists, goodMatchesOneToOne raises on `for m, n in matches` and the bare except skips
    the class; an empty side skips it too.  (As shipped the loop unpacks two values from a three-value return, so its
    except fires for EVERY class: this restates what the loop is written to compute.)  Returns {trainIdx: (queryIdx, d)}
    over the full arrays."""
    des1 = np.asarray(des1, np.float32)
    des2 = np.asarray(des2, np.float32)
    cls1 = np.asarray(cls1).reshape(-1)
    cls2 = np.asarray(cls2).reshape(-1)
    out = {}
    for c in range(n_classes):
        i1 = np.where(cls1 == c)[0]
        i2 = np.where(cls2 == c)[0]
        if len(i1) == 0 or len(i2) < 2:
            continue
        best, *_ = bf_match_one_to_one(des1[i1], des2[i2], ratio)
        for t, (q, d) in best.items():
            out[int(i2[t])] = (int(i1[q]), d)
    return out


def bf_match_crosscheck(des1, des2):
    """cv2.BFMatcher(cv2.NORM_L2, crossCheck=True).match — src/evaluation/descriptor.py:221-222, restated: (q, t) is
    kept iff t is the nearest train row of q and q the nearest query of t (L2, lowest index on ties).
    Returns {trainIdx: (queryIdx, distance)}."""
    des1 = np.asarray(des1, np.float32)
    des2 = np.asarray(des2, np.float32)
    d = np.sqrt(((des1[:, None, :] - des2[None, :, :]) ** 2).sum(-1, dtype=np.float32))
    nn = np.argmin(d, axis=1)
    rnn = np.argmin(d, axis=0)
    return {int(t): (int(q), float(d[q, t])) for t, q in enumerate(rnn) if nn[q] == t}


def cast_params(p, dtype):
    return {k: (v.astype(dtype) if v.dtype.kind == "f" else v) for k, v in p.items()}
