"""KP2DTiny model API on the MI355X-native engine.

Drop-in host mirror of the reference's ``src/kp2dtiny/models/kp2dtiny.py``: the same names
(``KP2DTinyV2``, ``KP2DTinyV3``, ``tiny_factory``, ``get_config``, ``KP2DTINY_CONFIGS``,
``KP2DTINYV3_CONFIGS``), constructor keywords, attributes and ``state_dict`` keys/shapes
(SURVEY.md §8b, App. C), so ``demo.py`` / ``eval_multitask.py``-style callers keep working:

    model = tiny_factory("S_A", 28, v3=True); model.load_state_dict(sd); model.to("cuda")
    model.eval(); model.training = False
    out = model(x); out = model.post_processing(out, H, W)

The ``torch.nn`` sub-modules below only HOLD parameters (they give ``state_dict`` /
``load_state_dict`` / ``.to()`` / ``parameters()`` for free).  ``forward`` and ``post_processing``
never run a torch op on them: they hand raw device pointers to the C ABI of ``libkp2d_hip.so``
(include/kp2d.h).  A CPU tensor, a missing library or an unbuilt configuration raises — there is no
fallback path.
"""
from __future__ import annotations

import copy
import contextlib
import ctypes as C
import inspect
import weakref
import os

import torch
from torch import nn

from ... import _lib

# ---------------------------------------------------------------------------------------------
# configuration tables (reference: models/kp2dtiny.py:46-218).  Values are the reference's data.
# ---------------------------------------------------------------------------------------------
_S_DIMS = [16, 32, 32, 64, 64, 128]
_N_DIMS = [16, 24, 24, 48, 48, 96]
_D_DIMS = [64, 128, 128, 256, 256, 512]


def _cfg(dims, att, nfeat=32, down=2, **extra):
    d = {"nfeatures": nfeat, "channel_dims": list(dims), "downsample": down, "use_attention": att}
    d.update(extra)
    return d


KP2DTINY_CONFIGS = {
    "S": _cfg(_S_DIMS, False, leaky_relu=True, encoder_dim=64),
    "S_A": _cfg(_S_DIMS, True, leaky_relu=True, encoder_dim=64),
    "N": _cfg(_N_DIMS, False, leaky_relu=True, num_clusters=32, encoder_dim=48),
    "N_A": _cfg(_N_DIMS, True, leaky_relu=True, num_clusters=32, encoder_dim=48),
    "D": _cfg(_D_DIMS, True, nfeat=128, leaky_relu=True, encoder_dim=128, global_descriptor_method="convap"),
    "F": _cfg([16, 32, 64, 128, 128, 256], False, nfeat=64, down=3, leaky_relu=True),
    "GEM_N": _cfg(_N_DIMS, False, leaky_relu=True, num_clusters=32, encoder_dim=48, global_descriptor_method="gem"),
    "GEM_S_A": _cfg(_S_DIMS, True, leaky_relu=True, encoder_dim=64, global_descriptor_method="gem"),
    "CONVAP_S_A": _cfg(_S_DIMS, True, leaky_relu=True, encoder_dim=64, global_descriptor_method="convap"),
}

KP2DTINYV3_CONFIGS = {
    "S": _cfg(_S_DIMS, False, bn_momentum=0.1, leaky_relu=True, encoder_dim=64),
    "S_A": _cfg(_S_DIMS, True, bn_momentum=0.1, leaky_relu=True, encoder_dim=64),
    "N": _cfg(_N_DIMS, False, bn_momentum=0.1, encoder_dim=48),
    "N_A": _cfg(_N_DIMS, True, bn_momentum=0.1, encoder_dim=48),
    "D": _cfg(_D_DIMS, False, nfeat=128, leaky_relu=True, encoder_dim=128, global_descriptor_method="convap"),
    "D_A": _cfg(_D_DIMS, True, nfeat=128, leaky_relu=True, encoder_dim=128, global_descriptor_method="convap"),
    "CONVAP_S_A": _cfg(_S_DIMS, True, bn_momentum=0.1, leaky_relu=True, encoder_dim=64,
                       global_descriptor_method="convap"),
}


def get_config(config, to_mcu=False, to_export=False, v3=False):
    """Name -> constructor kwargs (reference: kp2dtiny.py:245-281).

    Unlike the reference (which mutates the module-level dict when ``to_mcu`` / ``to_export`` are set,
    SURVEY.md App. B.20) a copy is returned, so later calls are not contaminated.
    """
    table = KP2DTINYV3_CONFIGS if v3 else KP2DTINY_CONFIGS
    if config not in table:
        raise ValueError("Config {} not supported, choose from ".format(config), list(table.keys()))
    conf = copy.deepcopy(table[config])
    if to_mcu:
        conf["upscale_method"] = "convtranspose"
        conf["leaky_relu"] = False
    if to_export:
        conf["remove_netvlad"] = True
    return conf


def tiny_factory(config, n_classes, to_mcu=False, to_export=False, v3=False):
    """Build a model from a config name (reference: kp2dtiny.py:221-242)."""
    conf = get_config(config, to_mcu=to_mcu, to_export=to_export, v3=v3)
    cls = KP2DTinyV3 if v3 else KP2DTinyV2
    return cls(**conf, nClasses=n_classes)


# ---------------------------------------------------------------------------------------------
# parameter containers: same attribute tree as the reference => same state_dict keys
# ---------------------------------------------------------------------------------------------
# Weight-change tracking.  The engine re-uploads the state dict whenever a tensor of it changed (in-place writes bump
# ``_version``) or was replaced.  Rebuilding ``state_dict()`` on every call to see that cost ~185 us per call — as much
# as a whole single-frame step on the GPU — so the model keeps the list of its tensors and only walks the module tree
# again after something could have REPLACED a tensor: a parameter / buffer registration anywhere (global torch hooks),
# or an ``_apply`` (.to / .cuda / .float ...) on any holder.  Both bump this epoch.
# Supported ways to change weights, all seen on the next call: ``load_state_dict``, in-place writes (``p.add_()``,
# ``p.copy_()``, ``p.data.copy_()``), ``p.data = t`` (the signature includes every tensor's data pointer), ``setattr`` /
# ``register_parameter`` / ``register_buffer``, ``.to()`` / ``.float()``.  Writing into ``module._parameters[...]`` or
# ``module._buffers[...]`` behind torch's back is seen at the latest ``_SIG_RECHECK`` calls later (the cached tensor list
# is rebuilt from the module tree that often as a self-check).  The hooks below are process-global by torch's design;
# all they do is increment this counter.
_STRUCT_EPOCH = [0]
_SIG_RECHECK = 16


def _bump_epoch(*_a, **_k):
    _STRUCT_EPOCH[0] += 1


torch.nn.modules.module.register_module_parameter_registration_hook(_bump_epoch)
torch.nn.modules.module.register_module_buffer_registration_hook(_bump_epoch)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - guards against silent torch execution
        raise RuntimeError("parameter holder: the arithmetic runs in libkp2d_hip.so via the parent model")

    def _apply(self, fn, *a, **k):
        _bump_epoch()
        return super()._apply(fn, *a, **k)


class _CBR(_Holder):
    """Parameters of AnnotatedConvBnReLUModel (reference modules/base.py:14-46)."""

    def __init__(self, ci, co, bn_momentum=0.1):
        super().__init__()
        self.conv = nn.Conv2d(ci, co, 3, 1, 1, bias=False)
        self.bn = nn.BatchNorm2d(co, momentum=bn_momentum)


class _BackBone(_Holder):
    def __init__(self, c0, c1, c2, c3, c4, mom):
        super().__init__()
        self.conv1a, self.conv1b = _CBR(c0, c1, mom), _CBR(c1, c2, mom)
        self.conv2a, self.conv2b = _CBR(c2, c2, mom), _CBR(c2, c3, mom)
        self.conv3a, self.conv3b = _CBR(c3, c3, mom), _CBR(c3, c4, mom)
        self.conv4a, self.conv4b = _CBR(c4, c4, mom), _CBR(c4, c4, mom)


class _SimpleTaskHead(_Holder):
    def __init__(self, ci, ch, co, mom):
        super().__init__()
        self.convDa = _CBR(ci, ch, mom)
        self.convDb = nn.Conv2d(ch, co, 3, 1, 1)


class _TConvUp(_Holder):
    """Parameters of TransposedConvUpsampleModel (reference modules/base.py:80-117), the ``to_mcu`` upsampler."""

    def __init__(self, c):
        super().__init__()
        self.transposed_conv = nn.ConvTranspose2d(c, c // 4, 3, stride=2, padding=1, output_padding=1, bias=False)
        self.bn = nn.BatchNorm2d(c // 4, momentum=0.1)


def _check_upscale(method):
    if method not in _lib.UPSCALE_METHODS:
        raise NotImplementedError("Upscale method not implemented")      # heads.py:58 / segmentation.py:120


class _UpscaleHead(_Holder):
    def __init__(self, c0, c1, c2, c3, c4, c5, mom, upscale_method="pixelshuffle"):
        super().__init__()
        _check_upscale(upscale_method)
        if upscale_method == "convtranspose":
            self.upsample = _TConvUp(c2)       # registered before the convolutions (heads.py:53-58)
        self.convA = _CBR(c0, c1, mom)
        self.convB = nn.Conv2d(c1, c2, 3, 1, 1)
        self.confAa = _CBR(c3, c4, mom)
        self.confBb = nn.Conv2d(c4, c5, 3, 1, 1)


class _ChannelLayerNorm(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))
        self.b = nn.Parameter(torch.zeros(1, dim, 1, 1))


class _PreNorm(_Holder):
    def __init__(self, dim, fn):
        super().__init__()
        self.fn = fn
        self.norm = _ChannelLayerNorm(dim)


class _ESA(_Holder):
    def __init__(self, dim, reduction_ratio=2):
        super().__init__()
        self.to_q = nn.Conv2d(dim, dim, 1, bias=False)
        self.to_kv = nn.Conv2d(dim, dim * 2, reduction_ratio, stride=reduction_ratio, bias=False)
        self.to_out = nn.Conv2d(dim, dim, 1, bias=False)


class _DsConv(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.Sequential(nn.Conv2d(dim, dim, 3, padding=1, groups=dim), nn.Conv2d(dim, dim, 1))


class _MixFFN(_Holder):
    def __init__(self, dim, expansion=2):
        super().__init__()
        h = dim * expansion
        self.net = nn.Sequential(nn.Conv2d(dim, h, 1), _DsConv(h), nn.GELU(), nn.Conv2d(h, dim, 1))


class _AttentionModule(_Holder):
    def __init__(self, c):
        super().__init__()
        self.att = _PreNorm(c, _ESA(c))
        self.mff = _PreNorm(c, _MixFFN(c))


class _SegHead(_Holder):
    """Parameters of the four segmentation heads (reference modules/decoders/segmentation.py)."""

    def __init__(self, c_in, c_hidden, c_exp, c_out, d1, mom, attention, n_feat=None, depth=False,
                 upscale_method="pixelshuffle"):
        super().__init__()
        _check_upscale(upscale_method)
        fused = n_feat is not None          # V3 "decoder fusion": feat + seg from one trunk
        last_in = c_hidden // 2 if fused else c_hidden
        c_hidden_b = c_hidden + c_hidden // 2 if (fused and depth) else c_hidden   # segmentation.py:190-193
        if attention:
            layers = [_CBR(c_in, c_hidden, mom), _AttentionModule(c_hidden), _AttentionModule(c_hidden),
                      _CBR(c_hidden, d1, mom), _CBR(c_hidden + d1 // 4, c_hidden, mom), _CBR(c_hidden, d1, mom),
                      _CBR(c_exp, c_hidden_b, mom), nn.Conv2d(last_in, c_out, 3, 1, 1)]
        else:
            layers = [_CBR(c_in, c_hidden, mom), _CBR(c_hidden, c_hidden, mom), _CBR(c_hidden, c_hidden, mom),
                      _CBR(c_hidden, c_hidden, mom), _CBR(c_hidden, d1, mom),
                      _CBR(c_hidden + d1 // 4, c_hidden, mom), _CBR(c_hidden, d1, mom), _CBR(c_exp, c_hidden_b, mom),
                      nn.Conv2d(last_in, c_out, 3, 1, 1)]
        self.convs = nn.ModuleList(layers)
        if fused:
            self.featB = nn.Conv2d(c_hidden // 2, n_feat, 3, 1, 1)
            if depth:
                self.featD = nn.Conv2d(c_hidden // 2, 1, 3, 1, 1, bias=False)
        if upscale_method == "convtranspose":       # segmentation.py:116-118 / :295-297 / :432-434 / :569-571
            self.upsample, self.upsample2 = _TConvUp(d1), _TConvUp(d1)

    def freeze(self, except_last_layer=False):
        for p in self.parameters():
            p.requires_grad = False
        if except_last_layer:
            for p in self.convs[-1].parameters():
                p.requires_grad = True


class _NetVLAD(_Holder):
    def __init__(self, num_clusters, dim):
        super().__init__()
        self.num_clusters, self.dim = num_clusters, dim
        self.alpha = 0
        self.conv = nn.Conv2d(dim, num_clusters, kernel_size=(1, 1), bias=False)
        self.centroids = nn.Parameter(torch.rand(num_clusters, dim))

    def init_params(self, clsts, traindescs):
        """NetVLAD.init_params, vladv2=False (reference aggregators/netvlad.py:41-56): soft-assignment weights are the
        unit-norm cluster centres scaled by alpha, alpha chosen so the runner-up cluster gets weight 0.01."""
        import numpy as np
        unit = clsts / np.linalg.norm(clsts, axis=1, keepdims=True)
        dots = np.sort(np.dot(unit, traindescs.T), axis=0)[::-1, :]          # per descriptor, descending
        self.alpha = (-np.log(0.01) / np.mean(dots[0, :] - dots[1, :])).item()
        dev = self.centroids.device
        self.centroids = nn.Parameter(torch.from_numpy(np.ascontiguousarray(clsts)).to(dev))
        self.conv.weight = nn.Parameter(torch.from_numpy(self.alpha * unit).unsqueeze(2).unsqueeze(3).to(dev))
        self.conv.bias = None


class _GeM(_Holder):
    """Parameters of GeM (reference modules/aggregators/gem.py:7-19)."""

    def __init__(self, p=3.0):
        super().__init__()
        self.p = nn.Parameter(torch.ones(1) * p)


class _ConvAP(_Holder):
    """Parameters of ConvAP (reference modules/aggregators/convap.py:19-26)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.channel_pool = nn.Conv2d(c_in, c_out, kernel_size=1, bias=True)


class _VPRHead(_Holder):
    """Parameters of VPRHead (reference modules/decoders/vpr.py:8-76): pooler chosen by ``method``."""

    def __init__(self, c_in, enc, num_clusters, mom, method="netvlad", remove_netvlad=False):
        super().__init__()
        self.convlad1, self.convlad2, self.convlad3 = _CBR(c_in, enc, mom), _CBR(enc, enc, mom), _CBR(enc, enc, mom)
        if method == "netvlad":
            if not remove_netvlad:
                self.netvlad = _NetVLAD(num_clusters, enc)     # NetVLADMemoryEfficient computes the same function
                self.global_desc_dim = num_clusters * enc
            else:
                self.global_desc_dim = 0
        elif method == "gem":
            self.netvlad = _GeM()
            self.global_desc_dim = enc * 16
        elif method == "convap":
            self.netvlad = _ConvAP(enc, enc)
            self.global_desc_dim = enc * 16
        else:
            raise ValueError(f"unknown global_descriptor_method {method!r}")


# ---------------------------------------------------------------------------------------------
# engine handle
# ---------------------------------------------------------------------------------------------
class _Engine:
    """One kp2d_model handle bound to a device; owns the cached workspace."""

    def __init__(self, cfg: _lib.Kp2dConfig):
        self.lib = _lib.load()
        self.handle = C.c_void_p()
        _lib.check(self.lib.kp2d_create(C.byref(cfg), C.byref(self.handle)))
        self.device = cfg.device
        self._ws = None
        self._ws_call = None               # a caller's own workspace for the forwards inside using_workspace()
        self.signature = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.kp2d_destroy(self.handle)
        except Exception:
            pass

    def expected(self):
        n = self.lib.kp2d_num_weights(self.handle)
        out = []
        key, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        for i in range(n):
            _lib.check(self.lib.kp2d_weight_info(self.handle, i, C.byref(key), shape, C.byref(nd)))
            out.append((key.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    def upload(self, state_dict):
        for key, shape in self.expected():
            t = state_dict[key].detach().to("cpu", torch.float32).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError(f"{key}: shape {tuple(t.shape)} != expected {shape}")
            sh = (C.c_int64 * max(1, t.dim()))(*t.shape)
            _lib.check(self.lib.kp2d_set_weight(self.handle, key.encode(), C.c_void_p(t.data_ptr()), sh, t.dim()))
        _lib.check(self.lib.kp2d_finalize_weights(self.handle))

    def workspace(self, B, H, W, device):
        need = self.lib.kp2d_workspace_bytes(self.handle, B, H, W)
        if need == 0:
            _lib.check(-1 if not self.lib.kp2d_last_error() else -5)
        if self._ws_call is not None:
            if self._ws_call.numel() < need or self._ws_call.device != device:
                raise RuntimeError(f"using_workspace(): the buffer holds {self._ws_call.numel()} bytes on {self._ws_call.device}, "
                                   f"the forward needs {need} on {device}")
            return self._ws_call
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    @contextlib.contextmanager
    def using_workspace(self, ws):
        """Forwards enqueued inside the block take `ws` (a uint8 device tensor of at least kp2d_workspace_bytes) instead of
        the engine's cached workspace, which stays untouched: streams that keep several forwards in flight (pipeline.
        FrameStream / BatchStream) give every slot its own buffer this way, and a plain ``net(x)`` beside them keeps the
        engine's — it can never land on a buffer a slot's kernels are still using."""
        prev, self._ws_call = self._ws_call, ws
        try:
            yield
        finally:
            self._ws_call = prev


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()


class _KP2DTinyBase(nn.Module):
    """Shared host logic of KP2DTinyV2 / KP2DTinyV3."""

    _version_id = 0

    # ---- engine plumbing ----------------------------------------------------------------------
    def _engine_config(self, device_index: int) -> _lib.Kp2dConfig:
        cfg = _lib.Kp2dConfig()
        cfg.struct_size = C.sizeof(_lib.Kp2dConfig)
        cfg.version = self._version_id
        for i, v in enumerate(self.channel_dims):
            cfg.channel_dims[i] = int(v)
        cfg.nfeatures, cfg.n_classes = int(self.nfeatures), int(self.nClasses)
        cfg.num_clusters, cfg.encoder_dim = int(self.num_clusters), int(self.encoder_dim)
        cfg.downsample = int(self.downsample)
        cfg.use_attention, cfg.leaky_relu = int(bool(self.use_attention)), int(bool(self.leaky_relu))
        cfg.remove_softmax = int(bool(getattr(self, "remove_softmax", False)))
        cfg.device = device_index
        cfg.global_descriptor = _lib.GLOBAL_DESCRIPTORS[self.global_descriptor_method]
        cfg.remove_netvlad = int(bool(self.remove_netvlad))
        cfg.depth = int(bool(self.depth))
        cfg.upscale_method = _lib.UPSCALE_METHODS[self.upscale_method]
        cfg.in_channels = 3 if getattr(self, "use_color", True) else 1     # KP2DTinyV3(use_color=False): kp2dtiny.py:718-721
        return cfg

    def _check_built(self):
        pass

    def _apply(self, fn, *a, **k):
        _bump_epoch()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        _bump_epoch()                      # assign=True replaces tensors without any registration hook
        return super().load_state_dict(*a, **k)

    def _weights_signature(self):
        cache = self.__dict__.get("_sig_cache")
        n = self.__dict__.get("_sig_calls", 0) + 1
        self.__dict__["_sig_calls"] = n
        if cache is None or cache[0] != _STRUCT_EPOCH[0] or n % _SIG_RECHECK == 0:
            cache = (_STRUCT_EPOCH[0], list(self.state_dict(keep_vars=True).values()))
            self.__dict__["_sig_cache"] = cache
        return tuple((id(t), t._version, t.data_ptr()) for t in cache[1])

    def _warn_if_training_semantics_expected(self):
        """The reference runs BatchNorm on batch statistics and applies Dropout2d while its sub-modules are in training
        mode (a caller that skips ``model.eval()``); this build implements inference only (running statistics, no
        dropout), so say so once instead of silently returning eval-mode numbers."""
        if self.backbone.training and not self.__dict__.get("_warned_train"):
            import warnings
            warnings.warn("KP2DTiny (MI355X build) implements inference semantics only: BatchNorm uses its running "
                          "statistics and Dropout2d is inactive even though the sub-modules are in training mode. "
                          "Call model.eval() (the reference does: demo.py, eval_multitask.py).", RuntimeWarning,
                          stacklevel=3)
            self.__dict__["_warned_train"] = True

    def _get_engine(self, device: torch.device, need_weights: bool = True) -> _Engine:
        if device.type != "cuda":
            raise RuntimeError(
                "KP2DTiny (MI355X build) runs on a HIP device only: move the model and the input to 'cuda'. "
                "There is no CPU path in this package.")
        self._check_built()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        eng = self.__dict__.get("_engine")
        if eng is None or eng.device != idx:
            eng = _Engine(self._engine_config(idx))
            self.__dict__["_engine"] = eng
        if not need_weights:               # post_processing: kp2d_post reads the configuration only
            return eng
        sig = self._weights_signature()
        if eng.signature != sig:
            eng.upload(self.state_dict())
            eng.signature = sig
        self._apply_precision(eng)
        return eng

    def _apply_precision(self, eng):
        mode = _lib.PRECISIONS.get(self.__dict__.get("_precision") or os.environ.get("KP2D_PRECISION", "f16x3"))
        if mode is None:
            raise ValueError("precision must be 'fp32' or 'f16x3'")
        if eng.lib.kp2d_get_precision(eng.handle) != mode:
            _lib.check(eng.lib.kp2d_set_precision(eng.handle, mode))

    def set_precision(self, name: str):
        """'f16x3' (default): split-fp16 matrix-core products with fp32 accumulation, fp32-grade error;
        'fp32': exact fp32 matrix-core products.  Extension of this build (the reference has one arithmetic)."""
        if name not in _lib.PRECISIONS:
            raise ValueError("precision must be 'fp32' or 'f16x3'")
        self.__dict__["_precision"] = name
        return self

    # ---- packed-weight exchange for multi-GPU frame sharding (sharding.py) ----------------------
    def packed_weights(self, device) -> torch.Tensor:
        eng = self._get_engine(torch.device(device))
        n = eng.lib.kp2d_packed_bytes(eng.handle)
        buf = torch.empty(n, dtype=torch.uint8, device=device)
        s = torch.cuda.current_stream(buf.device)
        _lib.check(eng.lib.kp2d_export_packed(eng.handle, _ptr(buf), C.c_void_p(s.cuda_stream)))
        return buf

    def load_packed_weights(self, buf: torch.Tensor):
        dev = buf.device
        if dev.type != "cuda":
            raise RuntimeError("packed weights must live on the HIP device")
        self._check_built()
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        eng = self.__dict__.get("_engine")
        if eng is None or eng.device != idx:
            eng = _Engine(self._engine_config(idx))
            self.__dict__["_engine"] = eng
        if buf.numel() != eng.lib.kp2d_packed_bytes(eng.handle):
            raise ValueError("packed weight blob has the wrong size for this configuration")
        s = torch.cuda.current_stream(dev)
        _lib.check(eng.lib.kp2d_import_packed(eng.handle, _ptr(buf), C.c_void_p(s.cuda_stream)))
        eng.signature = self._weights_signature()
        self._apply_precision(eng)

    # ---- reference API ------------------------------------------------------------------------
    def forward(self, x):
        """Reference: KP2DTinyV2.forward kp2dtiny.py:552-591 / KP2DTinyV3.forward :906-957."""
        c0 = 3 if getattr(self, "use_color", True) else 1
        if x.dim() != 4 or x.shape[1] != c0:
            raise ValueError(f"expected [B,{c0},H,W] input, got {tuple(x.shape)}")
        if x.dtype != torch.float32:
            raise TypeError("input must be float32")
        x = x.contiguous()
        B, _, H, W = x.shape
        return self._forward(x.device, B, H, W, x=x)

    def forward_frames(self, frames, size=None):
        """forward() straight from uint8 frames [B,Hs,Ws,3] on the device: /255, the bilinear resize to ``size`` = (H, W)
        and .sub(0.5).mul(2) (src/evaluation/visual_odometry.py:77-87) run as the first layer's prologue
        (kp2d_forward_frames): the float input tensor is never materialised.  Bit-identical to
        forward(pipeline.frames_to_input(frames, device, size))."""
        if frames.dim() != 4 or frames.shape[-1] != 3 or frames.dtype != torch.uint8:
            raise ValueError("expected uint8 frames of shape [B,Hs,Ws,3]")
        if frames.device.type != "cuda":
            raise RuntimeError("the frame front-end runs on the HIP device only")
        frames = frames.contiguous()
        B, Hs, Ws, _ = frames.shape
        H, W = (Hs, Ws) if size is None else (int(size[0]), int(size[1]))
        return self._forward(frames.device, B, H, W, frames=frames)

    def _forward(self, dev, B, H, W, x=None, frames=None):
        eng = self._get_engine(dev)
        self._warn_if_training_semantics_expected()
        q = 2 * self.cell    # the segmentation head pools the cell grid once more
        if H % q or W % q:
            raise ValueError(f"H and W must be divisible by {q} (got {H}x{W}); reference README.md:143")
        Hc, Wc = H // self.cell, W // self.cell
        H2, W2 = 2 * Hc, 2 * Wc        # dense maps sit one pixel-shuffle above the cell grid
        score = torch.empty(B, 1, Hc, Wc, device=dev)
        shift = torch.empty(B, 2, Hc, Wc, device=dev)
        feat = torch.empty(B, self.nfeatures, H2, W2, device=dev)
        seg = torch.empty(B, self.nClasses, H2, W2, device=dev)
        vdim = eng.lib.kp2d_vlad_dim(eng.handle, H, W)
        vlad = (torch.empty(B, self.encoder_dim, Hc, Wc, device=dev) if self.remove_netvlad
                else torch.empty(B, vdim, device=dev))
        depth = torch.empty(B, 1, H2, W2, device=dev) if self.depth else None
        ws = eng.workspace(B, H, W, dev)
        flags = 0 if self.training else _lib.KP2D_FWD_EVAL
        stream = torch.cuda.current_stream(dev).cuda_stream
        # inference mode: the layer that writes `seg` also writes the dense class map post_processing starts with (its
        # tile of logits is in LDS anyway), so post_processing need not read the logits again — if it gets THIS tensor,
        # unmodified (see post_processing)
        ids = None
        if not self.training and not self.sample_segmentation and os.environ.get("KP2D_FUSED_ARGMAX", "1") != "0":
            ids = torch.empty(B, 1, H2, W2, dtype=torch.int64, device=dev)
            _lib.check(eng.lib.kp2d_set_seg_ids(eng.handle, _ptr(ids), ids.numel()))
        try:
            self._run_forward(eng, x, frames, B, H, W, flags, score, shift, feat, seg, vlad, depth, ws, stream)
        finally:
            if ids is not None:
                eng.lib.kp2d_set_seg_ids(eng.handle, None, 0)
        self.__dict__["_seg_ids_cache"] = (weakref.ref(seg), seg._version, ids) if ids is not None else None
        out = {"score": score, "coord": shift, "feat": feat, "vlad": vlad, "seg": seg}
        if self.depth:
            out["depth"] = depth        # already sigmoid (kp2dtiny.py:588-590 / :955-956)
        return out

    def _run_forward(self, eng, x, frames, B, H, W, flags, score, shift, feat, seg, vlad, depth, ws, stream):
        if frames is None:
            _lib.check(eng.lib.kp2d_forward(eng.handle, _ptr(x), B, H, W, flags, _ptr(score), _ptr(shift), _ptr(feat),
                                            _ptr(seg), _ptr(vlad), _ptr(depth), _ptr(ws), ws.numel(), C.c_void_p(stream)))
        else:
            _lib.check(eng.lib.kp2d_forward_frames(eng.handle, _ptr(frames), B, frames.shape[1], frames.shape[2], H, W, flags,
                                                   _ptr(score), _ptr(shift), _ptr(feat), _ptr(seg), _ptr(vlad), _ptr(depth),
                                                   _ptr(ws), ws.numel(), C.c_void_p(stream)))

    def post_processing(self, out, H, W):
        """Reference: post_processing kp2dtiny.py:593-625 / :959-993 (mutates and returns ``out``)."""
        score, shift, feat = out["score"], out["coord"], out["feat"]
        if score.device.type != "cuda":
            raise RuntimeError("post_processing runs on the HIP device only")
        eng = self._get_engine(score.device, need_weights=False)
        score, shift, feat = score.contiguous(), shift.contiguous(), feat.contiguous()
        B, _, Hc, Wc = score.shape
        dev = score.device
        sample = self.training is False
        score_out = torch.empty_like(score)
        coord = torch.empty_like(shift)
        desc = seg_ids = seg = None
        fc, Hf, Wf = feat.shape[1], feat.shape[2], feat.shape[3]
        sc, Hs, Ws = 0, 0, 0
        if sample:
            seg = out["seg"].contiguous()
            sc, Hs, Ws = seg.shape[1], seg.shape[2], seg.shape[3]
            desc = torch.empty(B, fc, Hc, Wc, device=dev)
            # class ids the forward already wrote (kp2d_set_seg_ids) are taken over when `seg` is the very tensor that
            # forward returned, untouched since (same object, same version counter); any other dict goes the full way
            cache = self.__dict__.pop("_seg_ids_cache", None)
            if (cache is not None and not self.sample_segmentation and cache[0]() is out["seg"] and seg is out["seg"]
                    and seg._version == cache[1] and tuple(cache[2].shape) == (B, 1, Hs, Ws)):
                seg_ids, seg = cache[2], None
            else:
                seg_ids = (torch.empty(B, 1, Hc, Wc, dtype=torch.int64, device=dev) if self.sample_segmentation
                           else torch.empty(B, 1, Hs, Ws, dtype=torch.int64, device=dev))
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(eng.lib.kp2d_post(eng.handle, _ptr(score), _ptr(shift), _ptr(feat), _ptr(seg), B, int(H), int(W),
                                     Hc, Wc, fc, Hf, Wf, sc, Hs, Ws, _ptr(score_out), _ptr(coord), _ptr(desc),
                                     _ptr(seg_ids), int(bool(self.sample_segmentation)), C.c_void_p(stream)))
        if sample:
            out["seg"] = seg_ids
            feat = desc
        out["feat"] = feat
        out["coord"] = coord
        out["score"] = score_out
        return out

    # ---- small reference helpers kept verbatim in meaning --------------------------------------
    def gather_info(model):
        params = inspect.signature(model.__init__).parameters
        info = {
            "init_args": {n: getattr(model, n) for n in params if hasattr(model, n)},
            "total_params": sum(p.numel() for p in model.parameters()),
            "trainable_params": sum(p.numel() for p in model.parameters() if p.requires_grad),
            "netvlad_dim": model.global_desc_dim,
        }
        if model._version_id == 2:
            info.update(upscale_method=model.upscale_method, leaky_relu=model.leaky_relu,
                        use_attention=model.use_attention)
        return info

    def get_global_desc_dim(self):
        return self.global_desc_dim

    def get_netvlad_dim(self):
        return self.global_desc_dim

    def get_num_clusters(self):
        return self.vlad_head.netvlad.num_clusters

    def freeze_backbone(self):
        for p in self.backbone.parameters():
            p.requires_grad = False

    def freeze_segmentation(self, except_last_layer=False):
        self.seg_head.freeze(except_last_layer)

    def fuse(self):
        """BatchNorm is already folded into per-channel scale/shift when weights are packed."""
        return None

    def init_netvlad(self, clsts, traindescs):
        """Reference: kp2dtiny.py:490-491 -> NetVLAD.init_params (aggregators/netvlad.py:41-77, vladv2=False branch)."""
        self.vlad_head.netvlad.init_params(clsts, traindescs)

    def forward_with_tap(self, x, layer: str, shape):
        """forward(x) plus ONE intermediate activation as a planar [B, *shape] tensor (kp2d_set_tap, include/kp2d.h):
        how the tests compare the kernels with the reference's recorded intermediates layer by layer."""
        eng = self._get_engine(x.device)
        tap = torch.full((x.shape[0],) + tuple(int(v) for v in shape), float("nan"), device=x.device)
        _lib.check(eng.lib.kp2d_set_tap(eng.handle, layer.encode(), _ptr(tap), tap.numel()))
        try:
            out = self.forward(x)
        finally:
            _lib.check(eng.lib.kp2d_set_tap(eng.handle, None, None, 0))
        return out, tap

    def only_encoder(self, x):
        """Reference: kp2dtiny.py:515-518 — backbone + VPR encoder, channel-wise L2-normalised (vpr.py:84-87)."""
        c0 = 3 if getattr(self, "use_color", True) else 1
        if x.dim() != 4 or x.shape[1] != c0 or x.dtype != torch.float32:
            raise ValueError(f"expected float32 [B,{c0},H,W] input, got {x.dtype} {tuple(x.shape)}")
        eng = self._get_engine(x.device)
        self._warn_if_training_semantics_expected()
        x = x.contiguous()
        B, _, H, W = x.shape
        q = 2 * self.cell
        if H % q or W % q:
            raise ValueError(f"H and W must be divisible by {q} (got {H}x{W})")
        enc = torch.empty(B, self.encoder_dim, H // self.cell, W // self.cell, device=x.device)
        ws = eng.workspace(B, H, W, x.device)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        null = C.c_void_p()
        _lib.check(eng.lib.kp2d_forward(eng.handle, _ptr(x), B, H, W, _lib.KP2D_FWD_ONLY_ENCODER, null, null, null, null,
                                        _ptr(enc), null, _ptr(ws), ws.numel(), C.c_void_p(stream)))
        return enc


def _common_init(self, *, nfeatures, device, channel_dims, bn_momentum, nClasses, num_clusters, downsample,
                 use_attention, mem_efficient, upscale_method, remove_netvlad, leaky_relu, depth, encoder_dim,
                 global_descriptor_method):
    self.device = device
    self.with_drop = True
    self.nfeatures, self.downsample, self.nClasses = nfeatures, downsample, nClasses
    self.sample_segmentation = False
    self.use_attention, self.leaky_relu = use_attention, leaky_relu
    self.remove_netvlad, self.upscale_method, self.depth = remove_netvlad, upscale_method, depth
    self.num_clusters, self.global_descriptor_method = num_clusters, global_descriptor_method
    self.mem_efficient = mem_efficient
    self.bn_momentum = bn_momentum
    self.cross_ratio = 2.0
    self.channel_dims = list(channel_dims)
    self.encoder_dim = encoder_dim if encoder_dim is not None else channel_dims[3]


class KP2DTinyV2(_KP2DTinyBase):
    """Reference: class KP2DTinyV2, kp2dtiny.py:284-647."""

    _version_id = 2

    def __init__(self, nfeatures=256, device="cpu", channel_dims=[32, 64, 128, 256, 256, 512], bn_momentum=0.1,
                 nClasses=8, num_clusters=64, downsample=3, use_attention=False, mem_efficient=False,
                 upscale_method="pixelshuffle", remove_netvlad=False, leaky_relu=True, depth=False, encoder_dim=None,
                 global_descriptor_method="netvlad", **kwargs):
        super().__init__()
        _common_init(self, nfeatures=nfeatures, device=device, channel_dims=channel_dims, bn_momentum=bn_momentum,
                     nClasses=nClasses, num_clusters=num_clusters, downsample=downsample,
                     use_attention=use_attention, mem_efficient=mem_efficient, upscale_method=upscale_method,
                     remove_netvlad=remove_netvlad, leaky_relu=leaky_relu, depth=depth, encoder_dim=encoder_dim,
                     global_descriptor_method=global_descriptor_method)
        c1, c2, c3, c4, c5, d1 = channel_dims
        mom = bn_momentum
        self.backbone = _BackBone(3, c1, c2, c3, c4, mom)
        self.score_head = _SimpleTaskHead(c4, c4, 1, mom)
        self.loc_head = _SimpleTaskHead(c4, c4, 2, mom)
        self.desc_head = _UpscaleHead(c4, c4, c3 * 4, c3 + c4, c4, nfeatures, mom, upscale_method)
        self.seg_head = _SegHead(c4, c5, c4 + c3, nClasses, d1, mom, use_attention, upscale_method=upscale_method)
        if depth:
            self.depth_head = _SegHead(c4, c5, c4 + c3, 1, d1, mom, use_attention, upscale_method=upscale_method)
        self.vlad_head = _VPRHead(c4, self.encoder_dim, num_clusters, mom, global_descriptor_method, remove_netvlad)
        self.cell = pow(2, self.downsample)
        self.training = True                       # reference force-sets this (kp2dtiny.py:456)
        self.global_desc_dim = self.vlad_head.global_desc_dim


class KP2DTinyV3(_KP2DTinyBase):
    """Reference: class KP2DTinyV3, kp2dtiny.py:650-1015 (fused score/loc head, fused seg+descriptor head)."""

    _version_id = 3

    def __init__(self, use_color=True, do_cross=True, with_drop=True, nfeatures=256, device="cpu",
                 channel_dims=[32, 64, 128, 256, 256, 512], bn_momentum=0.1, nClasses=8, num_clusters=64,
                 downsample=3, use_attention=False, encoder_dim=None, mem_efficient=False,
                 upscale_method="pixelshuffle", remove_netvlad=False, leaky_relu=True, remove_softmax=False,
                 depth=False, global_descriptor_method="netvlad", **kwargs):
        super().__init__()
        _common_init(self, nfeatures=nfeatures, device=device, channel_dims=channel_dims, bn_momentum=bn_momentum,
                     nClasses=nClasses, num_clusters=num_clusters, downsample=downsample,
                     use_attention=use_attention, mem_efficient=mem_efficient, upscale_method=upscale_method,
                     remove_netvlad=remove_netvlad, leaky_relu=leaky_relu, depth=depth, encoder_dim=encoder_dim,
                     global_descriptor_method=global_descriptor_method)
        self.with_drop, self.use_color, self.do_cross = with_drop, use_color, do_cross
        self.fuse_score_loc = True
        self.remove_softmax = remove_softmax
        c1, c2, c3, c4, c5, d1 = channel_dims
        mom = bn_momentum
        self.backbone = _BackBone(3 if use_color else 1, c1, c2, c3, c4, 0.1)
        self.score_loc_head = _SimpleTaskHead(c4, c4, 3, mom)
        self.seg_head = _SegHead(c4, c5, c4 + c3, nClasses, d1, mom, use_attention, n_feat=nfeatures, depth=depth,
                                 upscale_method=upscale_method)
        self.vlad_head = _VPRHead(c4, self.encoder_dim, num_clusters, mom, global_descriptor_method, remove_netvlad)
        self.cell = pow(2, self.downsample)
        self.training = True                       # kp2dtiny.py:813
        self.global_desc_dim = self.vlad_head.global_desc_dim
