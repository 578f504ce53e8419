"""Host-side mirror of the reference's LightGlue matcher (lightglue/lightglue.py:418-614), inference path.

Same constructor (``LightGlue(conf, weights_path=None)``), same ``state_dict`` keys and shapes, same ``forward(data)``
contract (``required_data_keys`` + ``view0/view1["image_size"]``) and the same prediction dict; the arithmetic runs in
``libkp2d_hip.so`` through include/kp2d_lightglue.h — the modules below only hold parameters.  Not built: training
(loss, checkpointing), early stopping (``depth_confidence``) and point pruning (``width_confidence``) — the reference
configs leave both at -1 — ``add_scale_ori`` and ``flash`` / mixed precision (the kernels are exact fp32).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch
from torch import nn

from .. import _lib


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the arithmetic runs in libkp2d_hip.so via LightGlue.forward")


def _ffn(d):
    return nn.Sequential(nn.Linear(2 * d, 2 * d), nn.LayerNorm(2 * d, elementwise_affine=True), nn.GELU(),
                         nn.Linear(2 * d, d))


class _PosEnc(_Holder):
    def __init__(self, M, dim, F_dim):
        super().__init__()
        self.Wr = nn.Linear(M, F_dim // 2, bias=False)


class _SelfBlock(_Holder):
    def __init__(self, d):
        super().__init__()
        self.Wqkv = nn.Linear(d, 3 * d, bias=True)
        self.out_proj = nn.Linear(d, d, bias=True)
        self.ffn = _ffn(d)


class _CrossBlock(_Holder):
    def __init__(self, d):
        super().__init__()
        self.to_qk, self.to_v, self.to_out = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.ffn = _ffn(d)


class _TransformerLayer(_Holder):
    def __init__(self, d):
        super().__init__()
        self.self_attn = _SelfBlock(d)
        self.cross_attn = _CrossBlock(d)


class _MatchAssignment(_Holder):
    def __init__(self, d):
        super().__init__()
        self.dim = d
        self.matchability = nn.Linear(d, 1, bias=True)
        self.final_proj = nn.Linear(d, d, bias=True)


class _TokenConfidence(_Holder):
    def __init__(self, d):
        super().__init__()
        self.token = nn.Sequential(nn.Linear(d, 1), nn.Sigmoid())


class _Conf(dict):
    """dict with attribute access (the reference uses an OmegaConf node: ``conf.n_layers`` and ``conf["n_layers"]``)."""

    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p()


class LightGlue(nn.Module):
    default_conf = {
        "name": "lightglue", "input_dim": 256, "add_scale_ori": False, "descriptor_dim": 256, "n_layers": 9,
        "num_heads": 4, "flash": False, "mp": False, "depth_confidence": -1, "width_confidence": -1,
        "filter_threshold": 0.0, "checkpointed": False, "weights": None, "weights_from_version": "v0.1_arxiv",
        "loss": {"gamma": 1.0, "fn": "nll", "nll_balancing": 0.5},
    }
    required_data_keys = ["keypoints0", "keypoints1", "descriptors0", "descriptors1"]

    def __init__(self, conf, weights_path=None) -> None:
        super().__init__()
        merged = dict(self.default_conf)
        merged.update(dict(conf))
        self.conf = conf = _Conf(merged)
        if conf.add_scale_ori:
            raise NotImplementedError("add_scale_ori=True is not built (no reference config uses it)")
        d, h, n = conf.descriptor_dim, conf.num_heads, conf.n_layers
        self.input_proj = nn.Linear(conf.input_dim, d, bias=True) if conf.input_dim != d else nn.Identity()
        head_dim = d // h
        self.posenc = _PosEnc(2, head_dim, head_dim)
        self.transformers = nn.ModuleList([_TransformerLayer(d) for _ in range(n)])
        self.log_assignment = nn.ModuleList([_MatchAssignment(d) for _ in range(n)])
        self.token_confidence = nn.ModuleList([_TokenConfidence(d) for _ in range(n - 1)])
        self._handle = None
        self._sig = None
        self._ws = None
        self._ws_call = None               # a caller's own workspace for the forwards inside using_workspace()
        if weights_path is not None:
            self.load_state_dict(torch.load(weights_path, map_location="cpu", weights_only=True))

    # ---- engine plumbing ----------------------------------------------------------------------
    def _engine(self, device):
        if device.type != "cuda":
            raise RuntimeError("LightGlue (MI355X build) runs on a HIP device only; there is no CPU path in this package")
        lib = _lib.load()
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._handle is None or self._handle[1] != idx:
            self._free()
            cfg = _lib.Kp2dLgConfig()
            cfg.struct_size = C.sizeof(cfg)
            cfg.input_dim, cfg.descriptor_dim = int(self.conf.input_dim), int(self.conf.descriptor_dim)
            cfg.n_layers, cfg.num_heads, cfg.device = int(self.conf.n_layers), int(self.conf.num_heads), idx
            h = C.c_void_p()
            _lib.check(lib.kp2d_lg_create(C.byref(cfg), C.byref(h)))
            self._handle, self._sig = (h, idx), None
        sig = tuple((id(t), t._version) for t in self.state_dict(keep_vars=True).values())
        if sig != self._sig:
            h = self._handle[0]
            for k, t in self.state_dict().items():
                a = np.ascontiguousarray(t.detach().cpu().numpy(), dtype=np.float32)
                shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
                _lib.check(lib.kp2d_lg_set_weight(h, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim))
            _lib.check(lib.kp2d_lg_finalize_weights(h))
            self._sig = sig
        return lib, self._handle[0]

    def using_workspace(self, ws):
        """Context manager: forwards inside take ``ws`` (a uint8 device tensor, grown by the caller) instead of the module's
        cached workspace — one buffer per slot for streams that keep several forwards in flight (pipeline.FrameStream)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            prev, self._ws_call = self._ws_call, ws
            try:
                yield
            finally:
                self._ws_call = prev
        return cm()

    def workspace_bytes(self, b, m, n, device):
        lib, h = self._engine(torch.device(device))
        return int(lib.kp2d_lg_workspace_bytes(h, b, m, n))

    def _free(self):
        if getattr(self, "_handle", None) is not None:
            _lib.load().kp2d_lg_destroy(self._handle[0])
            self._handle = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass

    def expected_weights(self):
        """[(key, shape)] the engine expects, in the reference's registration order (kp2d_lg_weight_info)."""
        lib, h = self._engine(torch.device("cuda", torch.cuda.current_device()))
        out = []
        key, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
        for i in range(lib.kp2d_lg_num_weights(h)):
            _lib.check(lib.kp2d_lg_weight_info(h, i, C.byref(key), shape, C.byref(nd)))
            out.append((key.value.decode(), tuple(shape[j] for j in range(nd.value))))
        return out

    # ---- forward (reference lightglue.py:484-614) ------------------------------------------------
    def forward(self, data: dict) -> dict:
        for key in self.required_data_keys:
            assert key in data, f"Missing key {key} in data"
        if self.training:
            raise NotImplementedError("training mode (losses, checkpointing) is outside the built inference path")
        if self.conf.depth_confidence > 0:
            # the reference itself cannot run this: check_if_stop / get_pruning_mask read self.confidence_thresholds
            # (lightglue.py:624, :636), which this fork never defines (only the method confidence_threshold, :615)
            raise NotImplementedError("depth_confidence > 0 (early stopping) is not built; the reference raises "
                                      "AttributeError on it (confidence_thresholds is never defined)")
        if self.conf.width_confidence > 0:
            raise NotImplementedError("width_confidence > 0 (point pruning) is not built (reference configs leave it at -1)")
        kpts0, kpts1 = data["keypoints0"], data["keypoints1"]
        desc0, desc1 = data["descriptors0"].contiguous(), data["descriptors1"].contiguous()
        b, m, _ = kpts0.shape
        b, n, _ = kpts1.shape
        dev = kpts0.device
        assert desc0.shape[-1] == self.conf.input_dim
        assert desc1.shape[-1] == self.conf.input_dim
        lib, h = self._engine(dev)

        def image_size(view):
            size = data.get(view, {}).get("image_size") if isinstance(data.get(view), dict) else None
            if size is None:
                return None
            size = torch.as_tensor(size, device=dev, dtype=torch.float32)
            return (size.expand(b, 2) if size.dim() == 1 else size).contiguous()

        size0, size1 = image_size("view0"), image_size("view1")
        # extension of this build: padded keypoint sets.  data["num_keypoints0" / "num_keypoints1"] ([B] int32) say how many
        # rows of each set exist; the rest is padding (no key in any attention, no assignment mass, matches -1).  The
        # reference gets exactly the selected rows (a new shape every frame, visual_odometry.py:198-258); a replayed HIP
        # graph needs static shapes (pipeline.FrameStream(match="lightglue")).
        cnt0, cnt1 = data.get("num_keypoints0"), data.get("num_keypoints1")
        if (cnt0 is None) != (cnt1 is None):
            raise ValueError("num_keypoints0 and num_keypoints1 go together")
        if cnt0 is not None:
            if size0 is None or size1 is None:
                raise ValueError("padded keypoint sets need view0 / view1 image_size")
            cnt0 = cnt0.to(device=dev, dtype=torch.int32).contiguous()
            cnt1 = cnt1.to(device=dev, dtype=torch.int32).contiguous()
            if cnt0.numel() != b or cnt1.numel() != b:
                raise ValueError("num_keypoints0 / num_keypoints1 must hold one count per pair")
        f32 = lambda t: t.to(torch.float32).contiguous()
        kpts0, kpts1, desc0, desc1 = f32(kpts0), f32(kpts1), f32(desc0), f32(desc1)
        d = self.conf.descriptor_dim
        scores = torch.empty(b, m + 1, n + 1, device=dev)
        m0 = torch.empty(b, m, dtype=torch.int64, device=dev)
        m1 = torch.empty(b, n, dtype=torch.int64, device=dev)
        ms0, ms1 = torch.empty(b, m, device=dev), torch.empty(b, n, device=dev)
        ref = torch.empty(b * (m + n), d, device=dev)      # one buffer: the library copies both images' rows at once
        ref0, ref1 = ref[:b * m].view(b, m, d), ref[b * m:].view(b, n, d)
        need = lib.kp2d_lg_workspace_bytes(h, b, m, n)
        if self._ws_call is not None:
            if self._ws_call.numel() < need or self._ws_call.device != dev:
                raise RuntimeError(f"using_workspace(): the buffer holds {self._ws_call.numel()} bytes, the forward needs {need}")
            ws = self._ws_call
        else:
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
            ws = self._ws
        stream = torch.cuda.current_stream(dev).cuda_stream
        _lib.check(lib.kp2d_lg_forward_counts(h, _ptr(kpts0), _ptr(kpts1), _ptr(desc0), _ptr(desc1), _ptr(size0), _ptr(size1),
                                              _ptr(cnt0), _ptr(cnt1), b, m, n, float(self.conf.filter_threshold), _ptr(scores),
                                              _ptr(m0), _ptr(m1), _ptr(ms0), _ptr(ms1), _ptr(ref0), _ptr(ref1),
                                              _ptr(ws), ws.numel(), C.c_void_p(stream)))
        # (no pruning is built: every point survives all layers; one fill instead of two ones_like * n)
        prune = torch.full((b * (m + n),), float(self.conf.n_layers), device=dev)
        return {
            "matches0": m0, "matches1": m1, "matching_scores0": ms0, "matching_scores1": ms1,
            "ref_descriptors0": ref0[:, None], "ref_descriptors1": ref1[:, None],
            "log_assignment": scores,
            "prune0": prune[:b * m].view(b, m), "prune1": prune[b * m:].view(b, n),
        }


__main_model__ = LightGlue
