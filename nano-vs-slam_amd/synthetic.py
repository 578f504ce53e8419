"""Seeded synthetic weights and frames (pure numpy): what demo.py, eval_multitask.py, bench.py and the tools load into
the model when no checkpoint is available (the reference ships none, README.md:220-221; there is no network here).

The same generators produced the golden fixtures (oracle/weights.py re-exports them), so everything is keyed by the
state-dict key NAME (crc32), not by iteration order: the product model, the oracle and the reference get bit-identical
tensors whatever order they enumerate their parameters in.

Recipe:
  conv weight  ~ N(0, 2/fan_in)          conv bias ~ N(0, 0.1^2)
  bn.weight    ~ U(0.5, 1.5)             bn.bias, bn.running_mean ~ N(0, 0.2^2)
  bn.running_var ~ U(0.5, 1.5)           LayerNorm g/b: 1/0 with small jitter
  netvlad.centroids ~ N(0, 0.3^2)        netvlad.conv.weight: N(0,2/fan_in) * 8
  *.convDb.weight *= head_gain
"""
from __future__ import annotations

import zlib

import numpy as np

__all__ = ["spread_tensor", "spread_state_dict", "synthetic_frames", "seeded_linear_state_dict",
           "trained_like_tensor", "trained_like_state_dict"]


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(key.encode())])


def spread_tensor(key: str, shape, seed: int = 1234, head_gain: float = 1.0) -> np.ndarray:
    """One tensor of the spread-init recipe, float32 (int64 for num_batches_tracked)."""
    shape = tuple(int(s) for s in shape)
    g = _rng(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, np.int64)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if leaf == "running_mean":
        return (g.standard_normal(shape) * 0.2).astype(np.float32)
    if leaf == "centroids":
        return (g.standard_normal(shape) * 0.3).astype(np.float32)
    if leaf == "g":  # channel LayerNorm gain (modules/segformer.py:66)
        return (1.0 + 0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "b":  # channel LayerNorm bias (modules/segformer.py:67)
        return (0.05 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "p":  # GeM exponent (aggregators/gem.py:10)
        return np.full(shape, 3.0, np.float32)
    if leaf == "weight":
        if len(shape) == 1:  # BatchNorm gamma
            return g.uniform(0.5, 1.5, shape).astype(np.float32)
        fan_in = int(np.prod(shape[1:]))
        w = g.standard_normal(shape) * np.sqrt(2.0 / fan_in)
        if key.endswith("netvlad.conv.weight"):
            w = w * 8.0
        if ".convDb." in key:
            w = w * head_gain
        return w.astype(np.float32)
    if leaf == "bias":
        parent = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
        if parent == "bn":
            return (g.standard_normal(shape) * 0.2).astype(np.float32)
        return (g.standard_normal(shape) * 0.1).astype(np.float32)
    raise KeyError(f"spread-init recipe has no rule for key {key!r}")


def spread_state_dict(shapes: dict, seed: int = 1234, head_gain: float = 1.0) -> dict:
    """shapes: {key: shape}.  Returns {key: np.ndarray}."""
    return {k: spread_tensor(k, s, seed, head_gain) for k, s in shapes.items()}


def _trained_var(prefix: str, n: int, seed: int) -> np.ndarray:
    """BatchNorm running_var of layer `prefix`: log-uniform over 1e-4 ... 1e2 (six decades, as trained nets show)."""
    return (10.0 ** _rng(seed, prefix + ".bn.running_var").uniform(-4.0, 2.0, n)).astype(np.float32)


def trained_like_tensor(key: str, shape, seed: int = 1234) -> np.ndarray:
    """One tensor of the "trained-like" recipe: statistics a checkpoint has and the spread recipe lacks.

      bn.running_var   log-uniform 1e-4 ... 1e2 per channel
      conv weights     heavy-tailed (Student t, 3 degrees of freedom) around N(0, 2/fan_in); in a conv + BN block row c
                       is scaled by sqrt(running_var[c]), so the conv's output variance is what its BatchNorm divides by
                       (as in a trained net) and activations stay O(1) while weights span ~1e-3 ... 1e2
      bn.running_mean  N(0, 0.3^2) * sqrt(running_var);  bn.weight U(0.3, 1.8);  bn.bias N(0, 0.3^2)
    Everything else as spread_tensor."""
    shape = tuple(int(s) for s in shape)
    leaf = key.rsplit(".", 1)[-1]
    parent = key.rsplit(".", 2)[-2] if key.count(".") >= 1 else ""
    g = _rng(seed, key)
    if parent == "bn" and leaf in ("running_var", "running_mean", "weight", "bias"):
        prefix = key[: -len(".bn." + leaf)]
        var = _trained_var(prefix, shape[0], seed)
        if leaf == "running_var":
            return var
        if leaf == "running_mean":
            return (g.standard_normal(shape) * 0.3 * np.sqrt(var)).astype(np.float32)
        if leaf == "weight":
            return g.uniform(0.3, 1.8, shape).astype(np.float32)
        return (g.standard_normal(shape) * 0.3).astype(np.float32)
    if leaf == "weight" and len(shape) == 4 and not key.endswith("netvlad.conv.weight"):
        fan_in = int(np.prod(shape[1:]))
        w = g.standard_t(3.0, shape) / np.sqrt(3.0) * np.sqrt(2.0 / fan_in)
        if parent == "conv":   # AnnotatedConvBnReLUModel: <prefix>.conv.weight + <prefix>.bn.*
            var = _trained_var(key[: -len(".conv.weight")], shape[0], seed)
            w = w * np.sqrt(var).reshape(-1, 1, 1, 1)
        return w.astype(np.float32)
    return spread_tensor(key, shape, seed)


def trained_like_state_dict(shapes: dict, seed: int = 1234) -> dict:
    return {k: trained_like_tensor(k, s, seed) for k, s in shapes.items()}


def synthetic_frames(B: int, H: int, W: int, seed: int = 7, smooth: bool = False) -> np.ndarray:
    """RGB frames in [-1, 1], float32 [B,3,H,W].

    ``smooth`` applies a 3x3 box filter so the score map has spatial structure.
    """
    x = np.random.default_rng(seed).random((B, 3, H, W), np.float32) * 2.0 - 1.0
    if smooth:
        p = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), mode="edge")
        acc = np.zeros_like(x)
        for dy in range(3):
            for dx in range(3):
                acc += p[:, :, dy:dy + H, dx:dx + W]
        x = (acc / 9.0 * 2.5).clip(-1, 1).astype(np.float32)
    return x


def seeded_linear_state_dict(shapes: dict, seed: int = 4321) -> dict:
    """Spread weights for a stack of Linear / LayerNorm layers (the LightGlue matcher): Linear ~ N(0, 1/fan_in) * gain,
    LayerNorm gamma ~ U(0.5, 1.5), biases ~ N(0, 0.1^2); keyed by name like ``spread_tensor``."""
    import math
    out = {}
    for k, shp in shapes.items():
        g = np.random.default_rng([seed, zlib.crc32(k.encode())])
        if k.endswith(".1.weight"):
            v = g.uniform(0.5, 1.5, shp)
        elif k.endswith("bias"):
            v = g.standard_normal(shp) * 0.1
        elif k == "posenc.Wr.weight":
            v = g.standard_normal(shp) * 2.0          # a few radians across the normalised image
        else:
            gain = 2.0 if ("Wqkv" in k or "to_qk" in k or "final_proj" in k) else 1.0
            v = g.standard_normal(shp) * gain / math.sqrt(shp[-1])
        out[k] = v.astype(np.float32)
    return out
