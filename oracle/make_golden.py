"""Generate tests/golden/*.npz by running the REFERENCE (imported read-only from /root/reference).

Runs ONLY in the build container (the reference never travels to the GPU box).  Usage:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/oracle/make_golden.py [--only NAME]

Fixtures hold arrays + JSON metadata only: inputs are regenerated from
``oracle/weights.py`` (seeded numpy), expected outputs come from the reference's
``forward`` / ``post_processing``.  Selector results (K1/K3) are computed from the
reference's outputs with the restated selectors in ``oracle/kp2d_oracle.py``
(their modules need cv2/kornia and cannot be imported here).
"""
from __future__ import annotations

import argparse
import contextlib
import io
import json
import os
import sys

import numpy as np

import importlib.util

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
sys.dont_write_bytecode = True
# The repo root holds import ALIASES named like the reference's packages (src/, lightglue/): it must never be on
# sys.path while the reference is imported, or "from src.kp2dtiny..." resolves to the product.  The oracle's own
# modules are therefore loaded by file path, and every repo path (also the script directory and "") is dropped.
sys.path[:] = [REFERENCE] + [p for p in sys.path
                             if p not in ("", ".", REFERENCE) and not os.path.abspath(p).startswith(REPO)]


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


import torch  # noqa: E402

orc = _load_by_path("_golden_kp2d_oracle", os.path.join(REPO, "oracle", "kp2d_oracle.py"))
_weights = _load_by_path("_golden_weights", os.path.join(REPO, "oracle", "weights.py"))
spread_state_dict, synthetic_frames = _weights.spread_state_dict, _weights.synthetic_frames
state_dict_for = _weights.state_dict_for


def reference_module():
    """The reference's model module — and nothing else: refuses to run against anything outside /root/reference."""
    import src.kp2dtiny.models.kp2dtiny as ref
    where = os.path.realpath(ref.__file__)
    if not where.startswith(REFERENCE + os.sep):
        raise RuntimeError(f"make_golden.py imported {where}: fixtures must come from the reference under {REFERENCE}, "
                           "never from this repo's aliases")
    for name, mod in list(sys.modules.items()):
        f = getattr(mod, "__file__", None)
        if (name == "src" or name.startswith("src.")) and f and not os.path.realpath(f).startswith(REFERENCE + os.sep):
            raise RuntimeError(f"module {name} resolved to {f}, outside the reference")
    return ref

# name -> (config, v3, n_classes, H, W, B, frame seed, smooth, dense stride, taps?)
CASES = {
    "v2_N_120x160": ("N", False, 28, 120, 160, 1, 7, False, 1, False),      # BASELINE cfg 0 (full outputs)
    "v2_N_32x48_taps": ("N", False, 28, 32, 48, 1, 9, False, 1, True),      # kernel bring-up, all intermediates
    "v3_SA_32x48_taps": ("S_A", True, 28, 32, 48, 1, 9, False, 1, True),    # attention bring-up
    "v2_S_120x160": ("S", False, 28, 120, 160, 2, 8, False, 1, False),
    "v2_S_240x320": ("S", False, 28, 240, 320, 2, 7, False, 4, False),      # BASELINE cfg 1/2 shape
    "v2_S_240x320_smooth": ("S", False, 28, 240, 320, 1, 9, True, 4, False),
    "v2_S_480x640": ("S", False, 28, 480, 640, 1, 8, False, 8, False),      # BASELINE cfg 5's extractor shape
    "v2_SA_120x160": ("S_A", False, 28, 120, 160, 1, 8, False, 2, False),
    "v3_S_120x160": ("S", True, 19, 120, 160, 1, 8, False, 2, False),
    "v3_SA_240x320": ("S_A", True, 28, 240, 320, 1, 7, False, 4, False),    # demo.py config
    "v3_SA_480x640": ("S_A", True, 19, 480, 640, 1, 7, False, 8, False),    # BASELINE cfg 3 shape
    "v3_NA_120x160": ("N_A", True, 28, 120, 160, 1, 7, False, 2, False),
    # config-selectable poolers; GeM's PixelUnshuffle(4) needs H/4, W/4 divisible by 4
    "v2_GEM_SA_128x160": ("GEM_S_A", False, 28, 128, 160, 1, 8, False, 4, False),
    "v2_GEM_N_64x96": ("GEM_N", False, 28, 64, 96, 2, 8, False, 2, False),
    "v3_CONVAP_SA_120x160": ("CONVAP_S_A", True, 19, 120, 160, 1, 8, False, 4, False),
    # depth=True variants (kp2dtiny.py:402-437, segmentation.py:190-193): config name + "+depth"
    "v2_S_depth_64x96": ("S+depth", False, 28, 64, 96, 1, 8, False, 2, False),
    "v3_SA_depth_64x96": ("S_A+depth", True, 19, 64, 96, 1, 8, False, 2, False),
    # remaining channel sets of get_config (kp2dtiny.py:46-219): LARGE_D (128-d descriptors, ConvAP), TINY_F (cell 8)
    "v2_D_64x96": ("D", False, 28, 64, 96, 1, 8, False, 2, False),
    "v2_F_64x96": ("F", False, 28, 64, 96, 1, 8, False, 2, False),
    "v3_D_64x96": ("D", True, 19, 64, 96, 1, 8, False, 2, False),
    "v3_DA_64x96": ("D_A", True, 19, 64, 96, 1, 8, False, 2, False),
    # to_mcu=True (kp2dtiny.py:271-273): TransposedConvUpsampleModel instead of PixelShuffle, ReLU.  The reference's
    # get_config MUTATES its module-level table for to_mcu, so these cases must run in their own process (--only).
    "v2_S_mcu_64x96": ("S+mcu", False, 28, 64, 96, 2, 8, False, 2, False),
    "v3_SA_mcu_64x96": ("S_A+mcu", True, 19, 64, 96, 1, 8, False, 2, False),
    "v2_NA_mcu_depth_64x96": ("N_A+mcu+depth", False, 28, 64, 96, 1, 8, False, 2, False),
    # KP2DTinyV3(use_color=False): one-channel frames (kp2dtiny.py:682, :718-721); frames = channel 0 of the RGB ones
    "v3_S_gray_64x96": ("S+gray", True, 19, 64, 96, 2, 8, False, 2, False),
    # trained-like statistics (nano-vs-slam_amd/synthetic.py::trained_like_tensor): BatchNorm running_var over six decades,
    # heavy-tailed conv weights scaled per output channel — what the f16x3 operand split has to survive on a checkpoint
    "v2_S_trained_120x160": ("S", False, 28, 120, 160, 1, 8, False, 1, False, "trained"),
    "v3_SA_trained_64x96": ("S_A", True, 19, 64, 96, 1, 8, False, 1, False, "trained"),
}


def build_reference(config, v3, n_classes, recipe="spread"):
    ref = reference_module()
    KP2DTinyV2, KP2DTinyV3, get_config, tiny_factory = ref.KP2DTinyV2, ref.KP2DTinyV3, ref.get_config, ref.tiny_factory
    base, *mods = config.split("+")
    with contextlib.redirect_stdout(io.StringIO()):
        if "depth" in mods or "gray" in mods:
            import copy
            conf = copy.deepcopy(get_config(base, to_mcu="mcu" in mods, v3=v3))
            extra = {"depth": True} if "depth" in mods else {}
            if "gray" in mods:
                extra["use_color"] = False
            model = (KP2DTinyV3 if v3 else KP2DTinyV2)(**conf, nClasses=n_classes, **extra)
        else:
            model = tiny_factory(base, n_classes, to_mcu="mcu" in mods, v3=v3)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = state_dict_for(recipe, shapes)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    model.eval()
    model.training = False
    return model, shapes, sd


def gaps(score_flat, ks, thr=0.7):
    """Guard metadata: min score gap at each top-k boundary and at the threshold."""
    s = np.sort(score_flat)[::-1]
    g = {"thr": float(np.min(np.abs(score_flat - thr)))}
    for k in ks:
        if k < len(s):
            g[f"k{k}"] = float(s[k - 1] - s[k])
    return g


def run_case(name, out_dir):
    config, v3, ncls, H, W, B, seed, smooth, stride, want_taps = CASES[name][:10]
    recipe = CASES[name][10] if len(CASES[name]) > 10 else "spread"
    model, shapes, sd = build_reference(config, v3, ncls, recipe)
    cfg = orc.get_config(config, v3)
    # the oracle's own key/shape table must equal the reference's registration order
    mine = orc.state_dict_shapes(cfg, ncls)
    assert list(mine.items()) == [(k, tuple(s)) for k, s in shapes.items()], f"{name}: state-dict layout drift"

    x = synthetic_frames(B, H, W, seed, smooth)[:, :cfg["in_channels"]].copy()
    taps = {}
    hooks = []
    if want_taps:
        def mk(nm):
            def hook(_m, _i, o):
                if isinstance(o, torch.Tensor):
                    taps[nm] = o.detach().numpy().copy()
            return hook
        for nm, mod in model.named_modules():
            if nm and nm.count(".") <= 3 and not nm.endswith((".quant", ".dequant", ".relu", ".dropout", ".pool")):
                hooks.append(mod.register_forward_hook(mk(nm)))
    with torch.no_grad():
        fwd = model(torch.from_numpy(x))
        fwd_np = {k: v.numpy().copy() for k, v in fwd.items()}
        post = model.post_processing(dict(fwd), H, W)
        post_np = {k: v.numpy().copy() for k, v in post.items()}
    for h in hooks:
        h.remove()

    arrays = {}
    meta = dict(name=name, config=config, v3=v3, n_classes=ncls, H=H, W=W, B=B, frame_seed=seed, smooth=smooth,
                weight_seed=1234, head_gain=1.0, weights=recipe, dense_stride=stride, torch=torch.__version__,
                n_params=int(sum(p.numel() for p in model.parameters())))
    arrays["fwd_score"] = fwd_np["score"]
    arrays["fwd_shift"] = fwd_np["coord"]
    arrays["fwd_vlad"] = fwd_np["vlad"]
    arrays["fwd_feat"] = fwd_np["feat"][:, :, ::stride, ::stride]
    arrays["fwd_seg"] = fwd_np["seg"][:, :, ::stride, ::stride]
    if "depth" in fwd_np:
        arrays["fwd_depth"] = fwd_np["depth"]
    arrays["post_score"] = post_np["score"]
    arrays["post_coord"] = post_np["coord"]
    arrays["post_feat"] = post_np["feat"]
    seg_ids = post_np["seg"]
    assert seg_ids.dtype == np.int64 and seg_ids.max() < 256
    arrays["post_seg_u8"] = seg_ids.astype(np.uint8)
    # top-2 margin of the seg map: argmax parity is only demanded where the margin is clear
    part = np.partition(fwd_np["seg"], -2, axis=1)
    arrays["seg_margin_f16"] = (part[:, -1] - part[:, -2]).astype(np.float16)

    # selectors on the reference's outputs
    sel = {}
    guard = []
    for b in range(B):
        sc, co, ft = post_np["score"][b:b + 1], post_np["coord"][b:b + 1], post_np["feat"][b:b + 1]
        flat = sc.reshape(-1)
        arrays[f"keep_idx_{b}"] = np.nonzero(flat > 0.7)[0].astype(np.int32)
        for k in (300, 1000, 4000):
            idx, _, _ = orc.select_k1(sc, co, ft, 0.7, k)
            arrays[f"k1_top{k}_idx_{b}"] = idx.astype(np.int32)
        guard.append(gaps(flat, (300, 1000, 1024, 4000)))
    k3_idx, k3_s, _, _ = orc.select_k3(post_np["score"], post_np["coord"], post_np["feat"], k=min(1024, flat.size))
    arrays["k3_idx"] = k3_idx.astype(np.int32)
    arrays["k3_scores"] = k3_s
    meta["gaps"] = guard
    for k, v in taps.items():
        arrays["tap:" + k] = v
    if want_taps:
        # only_encoder (kp2dtiny.py:515-518) and NetVLAD.init_params (aggregators/netvlad.py:51-63) on seeded inputs;
        # init_params REPLACES the model's NetVLAD parameters, so it runs last
        with torch.no_grad():
            arrays["only_encoder"] = model.only_encoder(torch.from_numpy(x)).numpy().copy()
        assert np.max(np.abs(orc.only_encoder(x, {k: np.asarray(v) for k, v in sd.items()}, cfg) - arrays["only_encoder"])) < 1e-5
        g = np.random.default_rng(77)
        K, C = model.vlad_head.netvlad.num_clusters, model.vlad_head.netvlad.dim
        clsts = g.standard_normal((K, C)).astype(np.float32)
        descs = g.standard_normal((500, C)).astype(np.float32)
        descs /= np.linalg.norm(descs, axis=1, keepdims=True)
        model.init_netvlad(clsts.copy(), descs.copy())
        nv = model.vlad_head.netvlad
        arrays["init_clsts"], arrays["init_descs"] = clsts, descs
        arrays["init_alpha"] = np.float64(nv.alpha)
        arrays["init_conv_weight"] = nv.conv.weight.detach().numpy().copy()
        arrays["init_centroids"] = nv.centroids.detach().numpy().copy()
        a, c, w = orc.netvlad_init_params(clsts.copy(), descs.copy())
        assert abs(a - nv.alpha) < 1e-9 * abs(a) and np.array_equal(w, arrays["init_conv_weight"])

    # cross-check the numpy oracle against the reference right here (fp32)
    p32 = {k: np.asarray(v) for k, v in sd.items()}
    o_f = orc.forward(x, p32, cfg)
    o_p = orc.post_processing(o_f, H, W, cfg)
    errs = {k: float(np.max(np.abs(o_f[k] - fwd_np[k]))) for k in ("score", "coord", "feat", "vlad", "seg")}
    errs["post_feat"] = float(np.max(np.abs(o_p["feat"] - post_np["feat"])))
    errs["post_coord"] = float(np.max(np.abs(o_p["coord"] - post_np["coord"])))
    errs["seg_argmax_mismatch"] = int((o_p["seg"] != post_np["seg"]).sum())
    if "depth" in fwd_np:
        errs["depth"] = float(np.max(np.abs(o_f["depth"] - fwd_np["depth"])))
    meta["oracle_vs_reference_fp32"] = errs
    # fp64 oracle vs reference: sizes the tolerance any faithful fp32 implementation needs
    p64 = orc.cast_params(p32, np.float64)
    o64 = orc.forward(x.astype(np.float64), p64, cfg)
    meta["reference_vs_fp64"] = {k: float(np.max(np.abs(o64[k] - fwd_np[k]))) for k in
                                 ("score", "coord", "feat", "vlad", "seg")}
    meta["absmax"] = {k: float(np.max(np.abs(fwd_np[k]))) for k in ("score", "coord", "feat", "vlad", "seg")}
    meta["frac_above_thr"] = float((post_np["score"] > 0.7).mean())

    path = os.path.join(out_dir, name + ".npz")
    np.savez_compressed(path, meta=np.frombuffer(json.dumps(meta).encode(), np.uint8), **arrays)
    print(f"{name}: {os.path.getsize(path) / 1e6:.2f} MB  oracle-vs-ref {errs}  ref-vs-fp64 {meta['reference_vs_fp64']}",
          flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--out", default=os.path.join(REPO, "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    torch.set_num_threads(8)
    for name in (a.only or CASES):
        run_case(name, a.out)


if __name__ == "__main__":
    main()
