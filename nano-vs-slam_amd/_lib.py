"""ctypes binding of libkp2d_hip.so (the C ABI declared in include/kp2d.h).

There is no CPU path behind this module: if the shared object is missing or does not load, every
entry point raises.  Build it with ``python3 -c "import __graft_entry__ as g; g.build()"`` (hipcc,
gfx950, in-tree) — see nano-vs-slam_amd/csrc/build.py.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# KP2D_LIB: another build of the SAME library (kernel A/B timing on one box: tools/ab_variants.sh); never a fallback
LIB_PATH = os.environ.get("KP2D_LIB") or os.path.join(_HERE, "csrc", "libkp2d_hip.so")

KP2D_FWD_EVAL = 1
KP2D_FWD_ONLY_ENCODER = 2
PRECISIONS = {"fp32": 0, "f16x3": 1}
GLOBAL_DESCRIPTORS = {"netvlad": 0, "gem": 1, "convap": 2}
UPSCALE_METHODS = {"pixelshuffle": 0, "convtranspose": 1}


class Kp2dConfig(C.Structure):
    """struct kp2d_config (include/kp2d.h)."""

    _fields_ = [
        ("struct_size", C.c_int32),
        ("version", C.c_int32),
        ("channel_dims", C.c_int32 * 6),
        ("nfeatures", C.c_int32),
        ("n_classes", C.c_int32),
        ("num_clusters", C.c_int32),
        ("encoder_dim", C.c_int32),
        ("downsample", C.c_int32),
        ("use_attention", C.c_int32),
        ("leaky_relu", C.c_int32),
        ("remove_softmax", C.c_int32),
        ("device", C.c_int32),
        ("global_descriptor", C.c_int32),
        ("remove_netvlad", C.c_int32),
        ("depth", C.c_int32),
        ("upscale_method", C.c_int32),
        ("in_channels", C.c_int32),
    ]


class Kp2dLgConfig(C.Structure):
    """struct kp2d_lg_config (include/kp2d_lightglue.h)."""

    _fields_ = [
        ("struct_size", C.c_int32),
        ("input_dim", C.c_int32),
        ("descriptor_dim", C.c_int32),
        ("n_layers", C.c_int32),
        ("num_heads", C.c_int32),
        ("device", C.c_int32),
    ]


class Kp2dError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"kp2d error {code}: {msg}")
        self.code = code


# every symbol include/kp2d.h declares: (restype, argtypes)
_P, _F, _I32, _I64 = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_int64)
SIGNATURES = {
    "kp2d_last_error": (C.c_char_p, []),
    "kp2d_abi_version": (C.c_int32, []),
    "kp2d_create": (C.c_int, [C.POINTER(Kp2dConfig), C.POINTER(_P)]),
    "kp2d_destroy": (None, [_P]),
    "kp2d_num_weights": (C.c_int, [_P]),
    "kp2d_weight_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), _I64, C.POINTER(C.c_int)]),
    "kp2d_set_weight": (C.c_int, [_P, C.c_char_p, _P, _I64, C.c_int]),
    "kp2d_finalize_weights": (C.c_int, [_P]),
    "kp2d_packed_bytes": (C.c_size_t, [_P]),
    "kp2d_export_packed": (C.c_int, [_P, _P, _P]),
    "kp2d_import_packed": (C.c_int, [_P, _P, _P]),
    "kp2d_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int]),
    "kp2d_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_uint32, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kp2d_forward_frames": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, _P, _P, _P, _P, _P, _P, _P,
                                      C.c_size_t, _P]),
    "kp2d_post": (C.c_int, [_P, _P, _P, _P, _P] + [C.c_int] * 11 + [_P, _P, _P, _P, C.c_int, _P]),
    "kp2d_vlad_dim": (C.c_size_t, [_P, C.c_int, C.c_int]),
    "kp2d_select_topk": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P]),
    "kp2d_gather_keypoints": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P]),
    "kp2d_select_keypoints": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P, _P]),
    "kp2d_preprocess": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, _P]),
    "kp2d_match_descriptors": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, _P, _P, _P, _P, _P]),
    "kp2d_match_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "kp2d_match_descriptors_ex": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _P, _P, C.c_uint32,
                                            _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "kp2d_match_pairs": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "kp2d_match_topk_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "kp2d_match_topk_pairs": (C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P,
                                        C.c_size_t, _P]),
    "kp2d_set_profiling": (C.c_int, [_P, C.c_int]),
    "kp2d_profile_count": (C.c_int, [_P]),
    "kp2d_profile_get": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), _F,
                                   C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "kp2d_set_chunk_frames": (C.c_int, [_P, C.c_int]),
    "kp2d_set_option": (C.c_int, [_P, C.c_char_p, C.c_long]),
    "kp2d_set_seg_ids": (C.c_int, [_P, _P, C.c_size_t]),
    "kp2d_set_precision": (C.c_int, [_P, C.c_int]),
    "kp2d_get_precision": (C.c_int, [_P]),
    "kp2d_set_tap": (C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    # include/kp2d_lightglue.h
    "kp2d_lg_create": (C.c_int, [C.POINTER(Kp2dLgConfig), C.POINTER(_P)]),
    "kp2d_lg_destroy": (None, [_P]),
    "kp2d_lg_num_weights": (C.c_int, [_P]),
    "kp2d_lg_weight_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), _I64, C.POINTER(C.c_int)]),
    "kp2d_lg_set_weight": (C.c_int, [_P, C.c_char_p, _P, _I64, C.c_int]),
    "kp2d_lg_finalize_weights": (C.c_int, [_P]),
    "kp2d_lg_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int, C.c_int]),
    "kp2d_lg_forward": (C.c_int, [_P] + [_P] * 6 + [C.c_int] * 3 + [C.c_float] + [_P] * 7 + [_P, C.c_size_t, _P]),
    "kp2d_lg_forward_counts": (C.c_int, [_P] + [_P] * 8 + [C.c_int] * 3 + [C.c_float] + [_P] * 7 + [_P, C.c_size_t, _P]),
}

_lib = None


def load() -> C.CDLL:
    """Load the library once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the kp2dtiny HIP library is not built. "
            "Run `python3 -c 'import __graft_entry__ as g; g.build()'` at the repo root (needs hipcc). "
            "There is no CPU fallback for this path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library drift
        fn.restype = res
        fn.argtypes = args
    if lib.kp2d_abi_version() != 1:
        raise RuntimeError("libkp2d_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise Kp2dError(rc, load().kp2d_last_error().decode(errors="replace"))
