"""Build libkp2d_hip.so (gfx950) in-tree with hipcc.  No cmake, no JIT cache.

    python3 nano-vs-slam_amd/csrc/build.py [--force]

The shared object sits next to the sources so it travels with the repo snapshot to the GPU box
(it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["conv3x3.hip", "conv3x3_f16.hip", "conv3x3_wsm.hip", "conv3x3_s16.hip", "head3x3.hip", "netvlad.hip", "post.hip", "attention.hip", "mff_tail.hip", "match.hip", "lightglue.hip", "kp2d_api.cpp",
           "lightglue_api.cpp"]
HEADERS = ["kp2d_kernels.h", "device_guard.h", "device_logic.h", "conv_common.h", "conv_epilogue.inc", os.path.join("..", "..", "include", "kp2d.h"),
           os.path.join("..", "..", "include", "kp2d_lightglue.h")]
LIB = os.path.join(HERE, "libkp2d_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
# per-file additions.  attention.hip: without -fno-honor-nans every fmaxf of the running-maximum chain is preceded by
# v_max_f32 x, x (quieting a signalling NaN): 4 of ~60 vector instructions per 32 x 32 tile of a kernel bound by vector
# issue.  (Only NaNs are assumed away; the -inf masks of ragged key tiles are ordinary values.)
_NN = ["-fno-honor-nans"]
FILE_FLAGS = {} if os.environ.get("KP2D_NO_FILE_FLAGS") else {f: _NN for f in os.environ.get("KP2D_NN_FILES", "attention.hip").split(",")}      # (the switches: A/B builds)


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True, ablate: bool = False, exp: str = "", defines: tuple = ()) -> str:
    """ablate=True: the timing-ablation build (-DKP2D_ABLATE: KP2D_DBG bits switch phases of the conv kernels off,
    results are then wrong by design) into build_exp/libkp2d_ablate.so — selected with KP2D_LIB, never the product.
    exp="NAME", defines=("X", ...): an experimental build with -DX ... into build_exp/libkp2d_NAME.so (A/B runs, tools/ab_variants.sh)."""
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objdir = os.path.join(HERE, "build_exp", "obj_ablate") if ablate else os.path.join(HERE, "build")
    if exp:
        objdir = os.path.join(HERE, "build_exp", "obj_" + exp)
    os.makedirs(objdir, exist_ok=True)
    lib = os.path.join(HERE, "build_exp", "libkp2d_ablate.so") if ablate else LIB
    if exp:
        lib = os.path.join(HERE, "build_exp", "libkp2d_%s.so" % exp)
    flags = FLAGS + (["-DKP2D_ABLATE"] if ablate else []) + ["-D" + d for d in defines]

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src + ".o")
        path = os.path.join(HERE, src)
        if force or _stale(obj, [path] + hdrs):
            cmd = [HIPCC] + flags + FILE_FLAGS.get(src, []) + ["-c", path, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True, cwd=HERE)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    if force or _stale(lib, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True, cwd=HERE)
    return lib


if __name__ == "__main__":
    argv = sys.argv[1:]
    exp = argv[argv.index("--exp") + 1] if "--exp" in argv else ""
    defs = tuple(argv[i + 1] for i, a in enumerate(argv) if a == "--define")
    print(build(force="--force" in argv, ablate="--ablate" in argv, exp=exp, defines=defs))
