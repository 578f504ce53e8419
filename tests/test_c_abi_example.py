"""The C ABI without Python in the loop: examples/c_abi_forward.cpp is compiled with hipcc against libkp2d_hip.so and
run as its own process; its checksums must equal what the Python host layer produces from the same LCG tensors."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def lcg_stream(seed, n):
    """Vectorised replay of the example's LCG: s_{i+1} = a s_i + c (mod 2^32), value = (s >> 8) / 2^24."""
    out = np.empty(n, np.float32)
    s = np.uint64(seed)
    a, c, mask = np.uint64(1664525), np.uint64(1013904223), np.uint64(0xFFFFFFFF)
    for i in range(n):
        s = (s * a + c) & mask
        out[i] = np.float32(int(s) >> 8) * np.float32(1.0 / 16777216.0)
    return out


def test_c_program_matches_python_host(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    libdir = os.path.join(ROOT, "nano-vs-slam_amd", "csrc")
    exe = str(tmp_path / "c_abi_forward")
    subprocess.run([hipcc, "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_abi_forward.cpp"),
                    "-L" + libdir, "-lkp2d_hip", "-Wl,-rpath," + libdir, "-o", exe], check=True, timeout=600)
    res = subprocess.run([exe, "2"], check=True, capture_output=True, text=True, timeout=300)
    got = {}
    for line in res.stdout.splitlines():
        parts = line.split()
        for k, v in zip(parts[0::2], parts[1::2]):
            got[k] = float(v)

    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
    from nano_vs_slam_amd.selectors import select_topk
    model = tiny_factory("S", 28).to("cuda:0")
    order = [k for k, _ in model._get_engine(torch.device("cuda:0")).expected()]      # kp2d_weight_info order
    sd = {k: t for k, t in model.state_dict().items() if k.endswith("num_batches_tracked")}
    for i, k in enumerate(order):
        t = model.state_dict()[k]
        u = lcg_stream(12345 + 977 * i, t.numel())
        fan = int(np.prod(t.shape[1:])) if t.dim() > 1 else 1
        amp = np.float32(2.0) * np.sqrt(np.float32(3.0) / np.float32(fan)) if t.dim() > 1 else np.float32(0.4)
        positive = k.endswith("running_var") or k.endswith("bn.weight")
        v = (np.float32(0.5) + u) if positive else (u - np.float32(0.5)) * amp
        sd[k] = torch.from_numpy(v.astype(np.float32).reshape(tuple(t.shape)))
    model.load_state_dict(sd)
    model = model.to("cuda:0").eval()
    model.training = False
    x = torch.from_numpy((lcg_stream(777, 2 * 3 * 64 * 96) * np.float32(2.0) - np.float32(1.0)).reshape(2, 3, 64, 96)).to("cuda:0")
    with torch.no_grad():
        out = model(x)
        ref = {"score_sum": out["score"].double().sum().item(), "score_sq": (out["score"].double() ** 2).sum().item(),
               "shift_sum": out["coord"].double().sum().item(), "vlad_sq": (out["vlad"].double() ** 2).sum().item(),
               "vlad_sum": out["vlad"].double().sum().item()}
        post = model.post_processing(out, 64, 96)
        ref["desc_sum"] = post["feat"].double().sum().item()
        ref["desc_sq"] = (post["feat"].double() ** 2).sum().item()
        ref["coord_sum"] = post["coord"].double().sum().item()
        idx, _, _ = select_topk(post["score"], 300)
    ref["topk_first"] = float(idx[0, 0])
    ref["topk_idx_sum"] = float(idx.long().sum())
    assert abs(got["vlad_sq"] - 2.0) < 1e-4                      # two unit-norm NetVLAD vectors
    for k, v in ref.items():
        assert abs(got[k] - v) <= 2e-4 * max(1.0, abs(v)), (k, got[k], v)
