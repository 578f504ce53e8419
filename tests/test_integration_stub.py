"""INTEGRATION.md's binding stub is checked, not trusted.

CPU: every ``lib.kp2d_*(...)`` call printed in INTEGRATION.md passes exactly as many arguments as the prototype in
``include/*.h`` declares (two calls had lost an argument).
GPU: the fenced ctypes stub of section B is extracted, executed verbatim and compared with the package's own output.
"""
import ast
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _python_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```python\n(.*?)```", text, flags=re.S)


def _prototypes():
    protos = {}
    for h in ("kp2d.h", "kp2d_lightglue.h"):
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        for m in re.finditer(r"\b(kp2d_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
            args = m.group(2).strip()
            protos[m.group(1)] = 0 if args in ("", "void") else len(args.split(","))
    return protos


def test_every_documented_call_matches_its_prototype():
    protos = _prototypes()
    assert protos["kp2d_forward"] == 15 and protos["kp2d_post"] == 22
    seen = set()
    for block in _python_blocks():
        try:
            tree = ast.parse(block)
        except SyntaxError:
            continue            # illustrative fragments ("for ...: ...") are not stubs
        for node in ast.walk(tree):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("kp2d_") \
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == "lib":
                name = node.func.attr
                assert name in protos, f"INTEGRATION.md calls {name}, which no header declares"
                assert len(node.args) == protos[name], f"{name}: stub passes {len(node.args)} arguments, header declares {protos[name]}"
                seen.add(name)
    assert {"kp2d_create", "kp2d_set_weight", "kp2d_finalize_weights", "kp2d_workspace_bytes", "kp2d_forward",
            "kp2d_post", "kp2d_lg_forward"} <= seen


@pytest.mark.gpu
def test_stub_executes_and_matches_the_package():
    import torch
    from conftest import product_model
    from oracle.weights import synthetic_frames
    stub = next(b for b in _python_blocks() if "lib.kp2d_forward(" in b)
    model, _ = product_model("S", False, 28)
    model.set_precision("f16x3")       # the library's own default: the stub talks to the C ABI, which reads no KP2D_PRECISION
    x = torch.from_numpy(synthetic_frames(2, 64, 96, seed=5)).to("cuda:0")
    ns = {"model": model}
    cwd = os.getcwd()
    os.chdir(ROOT)              # the stub opens the library by its repo-relative path
    try:
        exec(compile(stub, "INTEGRATION.md", "exec"), ns)
    finally:
        os.chdir(cwd)

    class Self:                 # what the stub reads from the module it is pasted into
        training = False

    with torch.no_grad():
        got = ns["forward"](Self(), x)
        want = model(x)
        for k in ("score", "coord", "feat", "vlad", "seg"):
            assert torch.equal(got[k], want[k]), k
        got_p = ns["post_processing"](Self(), dict(got), 64, 96)
        want_p = model.post_processing(dict(want), 64, 96)
        for k in ("score", "coord", "feat", "seg"):
            assert got_p[k].dtype == want_p[k].dtype and torch.equal(got_p[k], want_p[k]), k
    torch.cuda.synchronize()
    ns["lib"].kp2d_destroy(ns["h"])
    assert np.isfinite(got["vlad"].cpu().numpy()).all()
