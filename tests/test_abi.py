"""The C-ABI shared library: loads on a CPU-only box and exports exactly what include/kp2d.h declares.
No compute call is made here (there is no GPU); compute goes through tests marked ``gpu``."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def lib_path():
    import __graft_entry__ as g
    return g._load_build_module().build(verbose=False)


def header_symbols():
    syms = set()
    for name in ("kp2d.h", "kp2d_lightglue.h"):
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(kp2d_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_header_and_binding_agree():
    from nano_vs_slam_amd import _lib
    assert header_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(lib_path):
    dll = ctypes.CDLL(lib_path)
    for sym in header_symbols():
        assert hasattr(dll, sym), sym


def test_config_struct_layout_matches_header():
    from nano_vs_slam_amd import _lib
    # struct_size, version, channel_dims[6], 8 scalars, device, global_descriptor, remove_netvlad, depth, upscale_method,
    # in_channels
    assert ctypes.sizeof(_lib.Kp2dConfig) == 22 * 4
    text = open(os.path.join(ROOT, "include", "kp2d.h")).read()
    body = text[text.index("typedef struct kp2d_config {"):text.index("} kp2d_config;")]
    names = re.findall(r"int32_t\s+([a-z_0-9]+)", body)
    assert names == [f[0] for f in _lib.Kp2dConfig._fields_]
    text = open(os.path.join(ROOT, "include", "kp2d_lightglue.h")).read()
    body = text[text.index("typedef struct kp2d_lg_config {"):text.index("} kp2d_lg_config;")]
    assert re.findall(r"int32_t\s+([a-z_0-9]+)", body) == [f[0] for f in _lib.Kp2dLgConfig._fields_]


def test_create_without_gpu_fails_loudly(lib_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from nano_vs_slam_amd import _lib
    lib = _lib.load()
    cfg = _lib.Kp2dConfig()
    cfg.struct_size = ctypes.sizeof(cfg)
    cfg.version = 2
    handle = ctypes.c_void_p()
    rc = lib.kp2d_create(ctypes.byref(cfg), ctypes.byref(handle))
    assert rc == -6 and b"no CPU path" in lib.kp2d_last_error()
    with pytest.raises(_lib.Kp2dError):
        _lib.check(rc)


def test_missing_library_is_an_error(monkeypatch, tmp_path):
    from nano_vs_slam_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "absent.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()
