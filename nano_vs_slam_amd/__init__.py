"""Import shim: the product package lives in the directory ``nano-vs-slam_amd/`` (a hyphen is not a
valid Python identifier), so ``import nano_vs_slam_amd`` resolves to it through this stub."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nano-vs-slam_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
