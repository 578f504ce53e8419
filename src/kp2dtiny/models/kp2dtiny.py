"""Alias so the reference's import lines keep working unchanged:

    from src.kp2dtiny.models.kp2dtiny import tiny_factory      # demo.py:1, eval_multitask.py:24, tests.py:1
    from kp2dtiny.models.kp2dtiny import KP2DTinyV2            # visual_odometry/frontend.py:5 (./src on sys.path)

Everything resolves to the MI355X-native implementation in ``nano-vs-slam_amd/kp2dtiny/models/kp2dtiny.py``.
"""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import *  # noqa: E402,F401,F403
from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import (KP2DTINY_CONFIGS, KP2DTINYV3_CONFIGS, KP2DTinyV2,  # noqa: E402,F401
                                                        KP2DTinyV3, get_config, tiny_factory)
