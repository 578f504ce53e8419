"""Alias so the reference's import line keeps working (src/visual_odometry/visual_odometry.py:10, ./src on sys.path):

    from visual_odometry.frontend import KP2DtinyFrontend
"""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
if _root not in _sys.path:
    _sys.path.insert(0, _root)

from nano_vs_slam_amd.visual_odometry.frontend import KP2DtinyFrontend  # noqa: E402,F401
