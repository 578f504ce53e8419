"""Seeded "spread" weights / frames for parity work (TEST INFRASTRUCTURE).

The generators themselves live in ``nano-vs-slam_amd/synthetic.py`` (pure numpy; the entry points and the benchmark use
them too and must not import ``oracle/``); this module keeps the names the tests and ``make_golden.py`` import.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from nano_vs_slam_amd.synthetic import (seeded_linear_state_dict, spread_state_dict, spread_tensor,  # noqa: E402,F401
                                         synthetic_frames)

__all__ = ["spread_state_dict", "synthetic_frames", "spread_tensor", "seeded_linear_state_dict"]
