"""Alias so the reference's import line keeps working unchanged (src/visual_odometry/visual_odometry.py:8):

    from lightglue.lightglue import LightGlue
"""
from nano_vs_slam_amd.lightglue.lightglue import LightGlue, __main_model__  # noqa: F401
