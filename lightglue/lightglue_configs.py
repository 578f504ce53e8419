"""Alias of nano-vs-slam_amd/lightglue/lightglue_configs.py (reference import: visual_odometry.py:9)."""
from nano_vs_slam_amd.lightglue.lightglue_configs import LIGHT_GLUE_CONFIGS, get_light_glue_config  # noqa: F401
