"""MI355X-native kp2dtiny multi-task inference path (drop-in for ETH-PBL/Nano-VS-SLAM's KP2DTiny model API).

Import as ``nano_vs_slam_amd`` (the directory name carries a hyphen; ``nano_vs_slam_amd/`` is a shim).
"""
__version__ = "0.1.0"
