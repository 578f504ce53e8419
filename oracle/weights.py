"""Seeded "spread" weights / frames for parity work (TEST INFRASTRUCTURE).

The generators themselves live in ``nano-vs-slam_amd/synthetic.py`` (pure numpy; the entry points and the benchmark use
them too and must not import ``oracle/``); this module keeps the names the tests and ``make_golden.py`` import.

The generator file is loaded BY PATH: importing it through the ``nano_vs_slam_amd`` package would need the repo root on
``sys.path``, and the repo root carries import aliases (``src/``, ``lightglue/``) that shadow the reference's packages
of the same name — which is how ``make_golden.py`` once came to "pin" the product against itself.
This module therefore never touches ``sys.path``.
"""
import importlib.util as _ilu
import os as _os

_path = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "nano-vs-slam_amd", "synthetic.py")
_spec = _ilu.spec_from_file_location("_kp2d_synthetic_for_oracle", _path)
_mod = _ilu.module_from_spec(_spec)
_spec.loader.exec_module(_mod)

seeded_linear_state_dict = _mod.seeded_linear_state_dict
spread_state_dict = _mod.spread_state_dict
spread_tensor = _mod.spread_tensor
synthetic_frames = _mod.synthetic_frames
trained_like_tensor = _mod.trained_like_tensor
trained_like_state_dict = _mod.trained_like_state_dict


def state_dict_for(recipe, shapes, seed=1234, head_gain=1.0):
    """Weights of a fixture's recipe: "spread" (default) or "trained" (fixture meta key "weights")."""
    if recipe == "trained":
        return trained_like_state_dict(shapes, seed=seed)
    return spread_state_dict(shapes, seed=seed, head_gain=head_gain)

__all__ = ["spread_state_dict", "synthetic_frames", "spread_tensor", "seeded_linear_state_dict", "trained_like_tensor",
           "trained_like_state_dict", "state_dict_for"]
