"""The fixture recipe (oracle/make_golden.py) still reproduces the committed fixtures from the REFERENCE.

Guards the failure VERDICT r1 (weak #1) found: with the repo root ahead of /root/reference on ``sys.path`` the recipe's
``from src.kp2dtiny.models.kp2dtiny import ...`` resolved to this repo's import alias, i.e. it would have pinned the
product against itself.  Runs only where the reference exists (the build container); the GPU box never has it.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

REFERENCE = "/root/reference"
needs_reference = pytest.mark.skipif(not os.path.isdir(os.path.join(REFERENCE, "src", "kp2dtiny")),
                                     reason="the reference tree only exists in the build container")


def _run_recipe(args, cwd, extra_env=None):
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    env.update(extra_env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "make_golden.py")] + args, cwd=cwd, env=env,
                          capture_output=True, text=True, timeout=600)


@needs_reference
@pytest.mark.parametrize("name", ["v2_N_32x48_taps", "v3_SA_32x48_taps"])
def test_recipe_regenerates_committed_fixture_bit_identically(name, tmp_path):
    # worst case on purpose: run from the repo root with the repo root on PYTHONPATH (both used to shadow the reference)
    r = _run_recipe(["--only", name, "--out", str(tmp_path)], cwd=ROOT, extra_env={"PYTHONPATH": ROOT})
    assert r.returncode == 0, r.stderr[-2000:]
    new = np.load(tmp_path / (name + ".npz"))
    old = np.load(os.path.join(GOLDEN, name + ".npz"))
    assert sorted(new.files) == sorted(old.files)
    for k in old.files:
        if k == "meta":
            continue
        assert np.array_equal(new[k], old[k]), k


@needs_reference
def test_recipe_refuses_a_model_module_outside_the_reference(tmp_path):
    """reference_module() must raise when ``src.kp2dtiny.models.kp2dtiny`` is not the reference's file."""
    code = (
        "import importlib.util, sys, types\n"
        f"spec = importlib.util.spec_from_file_location('mg', {os.path.join(ROOT, 'oracle', 'make_golden.py')!r})\n"
        "mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)\n"
        "for n in ('src', 'src.kp2dtiny', 'src.kp2dtiny.models'):\n"
        "    m = types.ModuleType(n); m.__path__ = []; sys.modules[n] = m\n"
        "fake = types.ModuleType('src.kp2dtiny.models.kp2dtiny'); fake.__file__ = '/tmp/not_the_reference.py'\n"
        "sys.modules['src.kp2dtiny.models.kp2dtiny'] = fake; sys.modules['src.kp2dtiny.models'].kp2dtiny = fake\n"
        "try:\n"
        "    mg.reference_module()\n"
        "except RuntimeError as e:\n"
        "    print('REFUSED', e); sys.exit(0)\n"
        "sys.exit(3)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=str(tmp_path), capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0 and "REFUSED" in r.stdout, (r.stdout, r.stderr[-1500:])


def test_oracle_weights_module_leaves_sys_path_alone():
    before = list(sys.path)
    import importlib
    import oracle.weights as w
    importlib.reload(w)
    assert sys.path == before
    assert callable(w.spread_state_dict) and callable(w.synthetic_frames)
