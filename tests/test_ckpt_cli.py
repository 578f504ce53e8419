"""The checkpoint wire format and the evaluation CLI's flags (reference utils/utils.py:9-30, eval_multitask.py:35-94,
:150-167): ``load_checkpoint`` contract on CPU; on the GPU a real ``.ckpt`` file goes through
``eval_multitask.py --model_path`` and must produce the outputs of a direct ``load_state_dict``."""
import glob
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT


def _sd(model, seed=1234):
    from oracle.weights import spread_state_dict
    sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed)
    return {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}


def test_load_checkpoint_contract(tmp_path, capsys):
    sys.path.insert(0, ROOT)
    from eval_multitask import load_checkpoint
    sd = {"a.weight": torch.arange(6.0).reshape(2, 3), "a.bias": torch.zeros(2)}
    full = tmp_path / "full.ckpt"
    torch.save({"state_dict": sd, "optimizer": {"lr": 0.1}, "epoch": 7, "config": {"name": "S"}}, full)
    got, opt, info = load_checkpoint(str(full), optimizer_key="optimizer")
    assert set(got) == set(sd) and torch.equal(got["a.weight"], sd["a.weight"])
    assert opt == {"lr": 0.1} and info == {"epoch": 7, "config": {"name": "S"}}
    got, opt, info = load_checkpoint(str(full))                       # no optimizer key asked: it stays in info
    assert opt is None and set(info) == {"optimizer", "epoch", "config"}
    bare = tmp_path / "bare.ckpt"
    torch.save(sd, bare)                                              # a bare state dict: info is None
    got, opt, info = load_checkpoint(str(bare), optimizer_key="optimizer")
    assert info is None and opt is None and set(got) == set(sd)
    assert "optimizer not found" in capsys.readouterr().out
    with pytest.raises(AssertionError):
        load_checkpoint(str(tmp_path / "weights.pth"))
    with pytest.raises(AssertionError):
        load_checkpoint(str(tmp_path / "missing.ckpt"))


class _Payload:
    """A pickled callable: what a hostile .ckpt would carry.  The safe loader must refuse it, not run it."""
    def __reduce__(self):
        return (os.system, ("echo kp2d-ckpt-code-ran > /dev/null",))


def test_checkpoint_with_pickled_code_is_refused(tmp_path):
    sys.path.insert(0, ROOT)
    from eval_multitask import load_checkpoint
    bad = tmp_path / "hostile.ckpt"
    torch.save({"state_dict": {"a": torch.zeros(1)}, "hook": _Payload()}, bad)
    with pytest.raises(RuntimeError, match="safe checkpoint loader"):
        load_checkpoint(str(bad))
    # a training checkpoint as train_multitask.py:553-562 writes it (optimizer state with tensors, epoch, nested dicts)
    good = tmp_path / "train.ckpt"
    opt = {"state": {0: {"step": torch.tensor(3.0), "exp_avg": torch.ones(2, 3)}}, "param_groups": [{"lr": 1e-3, "params": [0]}]}
    torch.save({"epoch": 12, "state_dict": {"a.weight": torch.ones(2, 3)}, "optimizer": opt, "config": {"name": "S", "v3": False},
                "start_results": {"keypoints": [0.1, 0.2]}, "current_results": None}, good)
    sd, o, info = load_checkpoint(str(good), optimizer_key="optimizer")
    assert torch.equal(sd["a.weight"], torch.ones(2, 3)) and o["param_groups"][0]["lr"] == 1e-3 and info["epoch"] == 12


def test_no_product_file_unpickles():
    """Every torch.load of the product passes weights_only=True (reference utils/utils.py:13 uses the unsafe default)."""
    import re
    files = ["eval_multitask.py", "demo.py", "bench.py", "__graft_entry__.py"]
    for base in ("nano-vs-slam_amd", "src", "lightglue", "tools"):
        files += [os.path.relpath(f, ROOT) for f in glob.glob(os.path.join(ROOT, base, "**", "*.py"), recursive=True)]
    for f in files:
        text = open(os.path.join(ROOT, f)).read()
        for m in re.finditer(r"torch\.load\(([^\n]*)", text):
            assert "weights_only=True" in m.group(1), f"{f}: {m.group(0)}"
        assert "pickle.load" not in text and "weights_only=False" not in text, f


def test_cli_accepts_every_reference_flag():
    """Every ``--flag`` the reference's parser defines (eval_multitask.py:35-94) parses here too."""
    sys.path.insert(0, ROOT)
    import eval_multitask
    ref_flags = ["--device", "--model_path", "--dataset_config", "--debug", "--num_workers", "--seed", "--n_classes",
                 "--model_type", "--wandb_project", "--dataset_name", "--config", "--batch_size", "--quantized",
                 "--wandb", "--keypoints", "--visloc", "--segmentation", "--depth", "--load_depth", "--vo", "--backend",
                 "--v3", "--result_dir"]
    argv = ["eval_multitask.py", "--device", "cuda", "--model_path", "x.ckpt", "--dataset_config", "d.json", "--debug",
            "--num_workers", "2", "--seed", "1", "--n_classes", "19", "--model_type", "KeypointNet", "--wandb_project",
            "p", "--dataset_name", "cocostuff", "--config", "S_A", "--batch_size", "2", "--quantized", "--wandb",
            "--keypoints", "--visloc", "--segmentation", "--depth", "--load_depth", "--vo", "--backend", "qnnpack",
            "--v3", "--result_dir", "r"]
    assert all(f in argv for f in ref_flags)
    old = sys.argv
    sys.argv = argv
    try:
        a = eval_multitask.parse_args()
    finally:
        sys.argv = old
    assert a.load_depth and a.quantized and a.wandb and a.backend == "qnnpack" and a.config == "S_A" and a.v3


@pytest.mark.gpu
@pytest.mark.parametrize("config,v3,ncls,depth", [("S", False, 28, False), ("S_A", True, 19, True)])
def test_ckpt_through_the_cli_equals_direct_load(tmp_path, config, v3, ncls, depth):
    from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import KP2DTinyV2, KP2DTinyV3, get_config
    model = (KP2DTinyV3 if v3 else KP2DTinyV2)(**get_config(config, v3=v3), nClasses=ncls, depth=depth)
    sd = _sd(model, seed=77)                                           # NOT the CLI's default stand-in weights
    sd["backbone.conv1a.bn.num_batches_tracked"] = torch.tensor(123)  # real checkpoints carry these
    ckpt = tmp_path / "model.ckpt"
    torch.save({"state_dict": sd, "optimizer": {"state": {}, "param_groups": []}, "epoch": 12,
                "config": {"name": config}}, ckpt)
    cmd = [sys.executable, "eval_multitask.py", "--model_path", str(ckpt), "--config", config, "--n_classes", str(ncls),
           "--keypoints", "--n_batches", "1", "--batch_size", "2", "--result_dir", str(tmp_path / "res"),
           "--quantized", "--backend", "qnnpack", "--wandb", "--seed", "5"]
    cmd += (["--v3"] if v3 else []) + (["--load_depth"] if depth else [])
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "--quantized" in r.stdout and "--wandb" in r.stdout and "Error loading" not in r.stdout
    res = json.load(open(glob.glob(str(tmp_path / "res" / "*.json"))[0]))
    assert res["checkpoint_info"]["epoch"] == "12" and "optimizer" not in res["checkpoint_info"]
    # the same weights loaded directly
    model.load_state_dict(sd, strict=True)
    model = model.to("cuda:0").eval()
    model.training = False
    probe = torch.from_numpy(np.random.default_rng(5).random((1, 3, 64, 96), np.float32) * 2 - 1).to("cuda:0")
    with torch.no_grad():
        out = model(probe)
    assert set(res["probe"]) == set(out)
    assert ("depth" in out) == depth
    for k, v in out.items():
        assert res["probe"][k] == [float(v.double().sum()), float(v.double().abs().max())], k
