"""Two ranks on ONE GPU (both on cuda:0, gloo control plane): the real multi-GPU entry point
``sharding.broadcast_model_weights`` with the HIP engine behind it, and bench.py's N > 1 path.

Only rank 0 ever sees the state dict; rank 1 builds the same configuration, imports the broadcast packed blob and must
produce bit-identical outputs on the same frames.  (RCCL itself needs one GPU per rank and is exercised by the driver's
8-GPU run; the code path is the same apart from the backend string.)
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

_WORKER = r'''
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["KP2D_ROOT"])
from nano_vs_slam_amd.kp2dtiny.models.kp2dtiny import tiny_factory
from nano_vs_slam_amd.sharding import broadcast_model_weights, shard_range, gather_vlad
from nano_vs_slam_amd.synthetic import spread_state_dict, synthetic_frames
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
dev = torch.device("cuda", 0)
model = tiny_factory("S", 28)
if rank == 0:                                   # the only rank with a "checkpoint"
    sd = spread_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()})
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
model = model.to(dev).eval(); model.training = False
broadcast_model_weights(model, dev, src=0)
frames = torch.from_numpy(synthetic_frames(6, 64, 96, seed=3)).to(dev)
with torch.no_grad():
    full = model(frames)                        # every rank runs ALL frames: outputs must agree across ranks
    lo, hi = shard_range(6, rank, world)
    mine = model(frames[lo:hi].contiguous())    # and its own shard
    for k in full:
        assert torch.equal(full[k][lo:hi], mine[k]), k
    allv = gather_vlad(mine["vlad"].cpu(), 6)
    assert torch.equal(allv, full["vlad"].cpu())
np.savez(os.path.join(os.environ["KP2D_OUT"], f"rank{rank}.npz"), **{k: v.cpu().numpy() for k, v in full.items()})
dist.barrier(); dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _spawn(cmds_env, timeout=600):
    procs = [subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for cmd, env in cmds_env]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    return outs


def test_rank1_matches_rank0_after_weight_broadcast(tmp_path):
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", KP2D_ROOT=ROOT,
                KP2D_OUT=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = _spawn([([sys.executable, "-c", _WORKER], dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)])
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    a, b = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert sorted(a.files) == ["coord", "feat", "score", "seg", "vlad"]
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert np.isfinite(a["vlad"]).all() and np.abs(a["score"]).max() > 0


def test_bench_two_ranks_global_batch():
    """bench.py --gpus 2 --global-batch 12 (gloo rehearsal on one GPU): shards of 6, one JSON line from rank 0."""
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--global-batch", "12", "--steps", "3", "--warmup", "1",
           "--height", "64", "--width", "96", "--backend", "gloo", "--profile-steps", "1"]
    outs = _spawn([(cmd, dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)])
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    lines = [ln for ln in outs[0][1].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][1].splitlines() if ln.startswith("{")]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 12 and j["config"]["frame_shards"] == 2
    assert j["scaling"] == "weak" and j["value"] > 0 and set(j["precision_modes"]) == {"f16x3", "fp32"}
    assert abs(j["value"] - 12 * 3 / (j["ms_per_step"] * 3e-3)) / j["value"] < 0.01
    # what the driver's 8-GPU run needs to prove about itself: backend, ranks the collective saw, the one broadcast
    c = j["collective"]
    assert c["backend"] == "gloo" and c["ranks_seen"] == 2 and c["steady_state_collectives"] == 0
    assert 1_000_000 < c["broadcast_bytes"] < 64_000_000 and c["broadcast_ms"] > 0
    pr = j["per_rank_frames_per_s"]
    assert 0 < pr["min"] <= pr["max"] and pr["min"] * 2 >= j["value"] * 0.5


def test_bench_cfg3_shape_two_rank_rehearsal():
    """BASELINE cfg 3 is `--gpus 8 --global-batch 256` (32 frames per GPU at 240x320); rehearsed here with the same
    per-rank share on two ranks of one GPU: `--gpus 2 --global-batch 64`."""
    port = _free_port()
    base = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--global-batch", "64", "--steps", "3", "--warmup", "1",
           "--backend", "gloo", "--profile-steps", "1", "--no-precision-modes"]
    outs = _spawn([(cmd, dict(base, RANK=str(r), LOCAL_RANK=str(r))) for r in range(2)])
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    j = json.loads([ln for ln in outs[0][1].splitlines() if ln.startswith("{")][0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 64 and "batch 32/GPU" in j["config"]["workload"]
    assert j["collective"]["ranks_seen"] == 2


def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r.stderr + r.stdout)


def test_bench_self_launch_two_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE: bench.py starts its two ranks itself (both on
    this box's one GPU, gloo control plane), one JSON line, ranks_seen == 2 — the command shape the driver uses."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--global-batch", "12", "--steps", "3", "--warmup", "1",
                        "--height", "64", "--width", "96", "--backend", "gloo", "--profile-steps", "1", "--no-precision-modes"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["collective"]["ranks_seen"] == 2 and j["launcher"].startswith("self")
    assert j["config"]["global_batch"] == 12 and j["value"] > 0
