"""Multi-process path of sharding.py on CPU: world_size 2, gloo backend, 127.0.0.1 rendezvous."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT
from nano_vs_slam_amd.sharding import broadcast_blob, gather_vlad, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_frames, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # (1) weight-blob broadcast: only rank 0 holds the bytes
        blob = torch.arange(4099, dtype=torch.int64).to(torch.uint8) if rank == 0 else None
        got = broadcast_blob(blob, 4099, "cpu", src=0)
        ok_blob = bool(torch.equal(got, torch.arange(4099, dtype=torch.int64).to(torch.uint8)))
        # (2) each rank "infers" only its own frames; no collective in the data path
        s, e = shard_range(n_frames, rank, world)
        full = torch.from_numpy(np.random.default_rng(5).standard_normal((n_frames, 8)).astype(np.float32))
        local = full[s:e] * 2.0
        # (3) optional all-gather of the per-frame VLAD rows, ragged shards
        allv = gather_vlad(local, n_frames)
        ok_gather = bool(torch.equal(allv, full * 2.0))
        ret[rank] = (ok_blob, ok_gather, e - s)
    finally:
        dist.destroy_process_group()


def _run(n_frames, world=2):
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, n_frames, ret), nprocs=world, join=True)
    return dict(ret)


def test_broadcast_and_gather_world2_even():
    r = _run(8)
    assert r[0][:2] == (True, True) and r[1][:2] == (True, True)
    assert r[0][2] + r[1][2] == 8


def test_broadcast_and_gather_world2_ragged():
    r = _run(7)
    assert all(v[0] and v[1] for v in r.values())
    assert sorted(v[2] for v in r.values()) == [3, 4]


class _FakeModel:
    """Duck-typed stand-in for the model's packed-weight exchange (packed_weights / load_packed_weights), so that
    ``broadcast_model_weights`` itself — the entry point bench.py and the entry scripts call — runs on CPU ranks."""

    def __init__(self, rank):
        self.blob = (torch.arange(70001, dtype=torch.int64) * 7 % 251).to(torch.uint8) if rank == 0 else None
        self.loaded = None

    def packed_weights(self, device):
        return self.blob.to(device)

    def load_packed_weights(self, buf):
        self.loaded = buf.clone()


def _worker_model(rank, world, port, ret):
    from nano_vs_slam_amd.sharding import broadcast_model_weights
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m = _FakeModel(rank)
        broadcast_model_weights(m, "cpu", src=0)
        want = (torch.arange(70001, dtype=torch.int64) * 7 % 251).to(torch.uint8)
        ret[rank] = (m.loaded is None) if rank == 0 else bool(torch.equal(m.loaded, want))
    finally:
        dist.destroy_process_group()


def test_broadcast_model_weights_world2():
    """Only rank 0 owns weights; rank 1 imports exactly the broadcast bytes; rank 0 does not re-import its own."""
    port = _free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker_model, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret) == {0: True, 1: True}


def test_shard_range_covers_cfg3():
    """BASELINE cfg 3: 256 frames over 8 GPUs = 32 contiguous frames per rank; ragged totals stay contiguous."""
    assert [shard_range(256, r, 8) for r in range(8)] == [(32 * r, 32 * r + 32) for r in range(8)]
    for n, w in [(7, 2), (100, 8), (5, 8)]:
        spans = [shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_bench_refuses_world_size_mismatch():
    """bench.py's n_gpus must be what was asked for AND what ran: WORLD_SIZE != --gpus exits non-zero before any GPU work."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=1 but --gpus 2" in (r.stderr + r.stdout)
    env["WORLD_SIZE"] = "4"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=4 but --gpus 8" in (r.stderr + r.stdout)


def test_bench_launches_its_own_ranks_when_no_launcher_is_around():
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (the way the driver runs --gpus 1) must start N
    ranks by itself: the parent spawns fresh children before anything touches a device, relays rank 0's one JSON line
    and the worst child code.  --dry-run = rendezvous only (this box has no GPU); the measured twin of this test is
    tests/test_multi_rank_gpu.py::test_bench_self_launch_two_ranks."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 2 and j["collective"]["ranks_seen"] == 2
    assert j["collective"]["ranks"] == [0, 1] and j["launcher"].startswith("self")


def test_bench_eight_rank_launch_shards_256_frames_and_caps_host_threads():
    """BASELINE cfg 3 as the driver would start it on an 8-GPU node — `python bench.py --gpus 8 --global-batch 256` — through
    --dry-run (no GPU here): eight children, ONE line, every rank seen, contiguous shards of 32 frames that tile the batch,
    and every rank's host thread pool capped at cores // 8 (eight Python ranks share the node's cores)."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "OMP_NUM_THREADS")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--global-batch", "256", "--dry-run"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == 8 and j["collective"]["ranks_seen"] == 8
    assert j["collective"]["ranks"] == list(range(8)) and j["global_batch"] == 256
    assert j["frame_shards"] == [[32 * i, 32 * (i + 1)] for i in range(8)]
    cap = max(1, j["host_cores"] // 8)
    assert all(1 <= t <= cap for t in j["host_threads_per_rank"]), (j["host_threads_per_rank"], cap)


def test_bench_self_launch_reports_a_failing_rank():
    """No device here: the real (non-dry) run must fail in every child and the parent must return non-zero, not hang."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "needs a HIP device" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
