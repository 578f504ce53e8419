"""Multi-process path of sharding.py on CPU: world_size 2, gloo backend, 127.0.0.1 rendezvous."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

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
