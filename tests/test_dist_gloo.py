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
