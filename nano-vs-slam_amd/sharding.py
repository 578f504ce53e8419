"""Frame-batch sharding over the GPUs of one MI355X node — new work, the reference is
single-process / single-device.

One process per GPU (torchrun style).  Frames are independent (BatchNorm in eval mode, per-frame NetVLAD
and attention), so rank r of R owns the contiguous frame range ``shard_range(B, r, R)`` and the steady
state has NO collective.  The only exchange is one broadcast of the packed weight blob (<= 4 MB) from
rank 0 at start-up — RCCL over xGMI when the backend is "nccl"; the same code runs on "gloo" with CPU
tensors for the world_size-2 tests.  ``gather_vlad`` is the optional all-gather of the 16 KB/frame
place-recognition vectors for callers that want the whole VPR matrix on every rank.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced partition: the first (n_frames % world) ranks get one extra frame."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_frames, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def broadcast_blob(blob: torch.Tensor | None, nbytes: int, device, src: int = 0, group=None) -> torch.Tensor:
    """Broadcast a uint8 blob of ``nbytes`` from ``src``; other ranks pass ``None`` and get a fresh tensor."""
    if dist.get_rank(group) == src:
        if blob is None or blob.numel() != nbytes:
            raise ValueError("source rank must provide the blob")
        buf = blob.to(device)
    else:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return buf


def broadcast_model_weights(model, device, src: int = 0, group=None) -> int:
    """Rank ``src`` packs its weights on the device; every other rank imports the broadcast blob, so only one
    rank needs the checkpoint (demo.py / eval_multitask.py load it once).  Returns the blob's size in bytes."""
    rank = dist.get_rank(group)
    size = torch.zeros(1, dtype=torch.int64, device=device)
    blob = None
    if rank == src:
        blob = model.packed_weights(device)
        size[0] = blob.numel()
    dist.broadcast(size, src=src, group=group)
    buf = broadcast_blob(blob, int(size.item()), device, src, group)
    if rank != src:
        model.load_packed_weights(buf)
    return int(buf.numel() * buf.element_size())


def gather_vlad(vlad_local: torch.Tensor, n_frames: int, group=None) -> torch.Tensor:
    """All-gather per-rank VLAD rows [b_r, D] into [n_frames, D] in frame order (ragged shards allowed)."""
    world = dist.get_world_size(group)
    D = vlad_local.shape[1]
    base = (n_frames + world - 1) // world
    pad = torch.zeros(base, D, dtype=vlad_local.dtype, device=vlad_local.device)
    pad[: vlad_local.shape[0]] = vlad_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    rows = []
    for r in range(world):
        s, e = shard_range(n_frames, r, world)
        rows.append(parts[r][: e - s])
    return torch.cat(rows, 0)
