"""Batch sharding across the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" is RCCL).

QPs are independent, so the path shards along the batch axis with NO data-path collective: rank g solves the
contiguous slice ``shard_bounds(B, G, g)`` on its own GPU.  The only optional exchange is an all-gather of the
solved stage-0 ground-reaction forces (12 floats per QP; 384 KiB per rank at B = 65536, G = 8), which on the
fully connected xGMI node is one direct step per peer -- latency-, not bandwidth-bound.
"""
from __future__ import annotations


def shard_bounds(B: int, world: int, rank: int):
    """Contiguous [lo, hi) of the batch owned by `rank`; the first B % world ranks get one extra QP."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    base, extra = divmod(B, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_batch(batch: dict, world: int, rank: int) -> dict:
    B = len(batch["x0"])
    lo, hi = shard_bounds(B, world, rank)
    return {k: v[lo:hi] for k, v in batch.items() if hasattr(v, "__len__") and len(v) == B}


def all_gather_stage0(u_local, B_total: int, group=None):
    """u_local [B_local, N, 12] (any device/backend) -> stage-0 GRFs of the whole batch [B_total, 12], in batch order."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [shard_bounds(B_total, world, g)[1] - shard_bounds(B_total, world, g)[0] for g in range(world)]
    mine = u_local[:, 0, :].contiguous()
    assert mine.shape[0] == sizes[rank], "local batch does not match this rank's shard"
    if len(set(sizes)) == 1:
        out = torch.empty((B_total, 12), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(out, mine, group=group)
        return out
    pad = max(sizes)                                   # ragged shards: pad to the largest, gather, trim
    buf = torch.zeros((pad, 12), dtype=mine.dtype, device=mine.device)
    buf[: sizes[rank]] = mine
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)
