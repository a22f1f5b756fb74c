"""Image-sharded inference across the GPUs of one node.

Captioning shards by image with no data-path collective: the reference generates captions one
image at a time (TransformerTest.py:126-164, seq2seqbasic.py:155), so any partition of the image list
gives identical captions.  Each rank (one process per GPU) takes a contiguous slice; the token ids are
gathered on the host at the end.  ``gather_captions`` is the only communication and moves a few bytes
per caption (``torch.distributed`` object gather: RCCL/xGMI is not needed for it, gloo is enough).
"""
from __future__ import annotations

import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous [lo, hi) slice of ``n_items`` for ``rank``; the first ``n_items % world`` ranks get one extra."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_captions(local_captions: list, group=None) -> list | None:
    """Concatenate every rank's caption list in rank order on rank 0 (None elsewhere)."""
    if not dist.is_available() or not dist.is_initialized():
        return list(local_captions)
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    out = [None] * world if rank == 0 else None
    dist.gather_object(list(local_captions), out, dst=0, group=group)
    if rank != 0:
        return None
    return [c for part in out for c in part]
