"""
Sharding of independent clips over the GPUs of a node (one process per GPU, torch.distributed).

Clips never depend on each other: every rank converts its contiguous slice of the
batch with no data-path collective; the only collective is an optional all_gather of the int16 PCM
when one consumer needs the whole batch.  Backend "nccl" is RCCL on ROCm; the CPU tests run the same
code over "gloo".
"""
import typing as T

import torch
import torch.distributed as dist


def shard_range(n_items: int, world_size: int, rank: int) -> T.Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n_items for `rank` (first n_items % world ranks get one more)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_clips(local: torch.Tensor, n_items: int, group: T.Optional[T.Any] = None) -> torch.Tensor:
    """all_gather of per-rank results (clips on dim 0, ragged by at most one) into the full batch, rank order."""
    world = dist.get_world_size(group)
    if world == 1:
        return local
    sizes = [shard_range(n_items, world, r) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((longest,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    # moved as raw bytes: every backend carries uint8, not every backend carries int16
    raw = pad.contiguous().view(torch.uint8).reshape(-1)
    parts = [torch.empty_like(raw) for _ in range(world)]
    dist.all_gather(parts, raw, group=group)
    parts = [p.view(local.dtype).reshape(pad.shape) for p in parts]
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)
