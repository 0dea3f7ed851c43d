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


def sharded_map(
    convert: T.Callable[[int, int], torch.Tensor], n_items: int, group: T.Any = None
) -> torch.Tensor:
    """
    The multi-GPU form of a batch call: rank r runs `convert(lo, hi)` on its slice [lo, hi) of the
    clips (it must return a tensor with hi - lo rows, also when the slice is empty) and the per-rank
    results are all_gathered into the full batch on every rank.  `group=None` runs the whole batch
    locally (no process group needed); `group=True` means the default group.
    """
    if group is None:
        return convert(0, n_items)
    pg = None if group is True else group
    lo, hi = shard_range(n_items, dist.get_world_size(pg), dist.get_rank(pg))
    local = convert(lo, hi)
    if local.shape[0] != hi - lo:
        raise ValueError(f"convert({lo}, {hi}) returned {local.shape[0]} rows")
    return gather_clips(local, n_items, pg)
