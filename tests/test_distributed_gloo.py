"""world_size-2 CPU (gloo) test of the clip sharding used by the multi-GPU path (bench.py --gpus N)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from riffusion.batch_shard import gather_clips, shard_range


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, n_items):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_items, world, rank)
    # stand-in for the per-rank conversion: clip i -> a (3, 2) int16 block filled with i
    local = torch.stack([torch.full((3, 2), i, dtype=torch.int16) for i in range(lo, hi)]) if hi > lo else torch.zeros((0, 3, 2), dtype=torch.int16)
    full = gather_clips(local, n_items)
    assert full.shape == (n_items, 3, 2)
    assert torch.equal(full[:, 0, 0], torch.arange(n_items, dtype=torch.int16))
    # timing reduction of bench.py: max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [5, 8])
def test_two_rank_gather(n_items):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_items), nprocs=2, join=True)
