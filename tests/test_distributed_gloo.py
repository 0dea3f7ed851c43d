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


def _map_worker(rank, world, port, n_items):
    from riffusion.batch_shard import result_rows, sharded_map

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    calls = []

    def convert(lo, hi):  # stand-in for "decode clips lo..hi on this rank's GPU": (hi - lo, 4, 1) int16 PCM
        calls.append((lo, hi))
        return torch.arange(lo, hi, dtype=torch.int16).reshape(-1, 1, 1).repeat(1, 4, 1)

    full = sharded_map(convert, n_items, dist.group.WORLD, gather="all")
    assert calls == [shard_range(n_items, world, rank)]  # each rank converts only its own slice, once
    calls.clear()
    assert full.shape == (n_items, 4, 1) and torch.equal(full[:, 0, 0], torch.arange(n_items, dtype=torch.int16))
    assert torch.equal(sharded_map(convert, n_items, True, gather="all"), full)  # True = default group
    # gather="none" (the default since round 4: the mode that scales): the own shard only, no collective;
    # gather="rank0": the whole batch on rank 0, own shard elsewhere
    lo, hi = shard_range(n_items, world, rank)
    own = sharded_map(convert, n_items, dist.group.WORLD)
    assert own.shape == (hi - lo, 4, 1) and torch.equal(own[:, 0, 0], torch.arange(lo, hi, dtype=torch.int16))
    assert torch.equal(own, sharded_map(convert, n_items, dist.group.WORLD, gather="none"))
    assert result_rows(n_items, dist.group.WORLD, "none") == (lo, hi) == result_rows(n_items, dist.group.WORLD)
    r0 = sharded_map(convert, n_items, dist.group.WORLD, gather="rank0")
    assert torch.equal(r0, full if rank == 0 else own)
    assert result_rows(n_items, dist.group.WORLD, "rank0") == ((0, n_items) if rank == 0 else (lo, hi))
    assert result_rows(n_items, dist.group.WORLD, "all") == (0, n_items) and result_rows(n_items) == (0, n_items)
    with pytest.raises(ValueError):
        sharded_map(convert, n_items, dist.group.WORLD, gather="everyone")
    # round 5: `gather=None` at an entry point with a multi-rank group warns ONCE per process that the default is the own shard
    import warnings

    from riffusion import batch_shard

    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        assert batch_shard.default_gather(dist.group.WORLD) == "none" and batch_shard.default_gather(True) == "none"
    assert len(seen) == 1 and "own shard" in str(seen[0].message) and issubclass(seen[0].category, UserWarning)
    # a sub-group whose rank 0 is global rank 1: "rank0" means the GROUP's first rank
    sub = dist.new_group([1, 0]) if world == 2 else None
    if sub is not None:
        g = sharded_map(convert, n_items, sub, gather="rank0")
        glo, ghi = shard_range(n_items, 2, dist.get_rank(sub))
        assert g.shape[0] == (n_items if dist.get_rank(sub) == 0 else ghi - glo)
        assert torch.equal(g[:, 0, 0], torch.arange(n_items, dtype=torch.int16) if dist.get_rank(sub) == 0
                           else torch.arange(glo, ghi, dtype=torch.int16))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [1, 7])
def test_sharded_map_two_ranks(n_items):
    """The sharding used by SpectrogramImageConverter.audio_from_spectrogram_images(group=...) and bench.py's
    decode-stereo64 workload: world_size 2 over gloo, including a rank with an empty shard (n_items = 1)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_map_worker, args=(2, port, n_items), nprocs=2, join=True)


def test_sharded_map_without_group_runs_locally():
    from riffusion.batch_shard import sharded_map

    out = sharded_map(lambda lo, hi: torch.arange(lo, hi)[:, None], 5, None)
    assert out.shape == (5, 1)


def test_chunk_sink_on_cpu_and_gather_into_one_tensor():
    """ChunkSink degrades to plain copies without a GPU; an even split is gathered without a compaction copy."""
    from riffusion.batch_shard import ChunkSink

    for to_host in (False, True):
        sink = ChunkSink(5, (3, 2), torch.int16, torch.device("cpu"), to_host=to_host)
        for a in range(0, 5, 2):
            b = min(5, a + 2)
            dst = sink.rows(a, b)
            chunk = torch.full((b - a, 3, 2), a, dtype=torch.int16)
            if dst is not None:
                dst.copy_(chunk)
                chunk = dst
            sink.put(a, b, chunk)
        out = sink.finish()
        assert out.shape == (5, 3, 2) and out[:, 0, 0].tolist() == [0, 0, 2, 2, 4]
    assert ChunkSink(0, (3, 2), torch.int16, torch.device("cpu"), to_host=True).finish().shape == (0, 3, 2)
