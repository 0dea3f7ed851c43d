"""
Sharding of independent clips over the GPUs of a node (one process per GPU, torch.distributed).

Clips never depend on each other: every rank converts its contiguous slice of the
batch with no data-path collective.  What happens to the per-rank results afterwards is the
caller's choice (`gather=`):

    "none"   nothing: every rank keeps (and returns) its own shard - the reference's consumers want
             host audio for THEIR clips (server.py:159-183, cli.py:172-204 writes one file per clip)
    "rank0"  one `gather` to the group's rank 0 (one consumer needs the whole batch); the other
             ranks keep their own shard
    "all"    one `all_gather_into_tensor`: every rank ends up with the whole batch

Both collectives move the shards into ONE preallocated tensor (no list of parts + cat).  Backend
"nccl" is RCCL on ROCm; the CPU tests run the same code over "gloo".

`ChunkSource` / `ChunkSink` are the other half of a scalable batch call: host tiles are staged through pinned memory and
uploaded on a side stream, chunk k+1 while chunk k computes (the reference's API is host images in, host audio out:
spectrogram_image_converter.py:65-91); a rank's shard is produced chunk by chunk (bounded working set), and each finished
chunk is copied to a pinned host buffer on a side stream while the next chunk computes: both copies are off the critical path.
"""
import typing as T
import warnings

import torch
import torch.distributed as dist

GATHER_MODES = ("all", "rank0", "none")


_warned_default_gather = False


def default_gather(group: T.Any) -> str:
    """`gather=None` at the entry points.  Without a process group the question does not arise.  With one, the default is the
    mode that scales ("none": every rank returns its OWN clips, no data-path collective) - which is NOT what rounds 2-3 of this
    library did ("all": every rank got the whole batch), so the first such call of a process says so once: a caller that relied
    on the old default gets a different leading dimension, and should pass `gather="all"` (or "rank0") explicitly."""
    global _warned_default_gather
    if group is not None and not _warned_default_gather:
        pg = _resolve_group(group)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(pg) > 1:
            _warned_default_gather = True
            warnings.warn(
                "a process group was passed without `gather=`: every rank returns only its own shard of the batch "
                "(gather=\"none\"); pass gather=\"all\" or \"rank0\" for the whole batch (the default of earlier versions was \"all\")",
                stacklevel=3,
            )
    return "none"


def shard_range(n_items: int, world_size: int, rank: int) -> T.Tuple[int, int]:
    """Contiguous, balanced slice [lo, hi) of n_items for `rank` (first n_items % world ranks get one more)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank out of range")
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _resolve_group(group: T.Any) -> T.Any:
    """`True` means the default group."""
    return None if group is True else group


def gather_clips(
    local: torch.Tensor, n_items: int, group: T.Optional[T.Any] = None, dst: T.Optional[int] = None
) -> T.Optional[torch.Tensor]:
    """
    Per-rank results (clips on dim 0, ragged by at most one) -> the full batch in rank order.

    dst=None: all_gather, every rank returns the full batch.  dst=r (group rank): gather, only rank r
    returns it (the others get None).  The shards land in one preallocated tensor; when the clips
    divide evenly over the ranks that tensor IS the result (no further copy).
    """
    world = dist.get_world_size(group)
    if world == 1:
        return local
    rank = dist.get_rank(group)
    sizes = [shard_range(n_items, world, r) for r in range(world)]
    longest = max(hi - lo for lo, hi in sizes)
    row_shape = tuple(local.shape[1:])
    if local.shape[0] == longest and local.is_contiguous():
        mine = local
    else:
        mine = torch.zeros((longest,) + row_shape, dtype=local.dtype, device=local.device)
        mine[: local.shape[0]] = local
    # moved as raw bytes: every backend carries uint8, RCCL has no int16
    raw = mine.view(torch.uint8).reshape(-1)
    receiver = dst is None or rank == dst
    full = torch.empty((world * longest,) + row_shape, dtype=local.dtype, device=local.device) if receiver else None
    if dst is None:
        dist.all_gather_into_tensor(full.view(torch.uint8).reshape(-1), raw, group=group)
    else:
        parts = list(full.view(torch.uint8).reshape(world, -1).unbind(0)) if receiver else None
        dist.gather(raw, parts, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
    if not receiver:
        return None
    if all(hi - lo == longest for lo, hi in sizes):
        return full
    rows = full.reshape((world, longest) + row_shape)
    return torch.cat([rows[r, : hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)


def sharded_map(
    convert: T.Callable[[int, int], torch.Tensor], n_items: int, group: T.Any = None, gather: str = "none"
) -> torch.Tensor:
    """
    The multi-GPU form of a batch call: rank r runs `convert(lo, hi)` on its slice [lo, hi) of the
    clips (it must return a tensor with hi - lo rows, also when the slice is empty); `gather` says
    who receives what (module docstring).  `group=None` runs the whole batch locally (no process
    group needed); `group=True` means the default group.
    """
    if gather not in GATHER_MODES:
        raise ValueError(f"gather must be one of {GATHER_MODES}, got {gather!r}")
    if group is None:
        return convert(0, n_items)
    pg = _resolve_group(group)
    lo, hi = shard_range(n_items, dist.get_world_size(pg), dist.get_rank(pg))
    local = convert(lo, hi)
    if local.shape[0] != hi - lo:
        raise ValueError(f"convert({lo}, {hi}) returned {local.shape[0]} rows")
    if gather == "none":
        return local
    full = gather_clips(local, n_items, pg, dst=None if gather == "all" else 0)
    return local if full is None else full


def result_rows(n_items: int, group: T.Any = None, gather: str = "none") -> T.Tuple[int, int]:
    """Which clips [lo, hi) of the batch a `sharded_map(..., group, gather)` call returns ON THIS RANK."""
    if group is None:
        return 0, n_items
    pg = _resolve_group(group)
    world, rank = dist.get_world_size(pg), dist.get_rank(pg)
    if gather == "all" or (gather == "rank0" and rank == 0):
        return 0, n_items
    return shard_range(n_items, world, rank)


class ChunkSink:
    """
    Destination of a shard that is produced chunk by chunk.

    to_host=False: one preallocated device tensor; `put` copies a chunk into its rows (stream order).
    to_host=True:  one pinned host tensor; `put` queues the chunk's device-to-host copy on a SIDE
                   stream behind an event on the compute stream, so the copy of chunk k runs while
                   chunk k+1 computes; `finish` waits for the side stream.  The pinned block comes
                   from torch's caching host allocator (first call pays the registration, later calls
                   reuse it) and is owned by the returned tensor - a fresh object per call, as the
                   reference's callers expect.
    On a machine without a GPU (the CPU tests) both forms degrade to plain copies.
    """

    def __init__(self, rows: int, row_shape: T.Sequence[int], dtype: torch.dtype, device: torch.device, to_host: bool):
        self.to_host = to_host
        self.cuda = torch.device(device).type == "cuda"
        shape = (rows,) + tuple(row_shape)
        if to_host:
            self.out = torch.empty(shape, dtype=dtype, pin_memory=self.cuda)
            self.side = None  # side stream, made when the first of several chunks arrives
        else:
            self.out = torch.empty(shape, dtype=dtype, device=device)
            self.side = None
        self.device = device

    def rows(self, a: int, b: int) -> T.Optional[torch.Tensor]:
        """Device view a producer may write rows [a, b) into directly (None when the sink is on the host)."""
        return None if self.to_host else self.out[a:b]

    def put(self, a: int, b: int, chunk: torch.Tensor) -> None:
        if not self.to_host:
            if chunk.data_ptr() != self.out[a:b].data_ptr():
                self.out[a:b].copy_(chunk)
            return
        if not self.cuda:
            self.out[a:b].copy_(chunk)
            return
        if a == 0 and b == self.out.shape[0]:
            # the whole shard in one chunk (one tile per request, server.py:152-164): nothing to overlap with, so the copy
            # stays on the compute stream - no event, no second stream on the latency path
            self.out.copy_(chunk, non_blocking=True)
            self.side = torch.cuda.current_stream(self.device)
            return
        done = torch.cuda.Event()
        done.record(torch.cuda.current_stream(self.device))
        if self.side is None:
            self.side = torch.cuda.Stream(self.device)
        self.side.wait_event(done)
        with torch.cuda.stream(self.side):
            self.out[a:b].copy_(chunk, non_blocking=True)
        chunk.record_stream(self.side)  # the allocator must not hand the chunk out again before the copy has run

    def finish(self) -> torch.Tensor:
        if self.side is not None:
            self.side.synchronize()
        return self.out


class ChunkSource:
    """
    Input side of a shard that is consumed chunk by chunk: `get(i)` returns chunk i = rows [bounds[i][0], bounds[i][1]) of
    `items` as a tensor on `device`; `prefetch(i + 1)`, called AFTER chunk i's kernels have been queued, stages and uploads
    the next chunk underneath them.

    Device input: a view (nothing to move).  Host input, one chunk (one tile per request): one plain upload on the compute
    stream.  Host input, several chunks: a chunk is copied into one of two pinned staging blocks (a host memcpy, ~50 MB for
    64 tiles) and uploaded on a SIDE stream; the compute stream waits for a chunk's upload event, never the host.  Until
    round 5 `get(i)` did the staging copy of chunk i+1 BEFORE it returned chunk i, so the GPU idled during that memcpy
    whenever the host was not running ahead (for chunk 0 always); now the caller launches chunk i first and prefetches then.
    A caller that never calls `prefetch` still works: `get(i)` uploads chunk i on demand.  Chunk 0's upload is always exposed
    (it has nothing to hide behind).
    """

    def __init__(self, items: torch.Tensor, bounds: T.Sequence[T.Tuple[int, int]], device: torch.device):
        self.items, self.bounds, self.device = items, list(bounds), torch.device(device)
        self.staged = items.device.type == "cpu" and self.device.type == "cuda" and len(self.bounds) > 1
        self.ready: T.Dict[int, T.Tuple[torch.Tensor, T.Any]] = {}
        if self.staged:
            self.side = torch.cuda.Stream(self.device)
            rows = max(b - a for a, b in self.bounds)
            self.pinned = None if items.is_pinned() else [torch.empty((rows,) + tuple(items.shape[1:]), dtype=items.dtype, pin_memory=True) for _ in range(2)]
            self.pin_free: T.List[T.Any] = [None, None]  # event after which a staging block may be overwritten
            self._upload(0)

    def _upload(self, i: int) -> None:
        a, b = self.bounds[i]
        if self.pinned is None:
            src = self.items[a:b]
        else:
            j = i & 1
            if self.pin_free[j] is not None:
                self.pin_free[j].synchronize()  # the upload that last read this block (two chunks ago) has long finished
            src = self.pinned[j][: b - a]
            src.copy_(self.items[a:b])  # pageable -> pinned on the host, while the GPU computes the chunk queued before
        with torch.cuda.stream(self.side):
            dev = src.to(self.device, non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.side)
        if self.pinned is not None:
            self.pin_free[i & 1] = done
        self.ready[i] = (dev, done)

    def prefetch(self, i: int) -> None:
        """Stage and upload chunk i now (no-op for device input, a single chunk, an index past the end or a chunk already in
        flight).  Call it once the kernels of the chunk before have been launched."""
        if self.staged and 0 <= i < len(self.bounds) and i not in self.ready:
            self._upload(i)

    def get(self, i: int) -> torch.Tensor:
        a, b = self.bounds[i]
        if not self.staged:
            return self.items[a:b].to(self.device)
        if i not in self.ready:
            self._upload(i)  # nobody prefetched it: upload on demand
        dev, done = self.ready.pop(i)
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(done)
        dev.record_stream(cur)  # allocated on the side stream, consumed on the compute stream
        return dev
