"""Multi-GPU plumbing of the read-mapping hot path: one process per GPU, reads sharded, no data-path
collective.

The reference parallelises `smalt map -n T` with threads that pull read blocks from one queue and
share one read-only index (threads.c / rmap.c:1480-1560).  The MI355X form is
one process per GPU: the index image is built (or loaded) once on rank 0 and broadcast over
RCCL/xGMI, after which every rank maps its own contiguous shard of the reads; per-read results are
independent, so ranks only meet again to reduce the step statistics.  With `backend="gloo"` the same
code runs on CPU tensors (tests/test_shard_gloo.py).
"""
from __future__ import annotations

import os
from typing import Dict, List, Sequence, Tuple


def env_world() -> Tuple[int, int, int]:
    """(rank, world, local_rank) from the torch.distributed.run environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def shard_bounds(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of `n_items` reads for `rank`: sizes differ by at most one and the
    shards tile the input in rank order (so concatenating per-rank outputs restores the read order)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def broadcast_image(tensors: Dict[str, "object"], device, src: int = 0, order: Sequence[str] = ()):  # noqa: F821
    """Broadcast the index image (dict name -> tensor on rank `src`, anything elsewhere) to all ranks.
    Shapes and dtypes travel first (one int64 tensor), then each array in one collective.  Returns
    (dict name -> tensor on `device`, seconds spent in the data broadcasts); the seconds per array are left in
    `broadcast_image.last_by_array` (each broadcast is bracketed by a device synchronisation)."""
    import time

    import torch
    import torch.distributed as dist
    names = list(order) or sorted(tensors.keys())
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return {n: tensors[n] for n in names}, 0.0
    rank = dist.get_rank()
    dtypes = [torch.uint8, torch.int32, torch.int64, torch.uint32] if hasattr(torch, "uint32") else [torch.uint8, torch.int32, torch.int64]
    meta = torch.zeros(2 * len(names), dtype=torch.int64, device=device)
    if rank == src:
        for i, n in enumerate(names):
            t = tensors[n]
            assert t.dim() == 1 and t.dtype in dtypes, (n, t.dtype, t.shape)
            meta[2 * i] = t.numel()
            meta[2 * i + 1] = dtypes.index(t.dtype)
    dist.broadcast(meta, src)
    m = meta.cpu().tolist()
    out = {}
    sync = torch.cuda.synchronize if getattr(device, "type", str(device)) == "cuda" else (lambda: None)
    sync()
    t0 = time.time()
    by_array = {}
    for i, n in enumerate(names):
        t = tensors[n] if rank == src else torch.empty(int(m[2 * i]), dtype=dtypes[int(m[2 * i + 1])], device=device)
        ta = time.time()
        dist.broadcast(t, src)
        sync()
        by_array[n] = {"seconds": time.time() - ta, "bytes": int(t.numel() * t.element_size())}
        out[n] = t
    broadcast_image.last_by_array = by_array
    return out, time.time() - t0


def reduce_step(dt: float, counts: Sequence[float], device) -> Tuple[float, List[float]]:
    """Max of the step time over ranks and the sum of the per-rank counters (bench.py contract)."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return dt, list(counts)
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    c = torch.tensor(list(counts), dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(c, op=dist.ReduceOp.SUM)
    return float(t.item()), [float(x) for x in c.cpu().tolist()]


def gather_in_rank_order(obj):
    """All ranks' per-shard result objects, in rank order (a list of length world on every rank)."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out


class BatchDealer:
    """One cursor over the sub-batches of a job, shared by all ranks: a rank that finishes early takes the next sub-batch
    instead of idling behind a repeat-rich shard.  This is the reference's scheme -- its worker threads pull read blocks
    from one FIFO (threads.c:548 tprocf) -- with the c10d key-value store of the process group as the queue head (an
    atomic add on the host side; no collective, nothing on the data path).  Without a process group it is a plain local
    counter.  `static=True` deals contiguous shards instead (shard_bounds), e.g. when the ranks hold different reads."""

    def __init__(self, nbatches: int, static: bool = False):
        import torch.distributed as dist
        self.n = int(nbatches)
        self.static = static
        self.rank, self.world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
        self.store = None
        if self.world > 1 and not static:
            from torch.distributed import distributed_c10d as c10d
            self.store = dist.PrefixStore("smalt_deal", c10d._get_default_store())
        self.key = None
        self.local = 0
        self.hi = 0

    def start(self, tag: str) -> None:
        """Begin a pass over the job; `tag` must be the same on all ranks and unique per pass."""
        self.key = "pass/%s" % tag
        if self.store is None:
            self.local, self.hi = shard_bounds(self.n, self.rank, self.world) if self.world > 1 else (0, self.n)

    def next_range(self, total: int, max_chunk: int, min_chunk: int = 16384):
        """Guided dealing over `total` reads instead of fixed sub-batches: (start, count) of the next piece for this rank, or
        None.  A piece is half an even share of what is left, between min_chunk and max_chunk reads -- large pieces while
        there is plenty of work, small ones at the end, so that no rank idles behind another one's last sub-batch (20 M reads
        in fixed sub-batches of 262 144 over 8 ranks cannot do better than 77/80 = 96 %).  The cursor is the same c10d store
        counter; two ranks that size their piece from the same reading of it only get somewhat larger pieces than intended."""
        if self.store is None:
            if self.n != total:                       # static: this rank's contiguous shard, in pieces of max_chunk
                self.n = total
                self.local, self.hi = shard_bounds(total, self.rank, self.world) if self.world > 1 else (0, total)
            if self.local >= self.hi:
                return None
            c = min(max_chunk, self.hi - self.local)
            self.local += c
            return self.local - c, c
        cur = self.store.add(self.key, 0)
        if cur >= total:
            return None
        want = max(min_chunk, min(max_chunk, (total - cur) // (2 * self.world)))
        want = (want + 4095) // 4096 * 4096
        start = self.store.add(self.key, want) - want
        if start >= total:
            return None
        return start, min(want, total - start)

    def next(self):
        """Index of the next sub-batch for this rank, or None when the job is dealt out."""
        if self.store is None:
            if self.local >= self.hi:
                return None
            self.local += 1
            return self.local - 1
        i = self.store.add(self.key, 1) - 1
        return i if i < self.n else None
