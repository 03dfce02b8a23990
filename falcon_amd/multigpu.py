"""Sharding of piles over ranks (one process per GPU) and the reduction of the
per-rank measurements.  Piles are independent units (the reference fans them out
over a process pool, consensus.py:264-274): the data path needs no collective;
``torch.distributed`` is only used to line the ranks up and to combine timings
(backend "nccl" = RCCL on the GPU box, "gloo" in the CPU tests)."""
from __future__ import annotations

from typing import List, Sequence, Tuple


def partition(n_items: int, rank: int, world: int) -> range:
    """Contiguous, balanced shard of ``range(n_items)`` owned by ``rank``."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def shard(items: Sequence, rank: int, world: int) -> List:
    r = partition(len(items), rank, world)
    return [items[i] for i in r]


def reduce_measurement(units: float, piles: float, elapsed_s: float, device=None) -> Tuple[float, float, float]:
    """(sum of units, sum of piles, max elapsed) over all ranks; identity when
    torch.distributed is not initialised."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return units, piles, elapsed_s
    sums = torch.tensor([units, piles], dtype=torch.float64, device=device)
    tmax = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    return float(sums[0].item()), float(sums[1].item()), float(tmax[0].item())


def gather_in_order(local_results: List, rank: int, world: int) -> List:
    """All ranks' results concatenated in rank order (= input order for contiguous
    shards).  Used by multi-process drivers that print on rank 0."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or world == 1:
        return list(local_results)
    parts = [None] * world
    dist.all_gather_object(parts, list(local_results))
    return [x for p in parts for x in p]


def gather_per_rank(values: Sequence[float], device=None) -> List[List[float]]:
    """Every rank's list of numbers, in rank order (a row per rank); [values] when
    torch.distributed is not initialised.  What rank 0 prints beside the whole-job figure, so
    that a straggling rank shows."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [list(values)]
    mine = torch.tensor(list(values), dtype=torch.float64, device=device)
    rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(rows, mine)
    return [[float(x) for x in r.tolist()] for r in rows]
