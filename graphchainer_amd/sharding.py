"""Read-parallel multi-GPU layout: one process per GPU, graph and index replicated, reads sharded.

The path shards embarrassingly (every read is independent, src/Aligner.cpp:492-1062), so there is no data-path
collective; torch.distributed is used only for the start barrier and the max-over-ranks of the step time.
"""


def shard_bounds(n_items, rank, world):
    """Contiguous, balanced shard [lo, hi) of n_items for `rank` of `world`."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def max_over_ranks(value, dist):
    """Max of a python float over all ranks (gloo or nccl)."""
    import torch
    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, dist):
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())
