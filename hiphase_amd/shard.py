"""Block sharding for the multi-GPU path: phase blocks are independent
(reference src/phaser.rs:406-411, src/main.rs:385-408), so ranks never exchange block data — a rank only needs
to know WHICH blocks are its own. Used by bench.py (one process per GPU under torch.distributed.run) and by
callers that hold a global list of blocks.

`shard_lpt` mirrors the in-library host queue of hp_astar_solve_batch(device_id=-1): sort by estimated work
(cells), deal in snake order so every rank gets a similar mix of large and small blocks.
"""


def shard_lpt(work, world_size):
    """work: list of per-block work estimates. Returns list[list[int]]: block indices per rank."""
    order = sorted(range(len(work)), key=lambda i: (-work[i], i))
    shards = [[] for _ in range(world_size)]
    for k, i in enumerate(order):
        rnd, pos = divmod(k, world_size)
        shards[pos if rnd % 2 == 0 else world_size - 1 - pos].append(i)
    return shards


def rank_seeds(base_seed, rank, n_blocks):
    """Deterministic, disjoint seed ranges per rank for the weak-scaling bench."""
    return [base_seed + rank * 1000003 + i for i in range(n_blocks)]


def gather_hets(dist, local_hets, device=None):
    """Sum of hets over all ranks (result reduction only — no block data moves)."""
    import torch
    t = torch.tensor([float(local_hets)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def max_over_ranks(dist, seconds, device=None):
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
