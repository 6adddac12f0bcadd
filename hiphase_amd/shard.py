"""Block sharding for the multi-GPU path: phase blocks are independent
(reference src/phaser.rs:406-411, src/main.rs:385-408), so ranks never exchange block data — a rank only needs
to know WHICH blocks are its own. Used by bench.py (one process per GPU under torch.distributed.run) and by
callers that hold a global list of blocks.

`shard_lpt` is the static form (one process per GPU, every rank derives the same partition from the global list): longest
processing time first - blocks by estimated work descending, each to the least-loaded rank so far (block sizes are heavy-tailed:
median 15 hets, a few in the thousands, docs/user_guide.md:257-260). `simulate_queue` models the dynamic form inside the library
(hp_solve_blocks / hp_blockstream_* with device_id = -1, hp_block.hip solve_blocks_over_devices): chunks of about
total / (8 x devices) records, largest first, pulled by whichever device is free.
"""
import heapq


def shard_lpt(work, world_size):
    """work: list of per-block work estimates. Returns list[list[int]]: block indices per rank (greedy LPT; ties by index, so
    every rank computes the same partition)."""
    order = sorted(range(len(work)), key=lambda i: (-work[i], i))
    shards = [[] for _ in range(world_size)]
    heap = [(0, r) for r in range(world_size)]
    for i in order:
        load, r = heapq.heappop(heap)
        shards[r].append(i)
        heapq.heappush(heap, (load + work[i], r))
    return shards


def simulate_queue(records, n_devices, chunks_per_device=8):
    """The library's multi-device work queue on a list of per-block record counts: -> (makespan, mean load) in records, with a
    chunk's time taken as proportional to its records. Mirrors solve_blocks_over_devices (hp_block.hip): blocks sorted by records
    descending, a chunk closes once it holds total / (chunks_per_device x devices) records, chunks are taken in that order by the
    device that is free first."""
    order = sorted(range(len(records)), key=lambda i: (-records[i], i))
    total = sum(r + 1 for r in records)
    target = max(1, total // (n_devices * chunks_per_device))
    chunks, acc = [], 0
    for i in order:
        if not chunks or acc >= target:
            chunks.append(0)
            acc = 0
        chunks[-1] += records[i] + 1
        acc += records[i] + 1
    free = [0] * n_devices
    heapq.heapify(free)
    for c in chunks:
        heapq.heappush(free, heapq.heappop(free) + c)
    return max(free), total / n_devices


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


def gather_per_rank(dist, values, device=None):
    """[values of rank 0, values of rank 1, ...] for a short list of floats per rank (timings / counts only: no block data moves)."""
    import torch
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.tolist()] for o in out]
