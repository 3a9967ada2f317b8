"""Multi-GPU sharding of the MAM stage: one process per GPU, reads are independent, no collective on
the data path.  The reference's unit of distribution is the 1000-read batch
(/root/reference/modules/pc.py:40); batches are dealt round-robin to ranks, balanced by Myers cells
when cell counts are known.  torch.distributed is used only to agree on timing and totals."""


def shard_range(n_items, rank, world):
    """Contiguous [begin, end) share of n_items for `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_items, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def deal_batches(n_reads, rank, world, batch=1000, cells=None):
    """Read indices of `rank`: batches of `batch` reads dealt round-robin, or - when per-read cell
    counts are given - greedily to the least loaded rank (long reads cost more). Deterministic."""
    batches = [range(b, min(b + batch, n_reads)) for b in range(0, n_reads, batch)]
    if cells is None:
        mine = [b for i, b in enumerate(batches) if i % world == rank]
    else:
        load = [0] * world
        owner = []
        order = sorted(range(len(batches)), key=lambda i: (-sum(cells[j] for j in batches[i]), i))
        assign = {}
        for i in order:
            r = min(range(world), key=lambda q: (load[q], q))
            assign[i] = r
            load[r] += sum(cells[j] for j in batches[i])
        mine = [b for i, b in enumerate(batches) if assign[i] == rank]
    return [j for b in mine for j in b]


def reduce_job(elapsed_s, reads, cells, device=None):
    """(max elapsed over ranks, total reads, total cells). Works with any initialised backend
    (nccl = RCCL on the GPU box, gloo in the CPU tests); identity when not distributed."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(elapsed_s), float(reads), float(cells)
    t = torch.tensor([float(elapsed_s)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    tot = torch.tensor([float(reads), float(cells)], dtype=torch.float64, device=device)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    return float(t.item()), float(tot[0].item()), float(tot[1].item())
