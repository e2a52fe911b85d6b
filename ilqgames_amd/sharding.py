"""Multi-GPU layout of a batch of independent game instances.

Instances share nothing, so the batch is cut into contiguous blocks, one per rank
(one process per GPU); the only exchange is the gather of the converged per-instance results to
rank 0 — RCCL over xGMI on the GPU box (backend "nccl"), gloo in the CPU tests.
"""


def instance_range(total, rank, world):
    """Contiguous block [lo, hi) of rank `rank`: the first total % world ranks take one extra."""
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_to_root(local, total, world, dst=0):
    """Gathers per-instance rows ([n_local, ...]) to rank `dst`; returns the [total, ...] tensor there,
    None elsewhere.  Blocks may differ in size by one row, so they are padded to the largest block
    for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank()
    sizes = [instance_range(total, r, world)[1] - instance_range(total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    pad = local
    if local.shape[0] < mx:
        pad = torch.cat([local, local.new_zeros((mx - local.shape[0],) + tuple(local.shape[1:]))])
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad.contiguous(), bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
