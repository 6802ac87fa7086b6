"""sharding.py — how frame pairs are spread over the GPUs of a node, and the one exchange step.

Frame pairs (and Monte-Carlo trials) are independent, so the data path needs no collective: rank r of W owns the
contiguous slice [r*B/W, (r+1)*B/W) of a global batch (or simply its own B pairs under weak scaling).  The only
exchange is an all_gather of the per-pair velocity records ([B_local, 8] float32: vx, vy, vz, residual, n_used, s_min,
rank, corners).  `dist` is torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests).
"""


def shard_range(total, rank, world):
    """Contiguous slice of `total` units owned by `rank`; remainders go to the low ranks."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_records(dist, local_records, out=None):
    """all_gather of equally sized [B_local, 8] float32 record tensors -> [world*B_local, 8], rank-major."""
    import torch
    world = dist.get_world_size()
    if out is None:
        out = torch.empty((world * local_records.shape[0],) + tuple(local_records.shape[1:]), dtype=local_records.dtype,
                          device=local_records.device)
    dist.all_gather_into_tensor(out, local_records.contiguous())
    return out


def max_over_ranks(dist, seconds, device="cpu"):
    import torch
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
