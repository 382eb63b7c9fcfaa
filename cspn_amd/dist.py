"""Multi-GPU plumbing for the hot path: one process per GPU, the batch sharded by rank,
NO collective inside the propagation (every op of reference cspn_pytorch/models/cspn.py is
per-sample).  The only exchange mirrors what nn.DataParallel does implicitly per forward
(reference cspn_pytorch/train.py:165, eval.py:117): replicate the backbone weights --
done here ONCE, as a single flat broadcast (RCCL over xGMI when the backend is "nccl")."""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous, balanced [lo, hi) of `total` independent samples for `rank` of `world`."""
    base, rem = divmod(int(total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(tensors, rank=None, world=None):
    """Slice every [B,...] tensor (or None) to this rank's contiguous chunk."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        lo, hi = shard_range(t.shape[0], rank, world)
        out.append(t[lo:hi])
    return out


def broadcast_flat_(tensors, src=0, group=None):
    """One collective for a whole list of same-dtype tensors (weights of the affinity
    backbone): pack -> broadcast -> unpack in place.  Returns the number of bytes sent."""
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return 0
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].view_as(t))
            off += n
    return flat.numel() * flat.element_size()


def broadcast_module_(module, src=0, group=None):
    """Replicate a module's parameters and buffers from `src` (per-dtype flat buffers)."""
    by_dtype = {}
    for t in list(module.parameters()) + list(module.buffers()):
        by_dtype.setdefault(t.dtype, []).append(t)
    return sum(broadcast_flat_(ts, src, group) for ts in by_dtype.values())


def gather_outputs(local_out, group=None):
    """all_gather of per-rank [B_r,1,H,W] results (equal B_r) -> [B,1,H,W] on every rank."""
    world = dist.get_world_size(group)
    parts = [torch.empty_like(local_out) for _ in range(world)]
    dist.all_gather(parts, local_out.contiguous(), group=group)
    return torch.cat(parts, 0)
