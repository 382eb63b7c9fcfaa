"""Multi-GPU plumbing for the hot path: one process per GPU, the batch sharded by rank,
NO collective inside the propagation (every op of reference cspn_pytorch/models/cspn.py is
per-sample).  The only exchange on the inference / evaluation path mirrors what nn.DataParallel
does implicitly per forward (reference cspn_pytorch/eval.py:117): replicate the backbone
weights -- done here ONCE, as a single flat broadcast (RCCL over xGMI when the backend is "nccl").

Training (reference cspn_pytorch/train.py:162-166 wraps the net in nn.DataParallel, which sums the
replicas' gradients into one model every step) additionally needs the gradients averaged across
ranks after every backward: call `allreduce_grads_(module)` before `optimizer.step()` (or wrap the
model in torch's DistributedDataParallel, which does the same bucketed).  broadcast_module_ alone
gives one diverging model per rank."""
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


def allreduce_grads_(module, group=None, average=True):
    """Training counterpart of nn.DataParallel's gradient reduction (reference train.py:162-166,196-199): one flat
    all_reduce per dtype over every parameter gradient, divided by the world size (each rank back-propagated the mean
    loss of ITS shard; the mean of the means is the loss of the whole batch when shards are equal).  Parameters whose
    .grad is None on this rank take part with zeros so that every rank issues the same collective.  Call between
    loss.backward() and optimizer.step().  Returns the number of bytes reduced."""
    world = dist.get_world_size(group)
    by_dtype = {}
    for p in module.parameters():
        if p.requires_grad:
            by_dtype.setdefault(p.dtype, []).append(p)
    total = 0
    for params in by_dtype.values():
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).detach().reshape(-1) for p in params])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat /= world
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[off:off + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[off:off + n].view_as(p))
                off += n
        total += flat.numel() * flat.element_size()
    return total


def gather_outputs(local_out, group=None):
    """all_gather of per-rank [B_r,...] results -> [B,...] on every rank, in rank order.  B_r may differ between ranks
    (shard_range hands out uneven chunks whenever B % world != 0): the per-rank sizes are exchanged first, every rank
    pads to the largest one for the collective, and the padding is trimmed afterwards."""
    world = dist.get_world_size(group)
    local_out = local_out.contiguous()
    n_local = torch.tensor([local_out.shape[0]], dtype=torch.int64, device=local_out.device)
    sizes = [torch.empty_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local, group=group)
    sizes = [int(x.item()) for x in sizes]
    n_max = max(sizes)
    if n_max == 0:
        return local_out
    if local_out.shape[0] < n_max:
        pad = local_out.new_zeros((n_max - local_out.shape[0],) + tuple(local_out.shape[1:]))
        local_out = torch.cat([local_out, pad], 0)
    parts = [torch.empty_like(local_out) for _ in range(world)]
    dist.all_gather(parts, local_out, group=group)
    return torch.cat([p[:n] for p, n in zip(parts, sizes)], 0)
