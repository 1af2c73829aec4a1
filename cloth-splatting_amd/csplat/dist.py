"""View-parallel execution over the GPUs of one node (one process per GPU, torch.distributed; backend "nccl" = RCCL
over xGMI on the GPU box, "gloo" in the CPU tests).  The reference is single-GPU; this is the new
capability described in SURVEY.md 5.8 / 8(e):

  * Gaussians, mesh and simulator are replicated; the cameras of a step are sharded round-robin over ranks;
  * after backward ONE all-reduce(sum) runs over a single flat fp32 buffer holding every parameter gradient
    (240 B per Gaussian + simulator grads -- xGMI is point-to-point, a single large message per step is the
    per-link-friendly shape; no bucketing is needed at 24-55 MB);
  * densification statistics are reduced with the operator that keeps replicas bit-identical:
    screen-space gradient norm accum -> sum, denom -> sum, max_radii2D -> max, visibility -> max (logical or).
"""
import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_views(views, rank=None, world=None):
    """views i with i mod world == rank."""
    if rank is None:
        rank = dist.get_rank() if is_dist() else 0
    if world is None:
        world = dist.get_world_size() if is_dist() else 1
    return [v for i, v in enumerate(views) if i % world == rank]


def allreduce_flat(tensors, op=None, group=None):
    """All-reduce a list of same-dtype tensors as ONE flat buffer, in place.  Returns the flat buffer."""
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return None
    flat = torch.cat([t.reshape(-1) for t in tensors])
    if is_dist():
        dist.all_reduce(flat, op=op or dist.ReduceOp.SUM, group=group)
    pieces, o = [], 0
    for t in tensors:
        n = t.numel()
        pieces.append(flat[o:o + n].view_as(t))
        o += n
    try:
        torch._foreach_copy_(tensors, pieces)        # one multi-tensor launch instead of one copy per parameter
    except (AttributeError, RuntimeError):
        for t, src in zip(tensors, pieces):
            t.copy_(src)
    return flat


def allreduce_gradients(params, group=None):
    """sum the .grad of every parameter over ranks (parameters without grad contribute zeros so that all ranks
    issue the same collective)."""
    grads = []
    for p in params:
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        grads.append(p.grad)
    return allreduce_flat(grads, group=group)


def reduce_densification_stats(viewspace_grad, radii, visibility, group=None):
    """train_utils.py:276-277,290-292: viewspace gradient (sum over cams -> sum over ranks), radii (max), visibility (any)."""
    if not is_dist():
        return viewspace_grad, radii, visibility
    dist.all_reduce(viewspace_grad, op=dist.ReduceOp.SUM, group=group)
    packed = torch.stack([radii.to(torch.int32), visibility.to(torch.int32)])
    dist.all_reduce(packed, op=dist.ReduceOp.MAX, group=group)
    return viewspace_grad, packed[0].to(radii.dtype), packed[1].to(torch.bool)
