"""Multi-GPU plumbing: one process per GPU, contiguous env shards, no data-path collective.

Episodes are independent, so the step path needs no communication at all.  What is offered here is
(1) the shard arithmetic -- rank r of W owns global envs [lo, hi) and passes ``env_offset = lo`` so that
its Philox streams are those of the global env ids, independent of W -- and (2) the OPTIONAL gather of
the per-rank observation slices into one [N] tensor for a central policy (``all_gather_into_tensor``
over RCCL/xGMI on GPUs, gloo on CPU tensors) with the matching scatter of actions (a local slice).
"""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world_size):
    """-> (lo, hi): contiguous, sizes differ by at most one, concatenation in rank order = range(n_total)."""
    base, extra = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_observations(local, n_total=None, group=None, force_collective=False):
    """local [n_r] on every rank -> [N] on every rank (rank order).  Shards may differ in size by one.
    force_collective: run the collective with a single rank too (rehearsals of the N > 1 path on one GPU)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    if n_total is None:
        sizes = torch.tensor([local.numel()], device=local.device, dtype=torch.int64)
        allsz = [torch.zeros_like(sizes) for _ in range(world)]
        dist.all_gather(allsz, sizes, group=group)
        n_total = int(sum(int(s) for s in allsz))
    lens = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    if len(set(lens)) == 1:
        out = torch.empty(n_total, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    pad = max(lens)
    buf = torch.zeros(pad, dtype=local.dtype, device=local.device)
    buf[:local.numel()] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[:l] for p, l in zip(parts, lens)])


def local_actions(global_actions, n_total, group=None):
    """the slice of a replicated [N] action tensor that belongs to this rank."""
    if not dist.is_initialized():
        return global_actions
    lo, hi = shard_range(n_total, dist.get_rank(group), dist.get_world_size(group))
    return global_actions[lo:hi]
