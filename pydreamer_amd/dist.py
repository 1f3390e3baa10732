"""Data parallelism over the batch axis: one process per GPU, RCCL all-reduce of each optimizer group's flat gradient
buffer over xGMI (torch.distributed backend "nccl" IS RCCL on ROCm).  The reference has no data parallelism at all
(SURVEY.md 2.2); this keeps single-GPU semantics: every rank steps identical replicated AdamW state.

Sharding rule: rank r takes columns B_r of every (T,B,...) tensor; B need not divide evenly
(B=50 over 8 ranks -> 7,7,6,6,6,6,6,6).  A rank's losses are means over ITS T*B_r rows, so its gradients are weighted
by B_r/B before the SUM all-reduce; the result equals the single-process gradient of the global-batch mean.
"""
import torch
import torch.distributed as dist


def shard_bounds(batch, world, rank):
    """[lo, hi) columns of the global batch owned by `rank` (first `batch % world` ranks get one extra)."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_obs(obs, world, rank):
    """Slice a time-major observation dict on the batch axis (dim 1)."""
    B = obs['action'].shape[1]
    lo, hi = shard_bounds(B, world, rank)
    return {k: v[:, lo:hi].contiguous() for k, v in obs.items()}, (lo, hi)


def attach(optimizers, local_batch, global_batch, group=None):
    """Enable gradient all-reduce inside FusedAdamW.clip_grad_norm for every optimizer group."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    w = float(local_batch) / float(global_batch)
    for opt in optimizers:
        opt.dp = (group, w)


def allreduce_scratch_async(opt):
    """Called on the side stream right after a pre-launched backward pass has been enqueued there (SURVEY.md 8(e):
    "issued as soon as each backward finishes (wm first, overlapped with actor/critic backward)"): weights the rank's
    gradient by B_r/B and starts the SUM all-reduce of the group's buffer; the handle is waited for in loss.backward()
    (FusedAdamW.adopt_scratch), so grad_clip() finds the global-batch gradient already in place."""
    if opt.dp is None:
        return
    group, w = opt.dp
    if w != 1.0:
        opt.scratch.mul_(w)
    opt.early_reduce = dist.all_reduce(opt.scratch, op=dist.ReduceOp.SUM, group=group, async_op=True)


def allreduce_grads(opt):
    """grad <- sum_r (B_r/B) grad_r, in place on the flat buffer (one collective per optimizer group)."""
    group, w = opt.dp
    if w != 1.0:
        opt.flat_grad.mul_(w)
    dist.all_reduce(opt.flat_grad, op=dist.ReduceOp.SUM, group=group)
