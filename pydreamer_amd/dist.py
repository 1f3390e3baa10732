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


class _LazyComms:
    """The optimizer groups' RCCL communicators (csrc/comm.hip), created at first use - by whichever thread gets there first."""

    def __init__(self, optimizers, group):
        import threading
        self.optimizers, self.group, self.comms, self.lock = list(optimizers), group, None, threading.Lock()

    def get(self, i):
        with self.lock:
            if self.comms is None:
                self.comms = _native_comms(self.optimizers, self.group)
        return self.comms[i]


class _NativeWork:
    """What FusedAdamW keeps of an early all-reduce issued through the native entry point: the collective sits on the stream
    the backward pass ran on, and loss.backward() joins that stream anyway - nothing to wait for on the host."""

    def wait(self):
        return True


def _native_comms(optimizers, group):
    """One RCCL communicator per optimizer group (csrc/comm.hip): rank 0 draws the ids, the job's torch.distributed group
    carries them (any backend), every rank joins - collective, in optimizer order."""
    import ctypes
    from . import hip as H
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    comms = []
    for i, _ in enumerate(optimizers):
        if i > 0 and _os.environ.get('DM_DP_NATIVE_ONE_COMM') == '1':      # experiment: one communicator for every group
            comms.append(comms[0])
            continue
        ids = [None]
        if rank == 0:
            buf = (ctypes.c_char * 128)()
            H.call('dm_rccl_unique_id', buf)
            ids = [bytes(buf)]
        dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        comm = ctypes.c_void_p()
        H.call('dm_rccl_comm_init', ctypes.byref(comm), world, ctypes.c_char_p(ids[0]), rank)
        comms.append(comm)
    return comms


def attach(optimizers, local_batch, global_batch, group=None, model=None, native=None, force=False):
    """Enable gradient all-reduce inside FusedAdamW.clip_grad_norm for every optimizer group.
    native (default: the environment's DM_DP_NATIVE=1): issue the collectives through the library's own entry point
    dm_allreduce_grads (include/dreamer_hip.h) on one RCCL communicator PER GROUP instead of torch.distributed - a group's
    all-reduce is then ordered by the stream it is enqueued on and drain() is not needed.  GPU tensors + RCCL only.
    model (a pydreamer_amd Dreamer whose init_optimizers() produced `optimizers`): the B_r/B weight is FOLDED into the scale
    argument every backward entry point already takes (models.WorldModel / ActorCritic.grad_weight), so the rank's gradient
    buffers come out of the backward kernels already weighted and no extra pass over the 92 MB buffer runs per step; without
    it (or for a group fed by autograd, the 1-element probe group) the buffer is multiplied before the collective."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force):
        return      # (force: a ONE-rank group still issues its collectives - bench.py --force-dp measures their cost on a 1-GPU box)
    w = float(local_batch) / float(global_batch)
    folded = set()
    if model is not None and hasattr(model, 'prepare_streams'):
        model.prepare_streams()      # the step's streams take their hardware queues before any communicator is built (models.Dreamer.prepare_streams)
    if model is not None and getattr(model, '_opt', None) is not None:
        model.wm.grad_weight = w
        model.ac.grad_weight = w
        folded = {id(model._opt[k]) for k in ('wm', 'actor', 'critic')}
    global _attached
    import os
    if native is None:
        native = os.environ.get('DM_DP_NATIVE', '0') == '1'
    # native: the communicators are created LAZILY, at the first all-reduce (like torch creates its own): round 6 measured that a
    # communicator created BEFORE the model's side streams exist costs the step +13 ms (profiles/r06_force_dp.txt, run G) - RCCL's
    # internal streams then take the hardware queues first and the step's busy streams end up sharing one.  Every rank reaches
    # its first all-reduce at the same point of the program, so the collective creation stays matched.
    lazy = _LazyComms(optimizers, group) if native else None
    for i, opt in enumerate(optimizers):
        opt.dp = (group, w)
        opt.dp_folded = id(opt) in folded
        opt.dp_comm = (lazy, i) if native else None
    _attached = True


import os as _os
# Round 6 measured the collectives of a ONE-rank RCCL group on a 1-GPU box (bench.py --force-dp, profiles/r06_force_dp.txt):
#  * torch's EARLY all-reduce - issued from the launcher thread right behind a pre-launched backward, overlapped with the actor /
#    critic backward - runs on torch's communication stream, a FIFTH busy stream where the runtime has four hardware queues by
#    default: +5 ... +12 ms per step although a one-rank collective moves no data (GPU_MAX_HW_QUEUES=8: +0.1 ... +0.4 ms);
#  * the LATE form - every group reduced inside grad_clip(), when the step's streams have drained - costs +0.03 ... +0.06 ms;
#  * the library's own all-reduce (native=True / DM_DP_NATIVE=1) is enqueued on the backward's own stream - no fifth stream - and is
#    free in both forms, but has only ever run with one rank.
# Default: torch, late.  DM_DP_EARLY=1 selects the overlapped form (with DM_DP_NATIVE=1, or with GPU_MAX_HW_QUEUES=8 in the
# environment) for a fabric where the transfer itself is the larger cost.
_EARLY = _os.environ.get('DM_DP_EARLY', '0') == '1'
_inflight = []      # futures of launcher-thread jobs that may issue collectives (models._Overlap.submit)
_attached = False   # set by attach(): only then can a launcher job issue a collective, and only then does drain() ever run


def track(fut):
    """Remember a launcher-thread job that may issue a collective.  Nothing is kept unless a data-parallel group is attached
    (an initialised process group of world size 1 never reaches drain(), so its futures - each pinning the job's result -
    would pile up for the life of the process); jobs that have already finished are dropped on the way."""
    if not _attached:
        return
    _inflight[:] = [f for f in _inflight if not f.done()]
    _inflight.append(fut)


def drain():
    """Collectives must be issued in the same order on every rank.  The pre-launched backward passes issue theirs from the
    launcher thread (allreduce_scratch_async); before the MAIN thread issues one (the accumulation path of
    FusedAdamW.adopt_scratch, clip_grad_norm for a group without an early reduce) every launcher job of the step has to have
    issued its own - otherwise two ranks can interleave them differently and RCCL pairs mismatched buffers."""
    while _inflight:
        fut = _inflight.pop(0)
        try:
            fut.result()
        except Exception:      # re-raised, with its traceback, by the loss.backward() that owns the future (models._Overlap)
            pass


def _weight(opt, buf):
    group, w = opt.dp
    if w != 1.0 and not getattr(opt, 'dp_folded', False):
        buf.mul_(w)
    return group


def allreduce_scratch_async(opt):
    """Called on the side stream right after a pre-launched backward pass has been enqueued there (SURVEY.md 8(e):
    "issued as soon as each backward finishes (wm first, overlapped with actor/critic backward)"): starts the SUM all-reduce
    of the group's (B_r/B-weighted) buffer; the handle is waited for in loss.backward() (FusedAdamW.adopt_scratch), so
    grad_clip() finds the global-batch gradient already in place."""
    if opt.dp is None or not _EARLY:
        return
    group = _weight(opt, opt.scratch)
    if getattr(opt, 'dp_comm', None) is not None:
        _native_allreduce(opt, opt.scratch)
        opt.early_reduce = _NativeWork()
        return
    opt.early_reduce = dist.all_reduce(opt.scratch, op=dist.ReduceOp.SUM, group=group, async_op=True)


def _native_allreduce(opt, buf):
    """dm_allreduce_grads on the CURRENT stream (include/dreamer_hip.h; the group's own communicator)."""
    from . import hip as H
    lazy, i = opt.dp_comm
    H.call('dm_allreduce_grads', H.fptr(buf), buf.numel(), lazy.get(i), H.stream())


def allreduce_grads(opt):
    """grad <- sum_r (B_r/B) grad_r, in place on the flat buffer (one collective per optimizer group)."""
    if getattr(opt, 'dp_comm', None) is not None:      # own communicator: ordered by the stream, no cross-group issue order to keep
        _weight(opt, opt.flat_grad)
        return _native_allreduce(opt, opt.flat_grad)
    drain()
    group = _weight(opt, opt.flat_grad)
    dist.all_reduce(opt.flat_grad, op=dist.ReduceOp.SUM, group=group)
