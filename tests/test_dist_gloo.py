"""CPU (-m "not gpu"): the N>1 path with world_size=2 over gloo — shard bounds + weighted SUM all-reduce of a flat
gradient buffer reproduce the global-batch mean gradient; uneven shards (B=5 -> 3,2)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from pydreamer_amd import dist as DP


class _FakeOpt:
    """Stands in for FusedAdamW's flat buffers (which need a GPU): allreduce_grads only touches flat_grad and dp."""

    def __init__(self, g):
        self.flat_grad = g
        self.dp = None


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(0)
        per_col = torch.randn(B, 1000, generator=g)                 # per-batch-column gradient contributions
        obs = dict(action=torch.zeros(4, B, 3), image=torch.arange(4 * B).float().view(4, B))
        shard, (lo, hi) = DP.shard_obs(obs, world, rank)
        assert shard['image'].shape == (4, hi - lo) and torch.equal(shard['image'], obs['image'][:, lo:hi])
        local_mean_grad = per_col[lo:hi].mean(0)                    # what a rank's backward produces (mean over ITS rows)
        opt = _FakeOpt(local_mean_grad.clone())
        DP.attach([opt], hi - lo, B)
        assert opt.dp is not None and abs(opt.dp[1] - (hi - lo) / B) < 1e-12
        DP.allreduce_grads(opt)
        ref = per_col.mean(0)
        assert torch.allclose(opt.flat_grad, ref, atol=1e-6), float((opt.flat_grad - ref).abs().max())
        out[rank] = (lo, hi)
    finally:
        dist.destroy_process_group()


def test_weighted_allreduce_world2_uneven():
    world, B = 2, 5
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert dict(out) == {0: (0, 3), 1: (3, 5)}


def _native_worker(rank, world, port, B, out):
    """The NATIVE exchange path of dist.py (DM_DP_NATIVE / attach(native=True): one communicator per optimizer group through the
    library's dm_rccl_* / dm_allreduce_grads entry points) with those three entry points STUBBED on the CPU - RCCL needs GPUs -
    by a gloo all-reduce: what is checked is dist.py's own logic (rank 0 draws one id per group, every rank receives it, joins in
    optimizer order, the weighted SUM lands in the flat buffer, the early all-reduce needs no host-side handle and no drain)."""
    import ctypes
    from pydreamer_amd import hip as H
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    calls = []
    bufs = {}

    def fake_call(name, *a):
        calls.append(name)
        if name == 'dm_rccl_unique_id':
            ctypes.memmove(a[0], bytes([len([c for c in calls if c == name])]) * 128, 128)
        elif name == 'dm_rccl_comm_init':
            comm_ref, n, idp, r = a
            assert n == world and r == rank
            comm_ref._obj.value = 1000 + idp.value[0]       # the id's first byte names the communicator
        elif name == 'dm_allreduce_grads':
            ptr, n, comm, stream = a
            t = bufs[ptr.value]
            assert t.numel() == n
            dist.all_reduce(t)
        else:
            raise AssertionError(name)
        return 0
    keep = (H.call, H.fptr, H.stream)
    H.call = fake_call
    H.fptr = lambda t: (bufs.__setitem__(t.data_ptr(), t), ctypes.c_void_p(t.data_ptr()))[1]
    H.stream = lambda: None
    try:
        g = torch.Generator().manual_seed(0)
        per_col = [torch.randn(B, 64, generator=g) for _ in range(3)]
        lo, hi = DP.shard_bounds(B, world, rank)
        opts = [_FakeOpt(pc[lo:hi].mean(0).clone()) for pc in per_col]
        for o in opts:
            o.scratch = o.flat_grad
        DP.attach(opts, hi - lo, B, native=True)
        assert calls == [], 'the communicators are created lazily, at the first all-reduce'
        # early (stream-ordered) form for group 0, plain form for the others; no launcher future is tracked and drain() is not
        # reached (the stub's single gloo group still wants one issue order, RCCL's per-group communicators would not)
        DP._EARLY = False                                    # the product default (round 6): no early all-reduce, nothing is issued
        DP.allreduce_scratch_async(opts[0])
        assert getattr(opts[0], 'early_reduce', None) is None and calls.count('dm_allreduce_grads') == 0
        DP._EARLY = True                                     # DM_DP_EARLY=1: the overlapped form
        DP.allreduce_scratch_async(opts[0])
        assert isinstance(opts[0].early_reduce, DP._NativeWork) and opts[0].early_reduce.wait()
        DP._inflight.append('poison: allreduce_grads must not drain on the native path')
        for i in (1, 2):
            DP.allreduce_grads(opts[i])
        assert DP._inflight == ['poison: allreduce_grads must not drain on the native path']
        DP._inflight.clear()
        assert calls.count('dm_allreduce_grads') == 3 and calls.count('dm_rccl_comm_init') == 3
        assert [o.dp_comm[0].get(o.dp_comm[1]).value for o in opts] == [1001, 1002, 1003], 'one communicator per group, ids in optimizer order'
        out[rank] = [float((o.flat_grad - pc.mean(0)).abs().max()) for o, pc in zip(opts, per_col)]
    finally:
        H.call, H.fptr, H.stream = keep
        dist.destroy_process_group()


def test_native_exchange_control_flow_world2():
    world, B = 2, 5
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_native_worker, args=(world, _free_port(), B, out), nprocs=world, join=True)
    assert all(e < 1e-6 for errs in dict(out).values() for e in errs), dict(out)


def test_attach_is_noop_single_process():
    opt = _FakeOpt(torch.ones(4))
    DP.attach([opt], 5, 5)
    assert opt.dp is None
