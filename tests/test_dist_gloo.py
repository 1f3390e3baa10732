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


def test_attach_is_noop_single_process():
    opt = _FakeOpt(torch.ones(4))
    DP.attach([opt], 5, 5)
    assert opt.dp is None
