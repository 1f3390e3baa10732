"""-m gpu: the data-parallel step through the REAL optimizer path (FusedAdamW + pydreamer_amd.dist), two ranks.

With >= 2 visible GPUs the ranks run one per GPU over RCCL ("nccl"); on a 1-GPU box both ranks share cuda:0 and talk
over gloo (which all-reduces CUDA tensors through host staging) - the same FusedAdamW.clip_grad_norm / (late or early)
all-reduce code runs either way.  Bars: posterior indices of every shard bit-identical to the matching
columns of the 1-rank run (uniforms are sliced from the global layout); parameters after clip + AdamW within 1e-5."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _slice_noise(noise, T, B, S, Hh, lo, hi):
    return dict(u_post=noise['u_post'][:, lo:hi].contiguous(),
                u_act=noise['u_act'].view(Hh, T, B)[:, :, lo:hi].reshape(Hh, -1).contiguous(),
                u_prior=noise['u_prior'].view(Hh, T, B, S)[:, :, lo:hi].reshape(Hh, -1, S).contiguous())


def _one_step(model, conf, opts, obs, noise, steps=2):
    out = None
    state = model.init_state(obs['action'].shape[1])
    for _ in range(steps):                          # 2 steps: the second one runs the steady-state buffer-swap path
        losses, state, metrics, tensors, _ = model.training_step(obs, state, noise=noise)
        for opt in opts:
            opt.zero_grad()
        for loss in losses:
            loss.backward()
        gm = model.grad_clip(conf.grad_clip, conf.grad_clip_ac)
        for opt in opts:
            opt.step()
        out = (gm, model.last_extras['post_idx'].clone())
    return out


def _worker(rank, world, port, overlap, fold, out, early=False):
    import torch.distributed as dist
    from oracle import dreamer_oracle as O
    from pydreamer_amd import config
    from pydreamer_amd import dist as DP
    from pydreamer_amd.models import Dreamer
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    ndev = torch.cuda.device_count()
    backend = 'nccl' if ndev >= world else 'gloo'
    dev = torch.device('cuda', rank % ndev)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        oconf = O.tiny_conf(batch_size=5, batch_length=4, imag_horizon=3)
        T, B, S, Hh = oconf.batch_length, oconf.batch_size, oconf.stoch_dim, oconf.imag_horizon
        params = O.make_params(oconf, seed=2)
        obs = {k: v.to(dev) for k, v in O.preprocess(O.synthetic_batch(oconf), oconf).items()}
        noise = {k: v.to(dev) for k, v in O.make_noise(oconf).items()}
        lo, hi = DP.shard_bounds(B, world, rank)
        c = O.make_conf(**{**vars(oconf), 'batch_size': hi - lo})
        conf = config.load_config('defaults', 'atari', **{k: getattr(c, k) for k in vars(c)})
        model = Dreamer(conf)
        model.load_state_dict(params, strict=True)
        model = model.to(dev)
        model.overlap_backward = overlap
        opts = model.init_optimizers(conf.adam_lr, conf.adam_lr_actor, conf.adam_lr_critic, conf.adam_eps)
        DP._EARLY = early       # the all-reduce behind each pre-launched backward (overlapped) instead of inside grad_clip (round 6: the default is late)
        DP.attach(opts, hi - lo, B, model=model if fold else None)    # fold: B_r/B inside the backward kernels' scales
        assert opts[0].dp is not None and opts[0].dp_folded == fold and not opts[1].dp_folded
        shard, _ = DP.shard_obs(obs, world, rank)
        gm, idx = _one_step(model, conf, opts, shard, _slice_noise(noise, T, B, S, Hh, lo, hi))
        torch.cuda.synchronize()
        out[rank] = dict(lo=lo, hi=hi, idx=idx.cpu(), params=[o.flat_param.cpu() for o in opts],
                         norms={k: float(v) for k, v in gm.items()}, backend=backend)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('overlap,fold,early', [(True, True, False), (False, True, False), (True, False, False), (True, True, True)])
def test_two_rank_step_equals_one_rank(hip, overlap, fold, early):
    import torch.multiprocessing as mp
    from oracle import dreamer_oracle as O
    from pydreamer_amd import config
    from pydreamer_amd.models import Dreamer
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), overlap, fold, out, early), nprocs=world, join=True)
    res = dict(out)
    assert set(res) == {0, 1}
    # the single-process run of the whole batch
    oconf = O.tiny_conf(batch_size=5, batch_length=4, imag_horizon=3)
    conf = config.load_config('defaults', 'atari', **{k: getattr(oconf, k) for k in vars(oconf)})
    model = Dreamer(conf)
    model.load_state_dict(O.make_params(oconf, seed=2), strict=True)
    model = model.to('cuda')
    model.overlap_backward = overlap
    opts = model.init_optimizers(conf.adam_lr, conf.adam_lr_actor, conf.adam_lr_critic, conf.adam_eps)
    obs = {k: v.to('cuda') for k, v in O.preprocess(O.synthetic_batch(oconf), oconf).items()}
    noise = {k: v.to('cuda') for k, v in O.make_noise(oconf).items()}
    gm, idx = _one_step(model, conf, opts, obs, noise)
    torch.cuda.synchronize()
    for r in (0, 1):
        lo, hi = res[r]['lo'], res[r]['hi']
        assert torch.equal(res[r]['idx'], idx[:, lo:hi].cpu()), f'rank {r}: posterior indices differ from the 1-rank run'
        for i, (a, o) in enumerate(zip(res[r]['params'], opts)):
            err = float((a - o.flat_param.cpu()).abs().max())
            assert err < 1e-5, f'rank {r} group {i}: parameters after 2 steps differ by {err}'
        for k, v in res[r]['norms'].items():
            assert abs(v - float(gm[k])) <= 1e-4 * max(abs(float(gm[k])), 1e-6), (k, v, float(gm[k]))
    # both ranks hold identical replicas
    for a, b in zip(res[0]['params'], res[1]['params']):
        assert torch.equal(a, b)


def test_bench_multi_rank_code_path_smoke(hip):
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), on whatever this box has:
    with one GPU both ranks share cuda:0 over gloo (DM_BENCH_ONE_DEVICE=1, the line carries an INVALID marker).  Checks the
    contract of the N > 1 line: one JSON line from rank 0, n_gpus, steps, the uneven 25/25 sharding, finite loss."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env['DM_BENCH_ONE_DEVICE'] = '1'
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--reps', '1', '--steps', '3', '--warmup', '1',
           '--prof-steps', '1']
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['steps'] == 3 and d['warmup'] == 1 and d['scaling'] == 'strong' and d['cpu_baseline'] is None
    assert d['value'] > 0 and abs(d['ms_per_step'] * d['value'] - 1e3) < 1e-6 * 1e3
    assert '[25, 25]' in d['config']['parallelism']
    dd = d['distributed']
    assert dd['world_size'] == 2 and len(dd['ms_per_step_per_rank']) == 2 and set(dd['allreduce_standalone']) == {'wm', 'probe', 'actor', 'critic'}
    assert dd['allreduce_standalone']['wm']['bytes'] > 80e6 and dd['allreduce_standalone']['wm']['ms'] > 0
    assert np.isfinite(d['loss_model_last']) and d['roofline'] is not None and d['roofline']['frac'] > 0


def _run_bench(world, extra_env, *flags):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **extra_env)
    if world > 1:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(world), '--master-addr', '127.0.0.1',
               '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', str(world)]
    else:
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1']
    out = subprocess.run(cmd + list(flags), env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_eight_rank_job_at_atari_literal_equals_one_rank(hip):
    """BASELINE configs[3] as far as one GPU can show it: bench.py --gpus 8 at the FULL Atari-literal dimensions, one process
    per rank as the driver launches it.  On a box with < 8 GPUs all ranks share cuda:0 and talk over gloo
    (DM_BENCH_ONE_DEVICE=1; same FusedAdamW / all-reduce-in-grad_clip / B_r/B-folded backward code as over RCCL).  Checks
    (SURVEY 8(e)): the 50 columns are dealt 7/7/6/6/6/6/6/6 (the 6-column shard shape of ranks 2-7 executes here at full
    width), every rank holds BIT-IDENTICAL parameters after 2 steps, the loss of the global batch (sum_r B_r/B loss_r) and the
    parameters equal the 1-rank run on the same global batch (the replay ring and the sampler uniforms are drawn in the global
    layout and sliced) up to fp32 summation order."""
    flags = ('--reps', '1', '--steps', '2', '--warmup', '0', '--prof-steps', '0', '--no-cpu-baseline', '--no-h2d-leg', '--ring', '2')
    ndev = torch.cuda.device_count()
    d8 = _run_bench(8, {'DM_BENCH_ONE_DEVICE': '1'} if ndev < 8 else {}, *flags)
    d1 = _run_bench(1, {}, *flags)
    dd = d8['distributed']
    assert d8['n_gpus'] == 8 and dd['world_size'] == 8
    assert dd['shard_columns'] == [7, 7, 6, 6, 6, 6, 6, 6]
    assert dd['backend'] == ('nccl' if ndev >= 8 else 'gloo')
    assert dd['replicas_identical'] is True
    l8, l1 = dd['loss_model_global'], d1['loss_model_last']
    print('loss_model of the global batch: 8 ranks', l8, '1 rank', l1)
    assert abs(l8 - l1) <= 1e-4 * abs(l1), (l8, l1)
    # parameters: sums of every optimizer group against the 1-rank run.  Not bit-equal: a rank's 6-7 columns run other kernel
    # variants (row counts per lane group), so gradients differ in the last bits, and AdamW's first steps move a parameter by
    # lr * sign-like(g) - an element whose gradient is rounding noise around zero lands 2 lr apart (measured: 1e-2 on the
    # actor's sum of 1.1 M parameters, 6e-5 on the world model's).  Bar: 2e-5 of the group's sum of absolute values.
    c8, c1 = dd['param_checksum_rank0'], d1['param_checksum']
    for i in range(4):
        assert abs(c8[i] - c1[i]) <= 2e-5 * c1[4 + i] + 1e-6, (i, c8, c1)


@pytest.mark.parametrize('world', [2, 4, 8])
def test_default_deployment_over_rccl_on_real_gpus(hip, world):
    """The deployment every real shard runs - one process per GPU over RCCL ("nccl"), data-parallel FusedAdamW with the (round 6: late, inside grad_clip; DM_DP_EARLY=1: early)
    all-reduce and the B_r/B weight folded into the backward kernels, the pipelined actor / critic optimizer, AND the persistent
    posterior kernel ON (its default for <= 32-column shards: 25/25, 13/13/12/12, 7/7/6/6/6/6/6/6 of the 50 columns) - at the
    full Atari-literal size, through bench.py as the driver launches it.  Needs `world` GPUs: two trainers cannot share one
    device with the persistent kernel on (each needs every CU at once), which is why the one-device smoke mode of the tests
    above switches it off; this test is the first place the real thing runs.  Asserted: RCCL saw `world` ranks, the persistent
    kernel ran on every rank and none gave up (dm_rssm_lds_status 0 everywhere), every rank holds bit-identical parameters
    after 2 steps, and the global batch's loss / parameter checksums equal the 1-rank run's up to fp32 summation order."""
    ndev = torch.cuda.device_count()
    if ndev < world:
        pytest.skip(f'needs {world} GPUs for one process per GPU over RCCL; this box has {ndev}')
    flags = ('--reps', '1', '--steps', '2', '--warmup', '0', '--prof-steps', '0', '--no-cpu-baseline', '--no-h2d-leg', '--ring', '2')
    dn = _run_bench(world, {}, *flags)
    d1 = _run_bench(1, {}, *flags)
    dd = dn['distributed']
    assert dn['n_gpus'] == world and dd['world_size'] == world and dd['backend'] == 'nccl' and dn['rccl_version']
    assert 'INVALID_smoke_all_ranks_on_one_device' not in dn
    assert sum(dd['shard_columns']) == 50 and max(dd['shard_columns']) <= 32
    assert all(dd['persistent_posterior_kernel_ran']), dd['persistent_posterior_kernel_ran']
    assert dd['rssm_lds_status'] == [0] * world
    assert dd['replicas_identical'] is True
    l8, l1 = dd['loss_model_global'], d1['loss_model_last']
    assert abs(l8 - l1) <= 1e-4 * abs(l1), (l8, l1)
    c8, c1 = dd['param_checksum_rank0'], d1['param_checksum']
    for i in range(4):
        assert abs(c8[i] - c1[i]) <= 2e-5 * c1[4 + i] + 1e-6, (i, c8, c1)


def test_native_rccl_entry_points_one_rank():
    """The native exchange step (include/dreamer_hip.h dm_rccl_* / dm_allreduce_grads, csrc/comm.hip) on real hardware as far as a
    1-GPU box can take it: RCCL is bound with dlopen, a ONE-rank communicator is created on the current device, and the in-place
    SUM all-reduce of two flat fp32 buffers - enqueued on two different non-default streams, one communicator each, as
    dist.attach(native=True) lays them out per optimizer group - leaves the data unchanged (the sum over one rank) and ordered
    behind the kernel that wrote the buffer on that stream.  N > 1 needs N GPUs (RCCL refuses two ranks on one device)."""
    import ctypes
    from pydreamer_amd import hip as H
    lib = H.lib()
    if not lib.dm_rccl_available():
        pytest.skip('librccl is not loadable on this box')
    assert lib.dm_rccl_version() > 20000
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    comms, streams, bufs, refs = [], [], [], []
    for i in range(2):
        idb = (ctypes.c_char * 128)()
        H.call('dm_rccl_unique_id', idb)
        comm = ctypes.c_void_p()
        H.call('dm_rccl_comm_init', ctypes.byref(comm), 1, ctypes.c_char_p(bytes(idb)), 0)
        assert comm.value
        comms.append(comm)
        streams.append(torch.cuda.Stream(dev))
    for i, (comm, st) in enumerate(zip(comms, streams)):
        with torch.cuda.stream(st):
            x = torch.randn(1_000_003 + i, device=dev)
            y = x * 3.0 + 1.0                      # the producer on this stream; the collective must see its result
            H.call('dm_allreduce_grads', H.fptr(y), y.numel(), comm, H.stream())
            z = y * 0.5                             # ... and the consumer the collective's
            bufs.append(z)
            refs.append((x * 3.0 + 1.0) * 0.5)
    torch.cuda.synchronize()
    for z, r in zip(bufs, refs):
        assert torch.equal(z, r)
    with pytest.raises(H.DreamerHipError):
        H.call('dm_allreduce_grads', None, 4, comms[0], None)
    for comm in comms:
        H.call('dm_rccl_comm_destroy', comm)


def test_default_shard_deployment_over_one_rank_rccl(hip):
    """What a 1-GPU box can run of the real deployment (round 6): `bench.py --force-dp --emulate-world 8 --pipeline` - rank 0's
    7-column shard of an 8-way split at the FULL Atari-literal size, the data-parallel code path switched on over a ONE-rank RCCL
    ("nccl") group: B_r/B folded into the backward kernels, every optimizer group all-reduced by RCCL inside grad_clip (the
    round-6 default), the pipelined actor / critic optimizer AND the persistent posterior kernel on.  Until this round RCCL had
    executed zero times.  Asserted: the backend is nccl, the persistent kernel ran and did not give up, the loss is finite and
    the line is marked as the diagnostic it is."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--force-dp', '--emulate-world', '8', '--pipeline', '--reps', '1',
           '--steps', '3', '--warmup', '1', '--prof-steps', '0', '--no-cpu-baseline', '--no-h2d-leg', '--ring', '2']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['INVALID_diagnostic_forced_one_rank_dp'] == 'torch.distributed nccl' and d['INVALID_diagnostic_emulated_world'] == 8
    di = d['distributed']
    assert di['backend'] == 'nccl' and di['world_size'] == 1 and di['rccl_version']
    assert di['persistent_posterior_kernel_ran'] == [True] and di['rssm_lds_status'] == [0]
    assert di['replicas_identical'] and np.isfinite(d['loss_model_last']) and d['loss_model_last'] > 0
    assert all(v['ms'] >= 0 for v in di['allreduce_standalone'].values())


def test_native_overlapped_allreduce_equals_late_torch_one_rank(hip):
    """The library's own exchange step in its OVERLAPPED form (DM_DP_NATIVE=1 DM_DP_EARLY=1: dm_allreduce_grads enqueued right
    behind each pre-launched backward on that backward's stream, one communicator per optimizer group, created at the first
    all-reduce) against the product default (torch.distributed, inside grad_clip) on the 7-column shard over a ONE-rank RCCL
    group: a sum over one rank is the identity, so after the same steps the parameter checksums must be EQUAL to the bit."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = {}
    for name, extra in (('late', {}), ('native_early', dict(DM_DP_NATIVE='1', DM_DP_EARLY='1'))):
        env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()), **extra)
        cmd = [sys.executable, os.path.join(root, 'bench.py'), '--force-dp', '--emulate-world', '8', '--reps', '1',
               '--steps', '3', '--warmup', '1', '--prof-steps', '0', '--no-cpu-baseline', '--no-h2d-leg', '--ring', '2']
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        lines[name] = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert lines['native_early']['INVALID_diagnostic_forced_one_rank_dp'] == 'native dm_allreduce_grads'
    a, b = (lines[k]['distributed']['param_checksum_rank0'] for k in ('late', 'native_early'))
    assert a == b, (a, b)
    assert lines['late']['loss_model_last'] == lines['native_early']['loss_model_last']


_ORDER_SCRIPT = r'''
import sys, warnings
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import dreamer_oracle as O
from pydreamer_amd import config
from pydreamer_amd.models import Dreamer
dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1)
if sys.argv[2] == 'collective_first':
    dist.all_reduce(torch.zeros(4, device=dev))
    torch.cuda.synchronize()
c = O.tiny_conf(batch_size=5, batch_length=4, imag_horizon=3)
conf = config.load_config('defaults', 'atari', **{k: getattr(c, k) for k in vars(c)})
model = Dreamer(conf).to(dev)
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter('always')
    model.init_optimizers(1e-4, 1e-4, 1e-4, 1e-5)
assert model._overlap is not None, 'init_optimizers() creates the streams'
print('WARNED' if any('RCCL communicator exists' in str(x.message) for x in w) else 'QUIET')
dist.destroy_process_group()
'''


@pytest.mark.parametrize('order,expect', [('collective_first', 'WARNED'), ('model_first', 'QUIET')])
def test_stream_creation_order_against_the_first_collective(hip, order, expect):
    """Round 6 measured that a communicator built BEFORE the step's streams hold their hardware queues slows every later step by
    10-20 ms (profiles/r06_force_dp.txt): init_optimizers() therefore creates and binds the streams at once (prepare_streams), and
    a process that ran a collective before that is told so."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, '-c', _ORDER_SCRIPT, root, order], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    marks = [l for l in r.stdout.splitlines() if l in ('WARNED', 'QUIET')]      # (RCCL prints its banner on stdout as well)
    assert marks == [expect], (r.stdout[-500:], r.stderr[-1500:])
