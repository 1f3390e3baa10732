"""grad-steps/s of the DreamerV2 gradient step on MI355X (BASELINE.json metric), 1..N GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one synthetic replay batch: Dreamer.training_step (world model fwd, dream,
actor-critic fwd) -> zero_grad -> 4 x backward -> grad_clip -> 4 x AdamW, i.e. train.py:165-198 of the reference.
Workload: BASELINE.json configs[1] "Atari defaults" at B=50,T=50,H=15, deter=600, stoch=32x32, fp32, offline synthetic
replay already resident in HBM (a ring of pre-generated batches).  With N>1 the GLOBAL batch stays 50 and is sharded on
the batch axis (7,7,6,6,6,6,6,6 at N=8) with one RCCL all-reduce per optimizer group -> "scaling": "strong".

Other workloads (diagnostic lines, marked as such): --workload atari-native = pydreamer's own `defaults+atari` (B=32, T=48,
deter 1024: what the reference's README measured), --workload dmc = BASELINE configs[4] at one GPU (defaults+dmc, B=T=50).

The JSON line also carries
  h2d_included : the same step fed from HOST memory through pydreamer_amd.replay.DeviceRing (pinned uint8 frames, copies
                 prefetched behind the forward) - SURVEY 8(d) asks for it next to `value`, which keeps the replay resident in HBM.
  roofline     : the GEMM kernel family (every dense contraction of the step runs on it), algorithmic 2MNK flops per
                 launch / HIP-event launch durations recorded on the launch stream in a profiled pass of the same steps
                 that directly follows the timed region (events are kept out of the timed region so `value` is unperturbed);
                 peak = 157.3 TFLOP/s fp32 MFMA (MI355X_MICROARCH.md).
  cpu_baseline : oracle/dreamer_oracle.py (torch CPU restatement pinned to the reference by goldens, kind "port") timed
                 on this box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

METRIC = 'grad-steps/sec (world-model+AC) at B=50,T=50,H=15, 64×64 obs, 1/2/4/8 GPU'
ORACLE_VS_REF = next((f for f in ('r05_oracle_vs_reference.json', 'r04_oracle_vs_reference.json', 'r03_oracle_vs_reference.json', 'r02_oracle_vs_reference.json')
                      if os.path.exists(os.path.join(ROOT, 'profiles', f))), 'r04_oracle_vs_reference.json')


def make_ring(conf, b_global, lo, hi, n_batches, device, seed):
    """Offline synthetic replay: uint8 frames -> x/255-0.5 CHW float, one-hot actions, tanh rewards.
    Every rank draws the GLOBAL batch (b_global columns) from the same seed and keeps its columns [lo, hi) - like the sampler
    uniforms (GlobalNoise) - so an N-rank job steps on exactly the batch the 1-rank job steps on (sharded ==
    unsharded); only one batch of the global layout is alive at a time."""
    g = torch.Generator(device=device).manual_seed(seed)
    T, A = conf.batch_length, conf.action_dim
    ring = []
    for i in range(n_batches):
        u8 = torch.randint(0, 256, (T, b_global, conf.image_size, conf.image_size, conf.image_channels), generator=g,
                           device=device, dtype=torch.uint8)
        image = u8[:, lo:hi].float().div_(255.0).sub_(0.5).permute(0, 1, 4, 2, 3).contiguous()
        del u8
        if conf.actor_dist == 'onehot':
            act = torch.randint(0, A, (T, b_global), generator=g, device=device)
            action = torch.nn.functional.one_hot(act[:, lo:hi], A).float()
        else:                                          # continuous control (DMC): U(-1, 1) actions
            action = (torch.rand(T, b_global, A, generator=g, device=device) * 2 - 1)[:, lo:hi].contiguous()
        reward = torch.tanh(torch.randn(T, b_global, generator=g, device=device))[:, lo:hi].contiguous()
        terminal = (torch.rand(T, b_global, generator=g, device=device) < 0.005).float()[:, lo:hi].contiguous()
        reset = torch.zeros(T, b_global, dtype=torch.bool, device=device)
        reset[0] = torch.rand(b_global, generator=g, device=device) < (1.0 / 200)
        if i == 0:
            reset[0, 0] = True
        ring.append(dict(image=image, action=action, reward=reward, terminal=terminal, reset=reset[:, lo:hi].contiguous()))
    return ring


def make_host_ring(conf, b_local, n_batches, seed):
    """The same synthetic replay as numpy batches in HOST memory, in the replay's native layout (uint8 (T,B,H,W,C) frames):
    what DeviceRing stages through pinned memory for the H2D-included leg."""
    import numpy as np
    rs = np.random.RandomState(seed)
    T, A = conf.batch_length, conf.action_dim
    ring = []
    for i in range(n_batches):
        act = rs.randint(0, A, (T, b_local))
        reset = np.zeros((T, b_local), bool)
        reset[0] = rs.rand(b_local) < 1.0 / 200
        ring.append(dict(image=rs.randint(0, 256, (T, b_local, conf.image_size, conf.image_size, conf.image_channels)).astype(np.uint8),
                         action=np.eye(A, dtype=np.float32)[act], reward=np.tanh(rs.randn(T, b_local)).astype(np.float32),
                         terminal=(rs.rand(T, b_local) < 0.005).astype(np.float32), reset=reset))
    return ring


class GlobalNoise:
    """Sampler uniforms drawn in the GLOBAL batch layout with a generator every rank seeds identically, then sliced to the
    rank's batch columns: a sharded run draws, column for column, what the unsharded run draws."""

    def __init__(self, conf, B, lo, hi, device, seed):
        self.g = torch.Generator(device=device).manual_seed(seed)
        self.T, self.B, self.lo, self.hi, self.dev = conf.batch_length, B, lo, hi, device
        self.S, self.H, self.A = conf.stoch_dim, conf.imag_horizon, conf.action_dim
        self.onehot = conf.actor_dist == 'onehot'
        self.gauss = not conf.stoch_discrete

    def draw(self):
        T, B, S, H, A, lo, hi = self.T, self.B, self.S, self.H, self.A, self.lo, self.hi
        lat = (lambda *sh: torch.randn(*sh, generator=self.g, device=self.dev)) if self.gauss else \
              (lambda *sh: torch.rand(*sh, generator=self.g, device=self.dev))
        n = dict(u_post=lat(T, B, S)[:, lo:hi], u_prior=lat(H, T, B, S)[:, :, lo:hi].reshape(H, -1, S))
        if self.onehot:
            n['u_act'] = torch.rand(H, T, B, generator=self.g, device=self.dev)[:, :, lo:hi].reshape(H, -1)
        else:
            n['eps_act'] = torch.randn(H, T, B, A, generator=self.g, device=self.dev)[:, :, lo:hi].reshape(H, -1, A)
        return n


def _effective_cores():
    """Host cores this process may really use: affinity mask capped by a cgroup CPU quota if there is one."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        q = open('/sys/fs/cgroup/cpu.max').read().split()
        if q[0] != 'max':
            cores = max(1, min(cores, int(float(q[0]) / float(q[1]))))
    except Exception:
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                cores = max(1, min(cores, quota // period))
        except Exception:
            pass
    return cores


def cpu_baseline_worker(sample_batch, threads):
    """Runs in a subprocess (see cpu_baseline): oracle grad steps on the FULL workload (all 50 batch columns, T=50, H=15),
    one warm-up step excluded, >= 5 timed steps (about 30 s of CPU work on 16 cores; at most 8 steps / 75 s)."""
    from oracle import dreamer_oracle as O
    torch.set_num_threads(threads)
    full = O.atari_literal_conf()
    conf = O.atari_literal_conf(batch_size=sample_batch)
    model = O.OracleDreamer(conf, O.make_params(conf))
    model.init_optimizers()
    state = model.init_state(conf.batch_size)
    obs = O.preprocess(O.synthetic_batch(conf), conf)
    noise = O.make_noise(conf)

    def one(state):
        losses, state, *_ = model.training_step(obs, state, noise)
        model.backward_clip_step(losses)
        return state
    tw = time.perf_counter()
    state = one(state)                       # warm-up (allocator, thread pools, first-touch) - excluded
    warm = time.perf_counter() - tw
    n, t0 = 0, time.perf_counter()
    while True:
        state = one(state)
        n += 1
        el = time.perf_counter() - t0
        if n >= 5 and (el >= 25.0 or n >= 8):       # >= 5 timed steps (three was thin), about 30 s of CPU work
            break
        if el >= 75.0:
            break
    frac = sample_batch / full.batch_size
    print(json.dumps(dict(value=(n / el) * frac, unit='grad-steps/s', cores=threads, kind='port',
                          sample=f'{n} timed grad step(s) (fwd + 4 bwd + clip + 4 AdamW) of oracle/dreamer_oracle.py on {sample_batch} of the '
                                 f'{full.batch_size} batch columns (T=50, H=15, same model) in {el:.1f} s with {threads} torch threads; one '
                                 f'warm-up step ({warm:.1f} s) excluded' +
                                 ('' if frac == 1.0 else f'; scaled by {sample_batch}/{full.batch_size} to full-batch grad-steps/s'))))


def cpu_baseline(sample_batch=50, threads_cap=32, timeout_s=260):
    """Oracle (test infrastructure, kind "port") as the CPU baseline, in a subprocess with a hard timeout so a pathological
    host (thread oversubscription cost 889 s for one step in the first run of round 1) can never stall the bench; returns a
    dict with value=None and the reason if it does not finish."""
    import subprocess
    cores = _effective_cores()
    threads = max(1, min(cores, threads_cap))
    cmd = [sys.executable, os.path.abspath(__file__), '--cpu-baseline-worker', str(sample_batch), str(threads)]
    env = dict(os.environ, OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), HIP_VISIBLE_DEVICES='')
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env, cwd=ROOT)
        line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
        res = json.loads(line)
        res['host_cores'] = cores
        # representativeness of the port: oracle / reference wall time per grad step, MEASURED in the build
        # container by oracle/time_vs_reference.py against the reference's own loop and committed under profiles/
        try:
            with open(os.path.join(ROOT, 'profiles', ORACLE_VS_REF)) as f:
                m = json.load(f)
            res['oracle_over_reference_time'] = m['oracle_over_reference_time']
            res['representative'] = bool(abs(m['oracle_over_reference_time'] - 1.0) <= 0.10)      # within +-10 %
            # what the reference's own Python loop would score on these cores: the port's rate x (port time / reference time)
            res['reference_equivalent'] = res['value'] * m['oracle_over_reference_time'] if res.get('value') else None
            res['oracle_over_reference_source'] = (f"profiles/{ORACLE_VS_REF}: oracle {m['oracle_s_per_step']:.2f} s vs reference "
                                                   f"{m['reference_s_per_step']:.2f} s per step, {m['batch_columns']} columns, {m['threads']} threads, "
                                                   f"build container")
        except (OSError, KeyError, ValueError):
            res['oracle_over_reference_time'] = None
        return res
    except Exception as e:       # timeout or crash: report, never hang
        return dict(value=None, unit='grad-steps/s', cores=threads, kind='port', host_cores=cores,
                    sample=f'oracle sample of {sample_batch}/50 batch columns did not finish within {timeout_s} s ({type(e).__name__})')


def csrc_sha():
    """Fingerprint of the kernel sources: PMC traffic files carry it, so a stale file is refused instead of silently reported."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'pydreamer_amd', 'csrc')
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.h')):
            with open(os.path.join(d, name), 'rb') as f:
                h.update(name.encode())
                h.update(f.read())
    return h.hexdigest()[:16]


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == '--cpu-baseline-worker':
        return cpu_baseline_worker(int(sys.argv[2]), int(sys.argv[3]))
    if len(sys.argv) >= 2 and sys.argv[1] == '--csrc-sha':
        return print(csrc_sha())
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--ring', type=int, default=8)
    ap.add_argument('--reps', type=int, default=3, help='the timed region of exactly --steps steps (barrier + synchronise on both sides) is repeated this '
                    'many times back to back; `value` / `ms_per_step` are the MEDIAN region (max over ranks each), min / max are printed beside it')
    ap.add_argument('--prof-steps', type=int, default=2)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pmc-json', default='',
                    help='per-kernel HBM traffic from two rocprofv3 PMC passes of THIS command (scripts/pmc_traffic.py); it carries a '
                         'fingerprint of the kernel sources and is refused (traffic = null) when that differs from the tree')
    ap.add_argument('--emulate-world', type=int, default=0, help='diagnostic only: run rank 0''s batch shard of an N-rank job on one GPU without the all-reduce (per-rank compute time at N GPUs); the line is marked invalid as a metric')
    ap.add_argument('--emulate-rank', type=int, default=0, help='with --emulate-world N: which rank''s shard (uneven splits: 50 columns over 8 ranks are 7,7,6,6,6,6,6,6)')
    ap.add_argument('--force-dp', action='store_true', help='diagnostic on a 1-GPU box: a ONE-rank nccl (= RCCL) group, the data-parallel code path switched on - '
                    'every optimizer group is all-reduced (over one rank) exactly where an N-rank job does it; DM_DP_NATIVE=1 selects the library\'s own '
                    'dm_allreduce_grads.  What the collectives and their stream cost the step, without a second GPU; the line is marked invalid as a metric')
    ap.add_argument('--dtype', choices=('f32', 'bf16'), default='f32', help='f32 = BASELINE configs[1] (the metric); bf16 = configs[2]: '
                    'conf.amp, GEMM operands in bf16 with fp32 accumulation and storage')
    ap.add_argument('--no-overlap', action='store_true', help='run all backward passes on one stream (A/B switch)')
    ap.add_argument('--pipeline', action='store_true', help='actor / critic clip + AdamW on the actor-critic stream behind their backward pass '
                    '(Dreamer.pipeline_ac_optimizer = True, OFF by default in the product: a trainer must then not touch actor / critic .grad between '
                    'backward() and step()).  The default line runs the product default; -0.1 ms at 50 columns, -0.3..-0.55 ms on 7..25-column shards')
    ap.add_argument('--no-pipeline', action='store_true', help='(kept for old command lines: the default now)')
    ap.add_argument('--workload', choices=('atari-literal', 'atari-native', 'dmc'), default='atari-literal',
                    help="atari-literal = BASELINE configs[1] (the metric); atari-native = pydreamer's own defaults+atari (B=32, T=48, deter 1024; "
                         "README.md:90-97); dmc = configs[4] at one GPU (defaults+dmc, actor_grad=reinforce, action_dim 6, B=T=50); the last two are diagnostic lines")
    ap.add_argument('--no-h2d-leg', action='store_true', help='skip the H2D-included leg (DeviceRing from host memory)')
    ap.add_argument('--h2d-steps', type=int, default=40)
    ap.add_argument('--shape-table', default='', help='write the per-shape GEMM table of the profiled pass to this file (diagnostic)')
    args = ap.parse_args()
    if not args.pmc_json:       # the committed counter summary of the same command (fp32 / bf16 step), if its fingerprint matches the tree
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r[0-9][0-9]_pmc_traffic' + ('_bf16' if args.dtype == 'bf16' else '') + '.json')))
        args.pmc_json = cands[-1] if cands else os.path.join(ROOT, 'profiles', 'none.json')      # the newest round's; refused below if its fingerprint is stale

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch with python -m torch.distributed.run --nproc-per-node N for --gpus N > 1')
    import torch.distributed as dist
    # DM_BENCH_ONE_DEVICE=1 (smoke test of the N > 1 code path on a 1-GPU box, tests/test_gpu_dist.py): every rank uses
    # cuda:0 and the ranks talk over gloo; the line is marked invalid as a metric
    one_device = bool(os.environ.get('DM_BENCH_ONE_DEVICE')) and world > 1
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)                   # before the process group: RCCL binds its communicator to the current device
    dev = torch.device('cuda', local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not one_device and torch.cuda.device_count() < world:
            raise SystemExit(f'--gpus {world} needs {world} visible GPUs (one process per GPU over RCCL); this box has '
                             f'{torch.cuda.device_count()} (DM_BENCH_ONE_DEVICE=1 runs the N-rank code path on one device over gloo: a smoke mode, not a metric)')
        dist.init_process_group('gloo' if one_device else 'nccl', rank=rank, world_size=world)
        if not one_device and (dist.get_backend() != 'nccl' or dist.get_world_size() != args.gpus):
            raise SystemExit(f'--gpus {args.gpus}: expected {args.gpus} ranks over nccl (= RCCL), got {dist.get_world_size()} over {dist.get_backend()}')
    elif args.force_dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        # (DM_BENCH_FORCE_BACKEND=gloo: the control plane over gloo, so that with DM_DP_NATIVE=1 the library's communicators are the
        #  only RCCL communicators of the process)
        dist.init_process_group(os.environ.get('DM_BENCH_FORCE_BACKEND', 'nccl'), rank=0, world_size=1)

    from pydreamer_amd import config, hip
    from pydreamer_amd import dist as DP
    from pydreamer_amd.models import Dreamer
    hip.call('dm_device_check')
    lds_clk = (ctypes.c_ulonglong * 16)()
    hip.lib().dm_rssm_lds_prof(lds_clk, 1)             # (allocates and zeroes the persistent kernel's phase clocks: non-zero at the end = it ran)
    if one_device:
        # the persistent posterior kernel needs every CU of the device at once (one workgroup per CU, resident together): two
        # PROCESSES launching theirs on the same device can each hold part of the chip and spin until their poll bounds trip.
        # One process per GPU - the only supported deployment - never sees this; the one-device smoke mode takes the launch chain.
        hip.lib().dm_rssm_lds_enable(0)

    def make_conf(**kw):
        if args.workload == 'atari-native':
            return config.load_config('defaults', 'atari', action_dim=18, **kw)
        if args.workload == 'dmc':
            return config.load_config('defaults', 'dmc', **{**dict(action_dim=6, actor_grad='reinforce', batch_size=50, batch_length=50), **kw})
        return config.atari_literal(**kw)
    gconf = make_conf()
    B = gconf.batch_size
    lo, hi = DP.shard_bounds(B, world, rank)
    if args.emulate_world > 1 and world == 1:
        lo, hi = DP.shard_bounds(B, args.emulate_world, args.emulate_rank)
    conf = make_conf(batch_size=hi - lo, amp=(args.dtype == 'bf16'))
    # algorithmic TFLOP per grad step (2 MAC, dense as the reference executes, backward = 2x forward)
    alg_tflop = {'atari-literal': 2.76, 'atari-native': 2.00, 'dmc': 4.83}[args.workload]
    torch.manual_seed(0)                               # identical replicas on every rank
    model = Dreamer(conf).to(dev)
    model.overlap_backward = not args.no_overlap
    model.pipeline_ac_optimizer = bool(args.pipeline) and not (args.no_pipeline or args.no_overlap)
    opts = model.init_optimizers(conf.adam_lr, conf.adam_lr_actor, conf.adam_lr_critic, conf.adam_eps)
    if os.environ.get('DM_BENCH_DP_IDLE') == '1' and args.force_dp:
        # experiment: the process group exists and has run one collective (its communicator and stream are there), but the step issues none
        dist.all_reduce(torch.zeros(8, device=dev))
        torch.cuda.synchronize()
    else:
        DP.attach(opts, hi - lo, B, model=model, force=args.force_dp)      # the B_r/B weight rides in the backward kernels' scale arguments
    ring = make_ring(conf, B, lo, hi, args.ring, dev, 1234)      # the global batch, this rank's columns
    noise = GlobalNoise(conf, B, lo, hi, dev, 777)     # global-layout sampler uniforms, the rank's columns sliced out
    state = {'s': model.init_state(hi - lo)}

    def step(i, eager=False):
        obs = ring[i % len(ring)]
        losses, new_state, metrics, tensors, _ = model.training_step(obs, state['s'], noise=noise.draw())
        state['s'] = new_state                          # keep_state (train.py:177-178)
        for opt in opts:
            opt.zero_grad()
        for loss in losses:
            loss.backward()
        model.grad_clip(conf.grad_clip, conf.grad_clip_ac)
        for opt in opts:
            opt.step()
        return metrics

    main_prio = os.environ.get('DM_MAIN_PRIO')            # experiment switch: run the caller's stream at another priority
    if main_prio is not None:
        torch.cuda.set_stream(torch.cuda.Stream(dev, priority=int(main_prio)))
    main_res = int(os.environ.get('DM_MAIN_RESERVE_CUS', '0'))    # experiment: the caller's stream without the first k CUs of every 32
    if main_res > 0:
        torch.cuda.set_stream(hip.cu_masked_stream([0xFFFFFFFF ^ ((1 << main_res) - 1)] * 8, dev))
    for i in range(args.warmup):
        step(i)
    # the timed region: EXACTLY --steps steps between barrier + synchronise, max over ranks; repeated --reps times back to back
    # (state, optimizer moments and replay ring carry on), the median region is the reported one
    regions, t_enq = [], []
    n_done = args.warmup
    for rep in range(max(1, args.reps)):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            metrics = step(n_done + i)
        t_enq.append(time.perf_counter() - t0)          # host done enqueuing; the GPU may still be running
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        n_done += args.steps
        per = [el]
        if world > 1:
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            per_rank = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(per_rank, t)
            per = [float(x.item()) for x in per_rank]
        regions.append(per)
    order = sorted(range(len(regions)), key=lambda r: max(regions[r]))
    mid = order[len(order) // 2]                        # the median region (by its max over ranks)
    elapsed = max(regions[mid])
    t_enqueued = t_enq[mid]
    rank_ms = [1e3 * x / args.steps for x in regions[mid]]
    region_ms = [1e3 * max(r) / args.steps for r in regions]
    loss_model = float(metrics['loss_model'])
    # What the host needs to ENQUEUE a step when nothing holds it back: three more steps right after the synchronise (empty
    # queues).  `host_enqueue_ms_per_step` above is taken over the timed region, where the runtime's queue back-pressure
    # makes the launching threads wait for the GPU once they are a few steps ahead - it tracks the GPU, not the host's cost.
    th = time.perf_counter()
    for i in range(3):
        step(n_done + i)
    host_free_ms = 1e3 * (time.perf_counter() - th) / 3
    torch.cuda.synchronize()
    dist_info = None
    if world > 1 or args.force_dp:      # (--force-dp: the same record over a ONE-rank RCCL group)
        # per-rank numbers (before the MAX) and the stand-alone cost of the step's all-reduces on this fabric
        mine = torch.tensor([1e3 * t_enqueued / args.steps], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        host_ms = [float(x.item()) for x in allr]
        ar_ms = {}
        for name, opt in zip(('wm', 'probe', 'actor', 'critic'), opts):
            buf = torch.zeros_like(opt.flat_grad)
            for _ in range(2):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                dist.all_reduce(buf)
            e1.record()
            torch.cuda.synchronize()
            ar_ms[name] = dict(bytes=4 * buf.numel(), ms=e0.elapsed_time(e1) / 5)
        try:
            rccl = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        # replicas: every rank must hold bit-identical parameters after the same steps (fp64 checksum of each group's flat
        # parameter buffer, gathered); and the loss of the GLOBAL batch = sum_r (B_r / B) loss_r, comparable with a 1-rank run
        chk = torch.tensor([float(o.flat_param.double().sum()) for o in opts] +
                           [float(o.flat_param.double().abs().sum()) for o in opts], device=dev, dtype=torch.float64)
        allc = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(allc, chk)
        replicas_identical = all(bool(torch.equal(c, allc[0])) for c in allc)
        gl = torch.tensor([loss_model * (hi - lo) / B], device=dev, dtype=torch.float64)
        dist.all_reduce(gl)
        shard_sizes = [DP.shard_bounds(B, world, r)[1] - DP.shard_bounds(B, world, r)[0] for r in range(world)]
        # the persistent posterior kernel (csrc/rssm_lds.hip; default for <= 32-column shards): did it run on every rank, and did
        # any rank's kernel give up in a spin loop (dm_rssm_lds_status, sticky)?
        hip.lib().dm_rssm_lds_prof(lds_clk, 0)
        pk = torch.tensor([1.0 if sum(lds_clk) > 0 else 0.0, float(hip.lib().dm_rssm_lds_status())], device=dev, dtype=torch.float64)
        allp = [torch.zeros_like(pk) for _ in range(world)]
        dist.all_gather(allp, pk)
        dist_info = dict(world_size=dist.get_world_size(), backend=dist.get_backend(), rccl_version=rccl,
                         persistent_posterior_kernel_ran=[bool(x[0].item()) for x in allp],
                         rssm_lds_status=[int(x[1].item()) for x in allp],
                         shard_columns=shard_sizes, replicas_identical=replicas_identical,
                         param_checksum_rank0=[float(x) for x in allc[0].tolist()], loss_model_global=float(gl.item()),
                         ms_per_step_per_rank=rank_ms, host_enqueue_ms_per_rank=host_ms, allreduce_standalone=ar_ms,
                         note='all-reduce of each optimizer group\'s flat fp32 gradient buffer, timed alone (in the step the world-model '
                              'group\'s is overlapped with the actor / critic backward)')

    # H2D-included leg: the same step fed from host memory through the DeviceRing (pinned uint8 frames, copies prefetched behind the forward)
    h2d = None
    if world == 1 and not args.no_h2d_leg and args.emulate_world <= 1:
        from pydreamer_amd.replay import DeviceRing
        import itertools
        host_ring = make_host_ring(conf, hi - lo, 4, 4321)
        dring = DeviceRing(itertools.cycle(host_ring), dev, depth=4)
        hstate = model.init_state(hi - lo)

        def hstep():
            nonlocal hstate
            obs = dring.next()
            losses, hstate, *_ = model.training_step(obs, hstate, noise=noise.draw())
            if not os.environ.get('DM_RING_NO_PREFETCH'):      # (A/B switch: the copies then sit in front of the next step)
                dring.prefetch()   # the next batch's H2D copies, behind this step's forward on the caller's stream (idle from here on)
            for opt in opts:
                opt.zero_grad()
            for loss in losses:
                loss.backward()
            model.grad_clip(conf.grad_clip, conf.grad_clip_ac)
            for opt in opts:
                opt.step()
        for _ in range(8):                     # (the ring's pinned slots are allocated and first filled during these)
            hstep()
        torch.cuda.synchronize()
        th = time.perf_counter()
        for _ in range(args.h2d_steps):
            hstep()
        torch.cuda.synchronize()
        th = time.perf_counter() - th
        dring.close()
        nbytes = sum(v.nbytes for v in host_ring[0].values())
        h2d = dict(value=args.h2d_steps / th, unit='grad-steps/s', ms_per_step=1e3 * th / args.h2d_steps, steps=args.h2d_steps,
                   host_bytes_per_step=nbytes, note='uint8 (T,B,64,64,3) frames + actions / rewards / flags from pinned host memory through '
                   'pydreamer_amd.replay.DeviceRing (depth 4; copies prefetched on the caller\'s stream behind the forward, no copy stream); x/255-0.5 and HWC->CHW happen inside the first conv\'s loader')

    # profiled pass: HIP events around every GEMM launch on the launch stream (same steps, right after the timed region)
    roof = None
    if args.prof_steps > 0:
        hip.call('dm_prof_begin', 8192 * args.prof_steps)
        model.overlap_backward = False      # one stream, one launcher thread: per-launch events then time that launch alone
        model.pipeline_ac_optimizer = False      # ... and no optimizer tail of the previous step beside the next forward
        if hasattr(model, 'join_optimizers'):
            model.join_optimizers()
        torch.cuda.synchronize()
        for i in range(args.prof_steps):
            step(n_done + 3 + i, eager=True)   # per-launch events need real launches, not a replay
        torch.cuda.synchronize()
        if args.shape_table:        # per-shape table of the profiled pass (diagnostic, stderr / file)
            cap = 8192 * args.prof_steps
            rows = (ctypes.c_double * (8 * cap))()
            nr = hip.lib().dm_prof_rows(rows, cap)
            agg = {}
            for i in range(nr):
                r = rows[8 * i:8 * i + 8]
                key = tuple(int(x) for x in r[:6])
                a = agg.setdefault(key, [0, 0.0, 0.0])
                a[0] += 1; a[1] += r[6]; a[2] += r[7]
            with open(args.shape_table, 'w') as f:
                f.write('kind M N K split flags launches/step ms/step us/launch TF/s\n')
                for key, a in sorted(agg.items(), key=lambda kv: -kv[1][2]):
                    f.write(' '.join(str(x) for x in key) + f' {a[0] / args.prof_steps:.1f} {a[2] / args.prof_steps:.3f} '
                            f'{1e3 * a[2] / a[0]:.1f} {a[1] / (a[2] * 1e-3) / 1e12 if a[2] > 0 else 0:.1f}\n')
        NK = 44
        out = (ctypes.c_double * (4 * NK))()
        n = hip.lib().dm_prof_end(out, NK)
        kinds = []
        names = {0: 'NT', 1: 'NN', 2: 'TN*', 3: 'TN'}
        tiles = ('128,128', '128,64', '64,64', '128,96', '96,128')
        for k in range(NK):
            cnt, fl, ms, by = out[4 * k], out[4 * k + 1], out[4 * k + 2], out[4 * k + 3]
            if cnt:
                # kinds: 0..19 register-staged tile kernel (tile*4 + layouts), 20..22 panel / whole-MLP kernels, 24+ the same tiles on
                # gemm_dma_kernel (csrc/gemm.hip dm_gemm_launch)
                # (bf16 mode: the same tile / layout runs as gemm_pipe_kernel, or gemm_h_kernel when both operands have bf16 twins)
                kk = k - 24 if k >= 24 else k
                if k >= 24:
                    kname, lay = f'gemm_dma_kernel<{tiles[kk >> 2]},{(kk >> 1) & 1},{kk & 1}>', names[kk & 3]
                elif k < 20:
                    kname = f"{'gemm_pipe_kernel' if args.dtype == 'bf16' else 'gemm_f32_kernel'}<{tiles[k >> 2]},{(k >> 1) & 1},{k & 1}>"
                    lay = names[k & 3]
                else:
                    kname = 'panel_linear_kernel<25,0,1' if k == 20 else 'panel_linear_kernel<25,1,2' if k == 21 else 'mlp_chain_fwd_kernel'
                    lay = 'row panel fwd' if k == 20 else 'row panel bwd' if k == 21 else 'whole-MLP forward, 16-row blocks'
                kinds.append(dict(kernel=kname,
                                  layout=lay, launches_per_step=cnt / args.prof_steps,
                                  avg_launch_us=1e3 * ms / cnt, gflop_per_step=fl / 1e9 / args.prof_steps,
                                  ms_per_step=ms / args.prof_steps, tflops=fl / (ms * 1e-3) / 1e12,
                                  alg_bytes_per_launch=by / cnt, alg_flops_per_launch=fl / cnt))
        tot_fl = sum(out[4 * k + 1] for k in range(NK))
        tot_ms = sum(out[4 * k + 2] for k in range(NK))
        dom = max(kinds, key=lambda d: d['ms_per_step'])
        peak = 157.3 if args.dtype == 'f32' else 2500.0       # dense MFMA peak of the operand type (MI355X_MICROARCH.md)
        # HBM traffic of the dominant kernel: measured in separate rocprofv3 PMC passes of this same command (FETCH_SIZE
        # and WRITE_SIZE cannot share a pass), summarised by scripts/pmc_traffic.py into profiles/ - read back here
        traffic, traffic_note, components = None, None, None
        try:
            with open(args.pmc_json) as f:
                pj = json.load(f)
            if args.workload != 'atari-literal':
                traffic_note = f"{os.path.relpath(args.pmc_json, ROOT)} was collected on the atari-literal workload, not on '{args.workload}': not applied"
            elif pj.get('csrc_sha') != csrc_sha():
                traffic_note = (f"{os.path.relpath(args.pmc_json, ROOT)} was collected for kernel sources {pj.get('csrc_sha')}, the tree is "
                                f"{csrc_sha()}: stale, refused")
            else:
                pmc = pj['kernels']
                key = dom['kernel'].replace('>', '') + ','        # "gemm_f32_kernel<64,64,0,0" + remaining template args
                keys = [key.replace(' ', '')] + ([key.replace(' ', '').replace('gemm_pipe_kernel', 'gemm_h_kernel')] if args.dtype == 'bf16' else [])
                hits = [v for k, v in pmc.items() if any(k.replace(' ', '').startswith(kk) for kk in keys)]
                if hits:
                    n_pmc = sum(h['launches'] for h in hits)
                    traffic = sum(h['hbm_bytes_per_launch'] * h['launches'] for h in hits) / max(n_pmc, 1)
                    traffic_note = (f"HBM bytes per launch, PMC FETCH_SIZE x2 + WRITE_SIZE in separate rocprofv3 passes of this command "
                                    f"({os.path.relpath(args.pmc_json, ROOT)}, kernel sources {pj['csrc_sha']}); whole step "
                                    f"{pj.get('total_gb_per_step')} GB")
                # per-component view: the step's heaviest kernels, each against the bound that applies to it - the
                # MFMA rate (live, from this run's profiled pass) for the tile kernels, the HBM rate (bytes and undisturbed
                # durations of the counter passes) for everything that streams
                comps = []
                stepsp = max(int(pj.get('steps_total', pj.get('steps_profiled', 1))), 1)
                for kname, v in (pmc.items() if all('avg_us' in v for v in list(pmc.values())[:1]) else []):
                    if not v.get('avg_us'):
                        continue
                    lps = v['launches'] / stepsp
                    tile = any(t in kname for t in ('gemm_f32_kernel', 'gemm_pipe_kernel', 'gemm_h_kernel', 'panel_linear', 'mlp_chain_fwd'))
                    chain = any(t in kname for t in ('skinny_gemm', 'rssm_lds', 'gru_gates', 'sample_onehot', 'st_softmax', 'z_embed'))
                    c = dict(kernel=kname[:60], launches_per_step=round(lps, 1), avg_us=round(v['avg_us'], 1), ms_per_step=round(lps * v['avg_us'] * 1e-3, 3),
                             hbm_gb_per_s=round(v['hbm_gb_per_s'], 1) if v.get('hbm_gb_per_s') else None,
                             hbm_frac=round(v['hbm_gb_per_s'] / 8000.0, 3) if v.get('hbm_gb_per_s') else None,
                             bound='mfma' if tile else ('latency (<= 64-row steps of the sequential chains: neither roof applies)' if chain else 'hbm'))
                    if tile:
                        k2 = kname.replace(' ', '').replace('panel_linear_bf16_kernel<25,1,', 'panel_linear_kernel<25,0,1,').replace(
                            'panel_linear_bf16_kernel<25,2,', 'panel_linear_kernel<25,1,2,')      # (bf16 panel kernels: <25, 1 = fwd | 2 = bwd, ...>)
                        hit = [d for d in kinds if k2.startswith(d['kernel'].replace('>', '').replace(' ', '') + ',') or
                               k2.startswith(d['kernel'].replace(' ', '')) or
                               (args.dtype == 'bf16' and k2.replace('gemm_h_kernel', 'gemm_pipe_kernel').startswith(d['kernel'].replace('>', '').replace(' ', '') + ','))]
                        if hit:
                            c.update(mfma_tflops=round(hit[0]['tflops'], 1), mfma_frac=round(hit[0]['tflops'] / peak, 3))
                    comps.append(c)
                comps.sort(key=lambda c: -c['ms_per_step'])
                components = comps[:16] or None
        except (OSError, KeyError, ValueError) as e:
            traffic_note = f'no PMC file ({type(e).__name__})' 
        roof = dict(bound='mfma', achieved=dom['tflops'], peak=peak, unit='TFLOP/s', frac=dom['tflops'] / peak, traffic=traffic,
                    traffic_unit=traffic_note,
                    pmc_fingerprint_matches=(traffic is not None) if args.workload == 'atari-literal' else None,
                    # not observed by THIS run: read back from the committed counter passes of the same command (fingerprint-checked)
                    traffic_source=('committed profile ' + os.path.relpath(args.pmc_json, ROOT)) if traffic is not None else None,
                    algorithmic_bytes_per_launch=dom.get('alg_bytes_per_launch'),
                    components=components,
                    components_note=('the 16 heaviest kernels of the step; launches, durations and HBM rates from the counter passes (dispatches '
                                     'serialised, single stream), MFMA rates from this run; bound = the roof that applies to the kernel (peak '
                                     f'{peak} TFLOP/s dense / 8000 GB/s)') if components else None,
                    kernel=dom['kernel'], avg_launch_us=dom['avg_launch_us'], launches_per_step=dom['launches_per_step'],
                    # the part holds ~2.07 GHz (not 2.4) under a full fp32 GEMM (in-kernel s_memtime / s_memrealtime probe and
                    # GRBM_GUI_ACTIVE: profiles/r05_gemm_lab.txt, r05_gemm_sq_counters.txt): `peak` stays nominal
                    **({'peak_at_measured_clock': round(157.3 * 2.07 / 2.4, 1), 'frac_of_peak_at_measured_clock': dom['tflops'] / (157.3 * 2.07 / 2.4),
                        'measured_clock_ghz': 2.07, 'measured_clock_source': 'profiles/r05_gemm_lab.txt (4096^3, in-kernel clock probe), profiles/r05_gemm_sq_counters.txt'}
                       if args.dtype == 'f32' else {}),
                    all_gemm=dict(tflops=tot_fl / (tot_ms * 1e-3) / 1e12, frac=tot_fl / (tot_ms * 1e-3) / 1e12 / peak,
                                  gflop_per_step=tot_fl / 1e9 / args.prof_steps, ms_per_step=tot_ms / args.prof_steps,
                                  launches_per_step=n / args.prof_steps),
                    kinds=kinds, note='HIP events on the launch stream around every GEMM launch, eager (un-graphed) pass of the same steps right after the timed region')

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.workload == 'atari-literal':
        cpu = cpu_baseline()

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        line = dict(metric=METRIC, value=args.steps / elapsed, unit='grad-steps/s', n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=ms, ms_per_step_min=min(region_ms), ms_per_step_median=ms,
                    ms_per_step_max=max(region_ms), timed_regions=len(region_ms), ms_per_step_regions=region_ms, higher_is_better=True, scaling='strong', vs_baseline=None,
                    dtype='f32' if args.dtype == 'f32' else 'bf16 (MFMA operands; fp32 accumulate, fp32 storage and non-GEMM math)', data='synthetic',
                    config=dict(workload={'atari-literal': 'atari-literal: defaults+atari, batch_size 50, batch_length 50, imag_horizon 15, '
                                         'deter_dim 600, stoch 32x32, hidden 1000, cnn_depth 48, action_dim 18, ',
                                          'atari-native': 'atari-native (pydreamer README.md:90-97): defaults+atari as shipped, batch_size 32, batch_length 48, imag_horizon 15, '
                                         'deter_dim 1024, stoch 32x32, hidden 1000, cnn_depth 48, action_dim 18, ',
                                          'dmc': 'dmc (BASELINE configs[4] at one GPU): defaults+dmc, batch_size 50, batch_length 50, imag_horizon 15, deter_dim 2048, '
                                         'action_dim 6, tanh_normal actor, actor_grad reinforce, '}[args.workload] + ('fp32' if args.dtype == 'f32' else 'amp/bf16') + '; '
                                         'fwd + 4 bwd + clip + 4 AdamW per step; replay resident in HBM',
                                global_batch=B, batch_length=conf.batch_length, imag_horizon=conf.imag_horizon,
                                parallelism=f'dp{world} (batch-sharded {[DP.shard_bounds(B, world, r)[1] - DP.shard_bounds(B, world, r)[0] for r in range(world)]})',
                                algorithmic_tflop_per_step=alg_tflop),
                    **({'INVALID_diagnostic_emulated_world': args.emulate_world} if args.emulate_world > 1 else {}),
                    **({'INVALID_diagnostic_forced_one_rank_dp': 'native dm_allreduce_grads' if os.environ.get('DM_DP_NATIVE') == '1' else 'torch.distributed nccl'}
                       if args.force_dp else {}),
                    **({'INVALID_smoke_all_ranks_on_one_device': True} if one_device else {}),
                    loss_model_last=loss_model,
                    param_checksum=[float(o.flat_param.double().sum()) for o in opts] + [float(o.flat_param.double().abs().sum()) for o in opts],
                    host_enqueue_ms_per_step=1e3 * t_enqueued / args.steps,
                    host_enqueue_unthrottled_ms_per_step=host_free_ms,
                    fp32_products='fp32 MFMA',
                    step_tflops=alg_tflop / (ms * 1e-3), step_frac_of_fp32_peak=alg_tflop / (ms * 1e-3) / 157.3,
                    h2d_included=h2d, distributed=dist_info, rccl_version=(dist_info or {}).get('rccl_version'),
                    **({} if args.workload == 'atari-literal' else {'INVALID_diagnostic_workload': args.workload}),
                    **({} if args.dtype == 'f32' else {'note_dtype': 'BASELINE configs[2] (mixed precision); the headline metric is the f32 line'}),
                    roofline=roof, cpu_baseline=cpu)
        print(json.dumps(line))
    if world > 1 or args.force_dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
