"""How far does the host run ahead of the GPU inside one gradient step (diagnostic)?  For a few steady-state steps: the
host time at marked points of the step and the GPU time at which an event recorded at that point completes, on one clock
(both relative to a synchronised origin), plus every C-ABI call that took the host more than 0.5 ms and the start/end of the
launcher thread's jobs.  usage: python scripts/host_lead.py [f32|bf16] [steps=4] [batch columns=50]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pydreamer_amd import config, hip
from pydreamer_amd import models as M

dtype = sys.argv[1] if len(sys.argv) > 1 else 'f32'
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cols = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
conf = config.atari_literal(amp=(dtype == 'bf16'), **({'batch_size': cols} if cols else {}))
torch.manual_seed(0)
model = M.Dreamer(conf).to(dev)
opts = model.init_optimizers(conf.adam_lr, conf.adam_lr_actor, conf.adam_lr_critic, conf.adam_eps)
ring = bench.make_ring(conf, conf.batch_size, 0, conf.batch_size, 2, dev, 1234)
noise = bench.GlobalNoise(conf, conf.batch_size, 0, conf.batch_size, dev, 777)
state = {'s': model.init_state(conf.batch_size)}
log = []            # (host_t, label, event or None)
T0 = [0.0]


def mark(label, stream=None, ev=True):
    e = None
    if ev:
        e = torch.cuda.Event(enable_timing=True)
        e.record(stream if stream is not None else torch.cuda.current_stream())
    log.append((time.perf_counter() - T0[0], label, e))


slow = []
orig_call = hip.call


def timed(name, *a):
    t = time.perf_counter()
    r = orig_call(name, *a)
    d = time.perf_counter() - t
    if d > 5e-4:
        slow.append((t - T0[0], d, name))
    return r


orig_submit = M._Overlap.submit


def submit(self, stream, wait_event, fn):
    label = 'wm_bwd' if stream is self.s_wm else 'ac_bwd'
    t_sub = time.perf_counter() - T0[0]

    def fn2():
        t = time.perf_counter() - T0[0]
        e_s = torch.cuda.Event(enable_timing=True)
        e_s.record(torch.cuda.current_stream())        # completes when the stream has passed the job's wait_event: the GPU-side start
        r = fn()
        log.append((t, f'job {label} start (submitted {1e3 * t_sub:.2f})', e_s))
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        log.append((time.perf_counter() - T0[0], f'job {label} enqueued', e))
        return r
    return orig_submit(self, stream, wait_event, fn2)


def step(i, probe):
    if probe:
        mark(f'--- step {i} start')
    losses, new_state, metrics, tensors, _ = model.training_step(ring[i % 2], state['s'], noise=noise.draw())
    state['s'] = new_state
    if probe:
        mark('training_step returned')
    for opt in opts:
        opt.zero_grad()
    for k, loss in enumerate(losses):
        loss.backward()
        if probe:
            mark(f'backward {k} joined')
    model.grad_clip(conf.grad_clip, conf.grad_clip_ac)
    for opt in opts:
        opt.step()
    if probe:
        mark('optimizer enqueued')


for i in range(6):
    step(i, False)
torch.cuda.synchronize()
hip.call = timed
M.H.call = timed
M._Overlap.submit = submit
T0[0] = time.perf_counter()
e0 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(nsteps):
    step(6 + i, True)
t_host = time.perf_counter() - T0[0]
torch.cuda.synchronize()
t_all = time.perf_counter() - T0[0]
print(f'{dtype}: {nsteps} steps, host done at {1e3 * t_host:.2f} ms, GPU done at {1e3 * t_all:.2f} ms ({1e3 * t_all / nsteps:.2f} ms/step)')
print('  host ms   gpu ms   lead   what')
for t, label, e in sorted(log, key=lambda x: x[0]):
    if e is not None:
        g = e0.elapsed_time(e)
        print(f'{1e3 * t:9.2f} {g:8.2f} {g - 1e3 * t:6.2f}   {label}')
    else:
        print(f'{1e3 * t:9.2f}                   {label}')
print('C-ABI calls that held the host > 0.5 ms:')
for t, d, name in slow:
    print(f'{1e3 * t:9.2f}  {1e3 * d:6.2f} ms  {name}')
