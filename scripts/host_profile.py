"""Host-side cost of one gradient step (diagnostic): cProfile over the enqueue path at a small per-rank batch."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd import config, hip
from pydreamer_amd.models import Dreamer
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
dev = torch.device('cuda', 0)
conf = config.atari_literal(batch_size=B)
torch.manual_seed(0)
model = Dreamer(conf).to(dev)
opts = model.init_optimizers(conf.adam_lr, conf.adam_lr_actor, conf.adam_lr_critic, conf.adam_eps)
ring = bench.make_ring(conf, B, 2, dev, 1)
state = model.init_state(B)

def step(i):
    global state
    losses, state, metrics, tensors, _ = model.training_step(ring[i % 2], state)
    for o in opts: o.zero_grad()
    for l in losses: l.backward()
    model.grad_clip(conf.grad_clip, conf.grad_clip_ac)
    for o in opts: o.step()

for i in range(5): step(i)
torch.cuda.synchronize()
# time spent inside the C library calls vs everything else
import ctypes
t_c = [0.0, 0]
orig = hip.call
def timed(name, *a):
    t = time.perf_counter(); r = orig(name, *a); t_c[0] += time.perf_counter() - t; t_c[1] += 1; return r
hip.call = timed
import pydreamer_amd.models as M, pydreamer_amd.optim as Oo
M.H.call = timed
t0 = time.perf_counter()
for i in range(10): step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f'B={B}: host enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step; inside C calls {1e3*t_c[0]/10:.2f} ms/step over {t_c[1]/10:.0f} calls')
M.H.call = orig
pr = cProfile.Profile(); pr.enable()
for i in range(5): step(i)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
