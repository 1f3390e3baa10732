"""Diagnostic: per-phase time of the persistent posterior-chain kernel (clock ticks of its workgroup 0, scaled to the measured total)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import dreamer_oracle as O
from pydreamer_amd import config, hip as H
from pydreamer_amd.models import Dreamer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T, D_, Hd, S, C, A, depth = 50, 600, 1000, 32, 32, 18, 8
oconf = O.make_conf(deter_dim=D_, hidden_dim=Hd, stoch_dim=S, stoch_discrete=C, cnn_depth=depth, action_dim=A, batch_size=B, batch_length=T)
conf = config.load_config('defaults', 'atari', **vars(oconf))
model = Dreamer(conf); model.load_state_dict(O.make_params(oconf, seed=4)); model = model.to('cuda')
cell = model.wm.core.cell
E, Z, F_ = 32 * depth, S * C, D_ + S * C
g = torch.Generator().manual_seed(12)
embed = torch.randn(T * B, E, generator=g).cuda()
action = F.one_hot(torch.randint(0, A, (T * B,), generator=g), A).float().cuda()
reset = (torch.rand(T * B, generator=g) < 0.02).to(torch.uint8).cuda()
h0, z0 = torch.tanh(torch.randn(B, D_, generator=g)).cuda(), torch.zeros(B, Z).cuda()
u = torch.rand(T * B, S, generator=g).cuda()
shp = model.wm.shape(T, B, 1)
ws = model.wm.workspace(shp, torch.device('cuda', 0))
P = H.rssm_struct(cell.ordered())
lib = H.lib()
lib.dm_rssm_persist_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
acts = torch.zeros(int(lib.dm_rssm_acts_floats(ctypes.byref(shp))), device='cuda')
feat, post, prior = (torch.zeros(T * B, n, device='cuda') for n in (F_, Z, Z))
idx = torch.zeros(T * B, S, dtype=torch.int32, device='cuda')
def run():
    H.call('dm_rssm_sequence_fwd', ctypes.byref(shp), H.fptr(embed), H.fptr(action), H.ptr(reset), H.fptr(h0), H.fptr(z0), H.fptr(u), None,
           ctypes.byref(P), H.fptr(acts), H.fptr(feat), H.fptr(post), H.fptr(prior), H.ptr(idx), H.ptr(ws), ws.numel(), H.stream())
for on in (1, 0):
    lib.dm_rssm_persist_enable(on)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.dm_rssm_persist_prof(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    out = (ctypes.c_ulonglong * 12)()
    lib.dm_rssm_persist_prof(out, 0)
    steps = 5 * (T - 1)
    names = ['z_embed', 'b1', 'gi+gh', 'b2', 'gates', 'b3', 'post_h', 'b4', 'post+sample', 'b5']
    total_us = e0.elapsed_time(e1) / 5 * 1e3
    print(f'B={B} persist={on}: {total_us:.0f} us per sequence call (T={T})')
    if on:      # tick frequency is not documented: scale the phases to the measured time of the persistent steps
        ticks = [out[i] / 5 for i in range(10)]
        scale = total_us * (T - 1) / T / max(sum(ticks), 1)
        print('   per step, us: ' + ', '.join(f'{n} {ticks[i] * scale / (T - 1):.1f}' for i, n in enumerate(names)))
