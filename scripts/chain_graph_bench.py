"""GPU microbenchmark (not a test): the posterior chain dm_rssm_sequence_fwd and the imagination chain dm_dream_rollout
through the C-ABI with FIXED buffers, eager launches vs the library's linear hipGraph replay (csrc/chain_graph.hip),
on an otherwise idle GPU and ONE stream: host time per call, GPU time per call, launches per call.
    python scripts/chain_graph_bench.py [B] [T] [H]      (default 50 50 15: Atari-literal; 7 = the 8-rank shard)"""
import ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pydreamer_amd import config, hip as H
from pydreamer_amd.models import Dreamer

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
Hh = int(sys.argv[3]) if len(sys.argv) > 3 else 15
dev = torch.device('cuda', 0)
conf = config.atari_literal(batch_size=B, batch_length=T, imag_horizon=Hh)
torch.manual_seed(0)
model = Dreamer(conf).to(dev)
cell = model.wm.core.cell
D_, S, C, A, E = conf.deter_dim, conf.stoch_dim, conf.stoch_discrete, conf.action_dim, model.wm.encoder.out_dim
Z, F_ = S * C, D_ + S * C
N = T * B
shp = model.wm.shape(T, B, Hh)
ws = model.wm.workspace(shp, dev)
embed = torch.randn(N, E, device=dev)
action = F.one_hot(torch.randint(0, A, (N,), device=dev), A).float()
reset = torch.zeros(N, dtype=torch.uint8, device=dev)
h0, z0 = torch.zeros(B, D_, device=dev), torch.zeros(B, Z, device=dev)
u = torch.rand(N, S, device=dev)
acts = torch.empty(int(H.lib().dm_rssm_acts_floats(ctypes.byref(shp))), device=dev)
feat, post, prior = torch.empty(N, F_, device=dev), torch.empty(N, Z, device=dev), torch.empty(N, Z, device=dev)
idx = torch.empty(N, S, dtype=torch.int32, device=dev)
P = H.rssm_struct(cell.ordered())
actor_p = model.ac.actor.struct()
M = N
shp_d = model.wm.shape(1, M, Hh)
ws_d = model.wm.workspace(shp_d, dev)
u_act, u_prior = torch.rand(Hh, M, device=dev), torch.rand(Hh, M, S, device=dev)
feats, actions = torch.empty(Hh + 1, M, F_, device=dev), torch.empty(Hh, M, A, device=dev)
act_idx = torch.empty(Hh, M, dtype=torch.int32, device=dev)
a_acts = torch.empty(model.ac.actor.acts_floats(Hh * M), device=dev)
a_logits = torch.empty(Hh * M, model.ac.actor.out_dim, device=dev)


def fwd():
    H.call('dm_rssm_sequence_fwd', ctypes.byref(shp), H.fptr(embed), H.fptr(action), H.ptr(reset), H.fptr(h0), H.fptr(z0),
           H.fptr(u), None, ctypes.byref(P), H.fptr(acts), H.fptr(feat), H.fptr(post), H.fptr(prior), H.ptr(idx), H.ptr(ws),
           ws.numel(), H.stream())


def dream():
    H.call('dm_dream_rollout', ctypes.byref(shp_d), M, H.fptr(feat), ctypes.byref(P), ctypes.byref(actor_p), H.fptr(u_act),
           H.fptr(u_prior), H.fptr(feats), H.fptr(actions), H.ptr(act_idx), H.fptr(a_acts), H.fptr(a_logits), H.ptr(ws_d),
           ws_d.numel(), H.stream())


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e3 * t_host / reps, e0.elapsed_time(e1) / reps


out = dict(B=B, T=T, H=Hh)
side = torch.cuda.Stream(dev)
for name, fn in (('rssm_sequence_fwd', fwd), ('dream_rollout', dream)):
    for stream_name, ctx in (('null stream', None), ('side stream', side)):
        res = {}
        for mode in (0, 1):
            H.lib().dm_chain_graph_enable(mode)
            H.call('dm_chain_graph_reset')
            if ctx is None:
                res['graph' if mode else 'eager'] = timeit(fn)
            else:
                with torch.cuda.stream(ctx):
                    res['graph' if mode else 'eager'] = timeit(fn)
        out[f'{name} / {stream_name}'] = {k: dict(host_ms=round(v[0], 3), gpu_ms=round(v[1], 3)) for k, v in res.items()}
H.lib().dm_chain_graph_enable(1)
print(json.dumps(out, indent=1))
