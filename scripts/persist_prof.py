"""Diagnostic: time of the posterior T loop (dm_rssm_sequence_fwd, T = 50, Atari-literal cell width) with the LDS-weight-stationary
persistent kernel (csrc/rssm_lds.hip) on and off, and the kernel's per-phase clock ticks (workgroup 0, 100 MHz wall clock).
    python scripts/persist_prof.py [B ...]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import dreamer_oracle as O
from pydreamer_amd import config, hip as H
from pydreamer_amd.models import Dreamer

T, D_, Hd, S, C, A, depth = 50, int(os.environ.get('DETER', 600)), 1000, 32, 32, 18, 8      # DETER=1024: pydreamer's shipped Atari cell
lib = H.lib()
for B in [int(x) for x in sys.argv[1:]] or [50, 25, 13, 7]:
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
    acts = torch.zeros(int(lib.dm_rssm_acts_floats(ctypes.byref(shp))), device='cuda')
    feat, post, prior = (torch.zeros(T * B, n, device='cuda') for n in (F_, Z, Z))
    idx = torch.zeros(T * B, S, dtype=torch.int32, device='cuda')

    def run():
        H.call('dm_rssm_sequence_fwd', ctypes.byref(shp), H.fptr(embed), H.fptr(action), H.ptr(reset), H.fptr(h0), H.fptr(z0), H.fptr(u), None,
               ctypes.byref(P), H.fptr(acts), H.fptr(feat), H.fptr(post), H.fptr(prior), H.ptr(idx), H.ptr(ws), ws.numel(), H.stream())

    keep = None
    for on in ((0,) if os.environ.get('NO_LDS') else (1, 0)):      # NO_LDS=1: launch schedules only (e.g. under rocprofv3)
        lib.dm_rssm_lds_enable(2 if on else 0)      # (level 2: any row count <= 64; the default level leaves B > 32 to the launch chain)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        out = (ctypes.c_ulonglong * 16)()
        lib.dm_rssm_lds_prof(out, 1)
        reps = 7
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        evs[0].record()
        for i in range(reps):
            run()
            evs[i + 1].record()
        torch.cuda.synchronize()
        lib.dm_rssm_lds_prof(out, 0)
        per = sorted(evs[i].elapsed_time(evs[i + 1]) * 1e3 for i in range(reps))
        th0 = time.perf_counter(); run(); th1 = time.perf_counter(); torch.cuda.synchronize()      # host time of one call's enqueue (idle queue)
        print(f'   host enqueue time of one call: {(th1 - th0) * 1e6:.0f} us')
        total_us = per[reps // 2]
        print(f'B={B} lds={on}: median {total_us:.0f} us per sequence call (T={T}; min {per[0]:.0f}, max {per[-1]:.0f}), status {lib.dm_rssm_lds_status()}')
        if on:
            # slot k holds the ticks between the previous tick of the step and tick k (csrc/rssm_lds.hip RL_TICK)
            order = [(0, 'A idx poll+gather+publish'), (1, 'B sweep x1'), (8, 'B LN+ELU'), (9, 'B gi MFMA'), (10, 'B reduce'),
                     (2, 'B gates+publish'), (3, 'C sweep h'), (14, 'C x2 MFMA+reduce+publish'), (4, 'C gh(t+1)'), (5, 'D sweep x2'),
                     (11, 'D LN+ELU'), (12, 'D logits MFMA'), (13, 'D reduce'), (6, 'D publish'), (7, 'L poll+sample+publish')]
            den = reps * (T - 1)
            print('   per step, us (workgroup 0): ' + ', '.join(f'{n} {out[k] * 0.01 / den:.2f}' for k, n in order)
                  + f'; sum {sum(out) * 0.01 / den:.2f}')
            keep = idx.clone()
        else:
            if keep is not None:
                print(f'   indices equal to the persistent kernel: {float((keep == idx).float().mean()):.6f}')
    # ---- the BPTT loop (dm_rssm_sequence_bwd: prior branch, loop, batched weight gradients): folded / unfolded launch schedule
    lib.dm_rssm_lds_enable(0 if os.environ.get('NO_LDS') else 2)
    run(); torch.cuda.synchronize()
    Gf, Gp, Gq = (torch.randn(T * B, n, generator=g).cuda() / (T * B) for n in (F_, Z, Z))
    grads = [None if p_ is None else torch.zeros_like(p_) for p_ in cell.ordered()]
    Gs = H.rssm_struct(grads, cls=H.dm_rssm_grads)
    dembed = torch.zeros(T * B, E, device='cuda')
    keep_g = None
    for on, fold in ((0, 1), (0, 0)):
        lib.dm_bptt_fold_enable(fold)      # launch schedule only: the LayerNorm backward stages folded into the products that consume them
        per = []
        for rep in range(6):
            dfeat, dpost, dprior = Gf.clone(), Gp.clone(), Gq.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            hb0 = time.perf_counter()
            H.call('dm_rssm_sequence_bwd', ctypes.byref(shp), H.fptr(embed), H.fptr(action), H.ptr(reset), ctypes.byref(P), H.fptr(acts),
                   H.fptr(feat), H.fptr(post), H.fptr(dfeat), H.fptr(dpost), H.fptr(dprior), ctypes.byref(Gs), H.fptr(dembed),
                   H.ptr(ws), ws.numel(), H.stream())
            hb1 = time.perf_counter()
            e1.record(); torch.cuda.synchronize()
            if rep:
                per.append(e0.elapsed_time(e1) * 1e3)
                host_us = (hb1 - hb0) * 1e6
        per.sort()
        flat = torch.cat([x.flatten() for x in grads if x is not None])
        print(f'B={B} bwd lds={on} fold={fold}: median {per[len(per) // 2]:.0f} us per dm_rssm_sequence_bwd call (T={T}; min {per[0]:.0f}; host enqueue {host_us:.0f} us), status {lib.dm_rssm_lds_status()}'
              + ('' if keep_g is None else f'; gradients vs the first schedule: rel-L2 {float((flat - keep_g).norm() / keep_g.norm()):.2e}'))
        keep_g = flat.clone() if keep_g is None else keep_g
    lib.dm_bptt_fold_enable(1)
lib.dm_rssm_lds_enable(1)
