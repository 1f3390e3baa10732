"""Wall-time ratio oracle / reference for one gradient step (build container only: imports /root/reference).

    python oracle/time_vs_reference.py [batch_columns=10] [steps=3] [out.json]  ->  profiles/r03_oracle_vs_reference.json

SURVEY 8(d): the CPU baseline that bench.py times on the GPU box is the oracle (kind "port"; the reference's Python never
travels).  Its representativeness is established HERE by running the reference's own loop (train.py:165-198: forward under
its nn.Modules and torch.distributions, 4 x backward, clip, 4 x AdamW) and the oracle's on the same synthetic
Atari-literal batch (same model size, T=50, H=15; `batch_columns` of the 50 columns), one warm-up step excluded.
bench.py reads the JSON and reports the ratio with its provenance instead of a hard-coded constant."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dreamer_oracle as O          # noqa: E402
from oracle import gen_golden as G              # noqa: E402


def main():
    cols = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    threads = min(8, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1))      # (taskset-aware)
    torch.set_num_threads(threads)
    rconf = G.reference_conf(['defaults', 'atari'], dict(batch_size=cols, batch_length=50, imag_horizon=15, deter_dim=600,
                                                         action_dim=18))
    oconf = O.make_conf(**{k: getattr(rconf, k) for k in O.DEFAULTS})
    obs = O.preprocess(O.synthetic_batch(oconf), oconf)
    noise = O.make_noise(oconf)
    params = O.make_params(oconf, seed=0)

    # both models are built first and their steps INTERLEAVED (oracle, reference, oracle, ...), so that slow drifts of the
    # container (other tenants, clocks) hit both alike; per step the forward / backward / clip+optimizer parts are timed too
    model = O.OracleDreamer(oconf, params)
    model.init_optimizers()
    state = model.init_state(cols)
    sys.path.insert(0, G.REF)
    from pydreamer.models import Dreamer
    import torch.distributions as D
    D.Distribution.set_default_validate_args(False)
    ref = Dreamer(rconf)
    ref.load_state_dict(params, strict=True)
    opts = ref.init_optimizers(rconf.adam_lr, rconf.adam_lr_actor, rconf.adam_lr_critic, rconf.adam_eps)
    rstate = ref.init_state(cols)
    t_or, t_ref, ph_or, ph_ref = [], [], [], []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        losses, state, *_ = model.training_step(obs, state, noise)
        t1 = time.perf_counter()
        model.backward_clip_step(losses)
        t2 = time.perf_counter()
        t_or.append(t2 - t0); ph_or.append((t1 - t0, t2 - t1))

        t0 = time.perf_counter()
        losses, rstate, metrics, tensors, _ = ref.training_step(obs, rstate)
        t1 = time.perf_counter()
        for opt in opts:
            opt.zero_grad()
        for loss in losses:
            loss.backward()
        t2 = time.perf_counter()
        ref.grad_clip(rconf.grad_clip, rconf.grad_clip_ac)
        for opt in opts:
            opt.step()
        t3 = time.perf_counter()
        t_ref.append(t3 - t0); ph_ref.append((t1 - t0, t3 - t1))
        print(f'step {i}: oracle {t_or[-1]:.2f} s (fwd {ph_or[-1][0]:.2f}, bwd+opt {ph_or[-1][1]:.2f}) | reference {t_ref[-1]:.2f} s '
              f'(fwd {ph_ref[-1][0]:.2f}, bwd+opt {ph_ref[-1][1]:.2f})', flush=True)
    o, r = sum(t_or[1:]) / steps, sum(t_ref[1:]) / steps
    med = lambda xs: sorted(xs)[len(xs) // 2]
    out = dict(oracle_s_per_step=o, reference_s_per_step=r, oracle_over_reference_time=o / r,
               oracle_over_reference_median=med(t_or[1:]) / med(t_ref[1:]),
               oracle_fwd_s=sum(p[0] for p in ph_or[1:]) / steps, oracle_bwd_opt_s=sum(p[1] for p in ph_or[1:]) / steps,
               reference_fwd_s=sum(p[0] for p in ph_ref[1:]) / steps, reference_bwd_opt_s=sum(p[1] for p in ph_ref[1:]) / steps,
               per_step_oracle_s=t_or[1:], per_step_reference_s=t_ref[1:], interleaved=True,
               batch_columns=cols, steps=steps, threads=threads, host='build container (no GPU)', warmup_steps_excluded=1,
               note='reference = /root/reference pydreamer.models.Dreamer driven by the train.py:165-198 section; same batch, '
                    'same weights, same torch build; within +-10 % means the oracle is a representative CPU baseline (SURVEY 8(d))')
    path = os.path.join(ROOT, 'profiles', sys.argv[3] if len(sys.argv) > 3 else 'r04_oracle_vs_reference.json')
    with open(path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
