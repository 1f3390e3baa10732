"""Generate golden vectors from the REAL reference (runs only in the build container, where /root/reference exists).

    python oracle/gen_golden.py            # writes tests/golden/*.npz

What it does:
  * builds `argparse.Namespace conf` from /root/reference/config/defaults.yaml exactly as launch.py:16-41 does
    (sections merged in order, then overrides);
  * imports `pydreamer.models.Dreamer` from /root/reference (never copied, never shipped);
  * loads the closed-form weights of `oracle.dreamer_oracle.make_params` into it (strict key/shape match);
  * patches `torch.multinomial` — the reference's only sampler call site (torch/distributions/categorical.py:147) —
    with the inverse-CDF rule over uniforms supplied in call order;
  * runs the trainer section train.py:165-198 (forward, 4 x backward, clip, 4 x AdamW) for 2 consecutive steps with
    the recurrent state carried (keep_state) so the critic_target refresh at call 0 and the TBTT carry are covered;
  * stores inputs (uint8 images, actions, rewards, resets, uniforms) and outputs (losses, metrics, tensors, sampled
    indices, per-parameter grad norms, a few full gradients, post-AdamW parameter checksums).

The fixtures are data only.  `tests/test_oracle_golden.py` replays them through oracle/dreamer_oracle.py.
"""
import os
import sys
from argparse import Namespace

import numpy as np
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'

from oracle import dreamer_oracle as O   # noqa: E402


def reference_conf(sections, overrides):
    with open(os.path.join(REF, 'config', 'defaults.yaml')) as f:
        allc = yaml.safe_load(f)
    conf = {}
    for s in sections:
        conf.update(allc[s])
    conf.update(overrides)
    return Namespace(**conf)


class MultinomialPatch:
    """Replaces torch.multinomial(probs_2d, 1, True) by the shared inverse-CDF rule on queued uniforms, and
    torch.normal(mean, std) (continuous actors: Normal.sample) by mean + std * eps on queued standard-normal noise, and
    torch.distributions' _standard_normal (Normal.rsample: Gaussian latents, stoch_discrete = 0) by queued draws."""

    def __init__(self):
        import torch.distributions.normal as tdn
        self.tdn = tdn
        self.orig_std = tdn._standard_normal
        self.std_queue = []
        self.queue = []
        self.eps_queue = []
        self.calls = []
        self.idx = []
        self.orig = torch.multinomial
        self.orig_normal = torch.normal

    def __enter__(self):
        torch.multinomial = self
        torch.normal = self.normal
        self.tdn._standard_normal = self.standard_normal
        return self

    def __exit__(self, *a):
        torch.multinomial = self.orig
        torch.normal = self.orig_normal
        self.tdn._standard_normal = self.orig_std

    def standard_normal(self, shape, dtype, device):
        eps = self.std_queue.pop(0)
        assert tuple(eps.shape) == tuple(shape), (eps.shape, shape)
        return eps.to(dtype)

    def normal(self, mean, std, *a, **kw):
        eps = self.eps_queue.pop(0)
        assert eps.shape == mean.shape, (eps.shape, mean.shape)
        return mean + std * eps

    def __call__(self, probs, num_samples, replacement=False, **kw):
        assert num_samples == 1 and probs.dim() == 2
        u = self.queue.pop(0).reshape(-1)
        assert u.numel() == probs.shape[0], (u.shape, probs.shape)
        idx = O.sample_inverse_cdf(probs, u)
        self.calls.append(tuple(probs.shape))
        self.idx.append(idx.clone())
        return idx.unsqueeze(-1)


def run(name, sections, overrides, steps=2, full_grads=(), save_image_rec_frames=1, slim=False):
    """slim: full-size configs - inputs are NOT stored (tests regenerate them from the same seeds through
    oracle.synthetic_batch / make_noise), tensors are reduced to checksums."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, REF)
    from pydreamer.models import Dreamer          # the reference, imported in place
    import torch.distributions as D
    D.Distribution.set_default_validate_args(False)   # train.py:30

    rconf = reference_conf(sections, overrides)
    oconf = O.make_conf(**{k: getattr(rconf, k) for k in O.DEFAULTS})
    model = Dreamer(rconf)
    params = O.make_params(oconf, seed=0)
    sd = model.state_dict()
    assert list(sd.keys()) == list(params.keys()), 'state_dict key order differs from oracle.param_shapes'
    for k in sd:
        assert tuple(sd[k].shape) == tuple(params[k].shape), (k, sd[k].shape, params[k].shape)
    model.load_state_dict(params, strict=True)
    optimizers = model.init_optimizers(rconf.adam_lr, rconf.adam_lr_actor, rconf.adam_lr_critic, rconf.adam_eps)

    T, B, S, H = rconf.batch_length, rconf.batch_size, rconf.stoch_dim, rconf.imag_horizon
    out = {'conf_json': np.array(repr(sorted(vars(oconf).items())))}
    state = model.init_state(B * rconf.iwae_samples)
    for step in range(steps):
        raw = O.synthetic_batch(oconf, seed=1234 + step, first=(step == 0))
        obs = O.preprocess(raw, oconf)
        noise = O.make_noise(oconf, seed=777 + step)
        with MultinomialPatch() as mp:
            onehot = rconf.actor_dist == 'onehot'
            gaussian = not rconf.stoch_discrete          # Gaussian latents draw through Normal.rsample, not multinomial
            lat_queue = mp.std_queue if gaussian else mp.queue
            lat_queue += [noise['u_post'][t] for t in range(T)]
            for i in range(H):
                if onehot:
                    mp.queue.append(noise['u_act'][i])
                else:
                    mp.eps_queue.append(noise['eps_act'][i])
                lat_queue.append(noise['u_prior'][i])
            losses, new_state, metrics, tensors, _ = model.training_step(obs, state)
            assert not mp.queue and not mp.eps_queue and not mp.std_queue, \
                f'{len(mp.queue)} uniforms / {len(mp.eps_queue)} + {len(mp.std_queue)} normals unused'
            M = T * B * rconf.iwae_samples
            if gaussian:
                post_idx = torch.zeros(T, B * rconf.iwae_samples, S, dtype=torch.long)
                lat_idx = torch.zeros(H, M, S, dtype=torch.long)
                act_idx = torch.stack(mp.idx).reshape(H, M) if onehot else torch.zeros(H, M, dtype=torch.long)
            else:
                post_idx = torch.stack(mp.idx[:T]).reshape(T, B * rconf.iwae_samples, S)
                if onehot:
                    act_idx = torch.stack(mp.idx[T::2]).reshape(H, M)
                    lat_idx = torch.stack(mp.idx[T + 1::2]).reshape(H, M, S)
                else:
                    act_idx = torch.zeros(H, M, dtype=torch.long)
                    lat_idx = torch.stack(mp.idx[T:]).reshape(H, M, S)
        for opt in optimizers:
            opt.zero_grad()
        for loss in losses:
            loss.backward()
        grad_metrics = model.grad_clip(rconf.grad_clip, rconf.grad_clip_ac)
        named = dict(model.named_parameters())
        grads = {k: v.grad.detach().clone() for k, v in named.items() if v.grad is not None}
        for opt in optimizers:
            opt.step()

        pre = f's{step}_'
        if not slim:
            for k, v in raw.items():
                out[pre + 'in_' + k] = v
            for k, v in noise.items():
                out[pre + 'in_' + k] = v.numpy()
            out[pre + 'in_state_h'] = state[0].numpy()
            out[pre + 'in_state_z'] = state[1].numpy()
        else:   # fingerprints of the regenerated inputs, so a drifting generator is reported as such and not as a parity bug
            out[pre + 'in_image_sum'] = np.array(int(raw['image_u8'].astype(np.int64).sum()))
            out[pre + 'in_u_post_sum'] = np.array(float(noise['u_post'].double().sum()))
        out[pre + 'losses'] = np.array([float(l) for l in losses], dtype=np.float64)
        for k, v in {**metrics, **grad_metrics}.items():
            out[pre + 'metric_' + k] = np.array(float(v), dtype=np.float64)
        for k, v in tensors.items():
            if slim:
                out[pre + 'tensor_' + k + '_sum'] = np.array(float(v.detach().double().sum()))
            elif k == 'image_rec':
                out[pre + 'tensor_image_rec_sum'] = np.array(float(v.double().sum()))
                out[pre + 'tensor_image_rec_frames'] = v[:save_image_rec_frames, :1].numpy()
            else:
                out[pre + 'tensor_' + k] = v.detach().numpy()
        out[pre + 'out_state_h'] = new_state[0].numpy()
        out[pre + 'out_state_z'] = new_state[1].numpy()
        out[pre + 'idx_post'] = post_idx.numpy().astype(np.uint8)
        out[pre + 'idx_act'] = act_idx.numpy().astype(np.uint8)
        # the FULL (H, M, S) latent index tensor of the imagination, slim fixtures included (1.2 MB of u8 at Atari-literal,
        # rssm.py:177-179): a test can then name the first step / group at which a trajectory left the reference's
        out[pre + 'idx_lat'] = lat_idx.numpy().astype(np.uint8)
        if slim:
            out[pre + 'idx_lat_rowsum'] = lat_idx.sum(-1).numpy().astype(np.uint16)     # (H, M): sum of the 32 indices
        out[pre + 'grad_norms'] = np.array([float(g.double().norm()) for g in grads.values()])
        out[pre + 'grad_names'] = np.array(list(grads.keys()))
        out[pre + 'grad_proj'] = np.array([O.grad_probe(g, i) for i, g in enumerate(grads.values())])     # (P, 2): directions
        for k in full_grads:
            out[pre + 'grad_' + k] = grads[k].numpy()
        post = dict(model.state_dict())
        out[pre + 'param_sums'] = np.array([float(v.double().sum()) for v in post.values()])
        out[pre + 'param_abs_sums'] = np.array([float(v.double().abs().sum()) for v in post.values()])
        state = new_state
        print(f'[{name}] step {step}: losses', out[pre + 'losses'], 'grad_norm', float(grad_metrics['grad_norm']))
    path = os.path.join(ROOT, 'tests', 'golden', f'{name}.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, f'{os.path.getsize(path) / 1024:.0f} KiB')


def run_amp(name, sections, overrides):
    """BASELINE configs[2] family: the reference's mixed-precision step (train.py:166 `autocast(enabled=conf.amp)`), run on
    CPU under torch.autocast('cpu', bfloat16).  Forward losses / metrics only (a loose pin: the build's bf16 mode keeps fp32
    storage, autocast also rounds layer outputs to bf16)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, REF)
    from pydreamer.models import Dreamer
    import torch.distributions as D
    D.Distribution.set_default_validate_args(False)
    rconf = reference_conf(sections, overrides)
    oconf = O.make_conf(**{k: getattr(rconf, k) for k in O.DEFAULTS})
    model = Dreamer(rconf)
    model.load_state_dict(O.make_params(oconf, seed=0), strict=True)
    T, B, S, H = rconf.batch_length, rconf.batch_size, rconf.stoch_dim, rconf.imag_horizon
    raw = O.synthetic_batch(oconf, seed=1234, first=True)
    obs = O.preprocess(raw, oconf)
    noise = O.make_noise(oconf, seed=777)
    out = {'conf_json': np.array(repr(sorted(vars(oconf).items())))}
    for tag, amp in (('fp32', False), ('bf16', True)):
        with MultinomialPatch() as mp:
            mp.queue = [noise['u_post'][t] for t in range(T)]
            for i in range(H):
                if rconf.actor_dist == 'onehot':
                    mp.queue.append(noise['u_act'][i])
                else:      # continuous actors draw through torch.normal (Normal.sample), as in run()
                    mp.eps_queue.append(noise['eps_act'][i])
                mp.queue.append(noise['u_prior'][i])
            with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16, enabled=amp):
                losses, _, metrics, _, _ = model.training_step(obs, model.init_state(B))
            post_idx = torch.stack(mp.idx[:T]).reshape(T, B, S)
        out[tag + '_losses'] = np.array([float(l) for l in losses], dtype=np.float64)
        for k, v in metrics.items():
            out[tag + '_metric_' + k] = np.array(float(v), dtype=np.float64)
        out[tag + '_idx_post'] = post_idx.numpy().astype(np.uint8)
        print(f'[{name}] {tag}: losses', out[tag + '_losses'])
    path = os.path.join(ROOT, 'tests', 'golden', f'{name}.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, f'{os.path.getsize(path) / 1024:.0f} KiB')


def run_amp_grads(name, sections, overrides):
    """Gradient side of run_amp: the same full-size step under torch.autocast('cpu', bfloat16) WITH the four backward passes
    (train.py:166-198 without the GradScaler, which bf16 does not need); stores the losses (they must equal run_amp's bf16
    losses: same inputs, same uniforms) and the per-parameter gradient norms BEFORE clipping.  Separate small fixture
    (<name>_grads.npz) so the forward fixture stays untouched.  Several minutes of CPU per backward at full width."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, REF)
    from pydreamer.models import Dreamer
    import torch.distributions as D
    D.Distribution.set_default_validate_args(False)
    rconf = reference_conf(sections, overrides)
    oconf = O.make_conf(**{k: getattr(rconf, k) for k in O.DEFAULTS})
    model = Dreamer(rconf)
    model.load_state_dict(O.make_params(oconf, seed=0), strict=True)
    T, B, S, H = rconf.batch_length, rconf.batch_size, rconf.stoch_dim, rconf.imag_horizon
    obs = O.preprocess(O.synthetic_batch(oconf, seed=1234, first=True), oconf)
    noise = O.make_noise(oconf, seed=777)
    with MultinomialPatch() as mp:
        mp.queue = [noise['u_post'][t] for t in range(T)]
        for i in range(H):
            if rconf.actor_dist == 'onehot':
                mp.queue.append(noise['u_act'][i])
            else:
                mp.eps_queue.append(noise['eps_act'][i])
            mp.queue.append(noise['u_prior'][i])
        with torch.autocast('cpu', dtype=torch.bfloat16, enabled=True):
            losses, _, metrics, _, _ = model.training_step(obs, model.init_state(B))
        post_idx = torch.stack(mp.idx[:T]).reshape(T, B, S)
    for loss in losses:
        loss.backward()
    named = dict(model.named_parameters())
    grads = {k: v.grad.detach() for k, v in named.items() if v.grad is not None}
    out = {'conf_json': np.array(repr(sorted(vars(oconf).items()))),
           'bf16_losses': np.array([float(l) for l in losses], dtype=np.float64),
           'bf16_idx_post': post_idx.numpy().astype(np.uint8),
           'bf16_grad_names': np.array(list(grads.keys())),
           'bf16_grad_norms': np.array([float(g.double().norm()) for g in grads.values()])}
    path = os.path.join(ROOT, 'tests', 'golden', f'{name}_grads.npz')
    np.savez_compressed(path, **out)
    print(f'[{name}_grads] losses', out['bf16_losses'], 'wrote', path, f'{os.path.getsize(path) / 1024:.0f} KiB')


def run_eval(name, sections, overrides, do_open_loop=False, iwae_samples=None):
    """Logging variants of training_step (train.py:353-359,380-385 call it with do_image_pred / do_dream_tensors):
    one forward with both flags; inputs, extra uniforms and every extra output are stored.
    iwae_samples = I: the call shape of evaluate() itself, which passes iwae_samples=eval_samples TOGETHER with the flags to
    a model whose conf.iwae_samples stays 1 (train.py:353-359,380-385)."""
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, REF)
    from pydreamer.models import Dreamer
    import torch.distributions as D
    D.Distribution.set_default_validate_args(False)
    rconf = reference_conf(sections, overrides)
    oconf = O.make_conf(**{k: getattr(rconf, k) for k in O.DEFAULTS})
    model = Dreamer(rconf)
    model.load_state_dict(O.make_params(oconf, seed=0), strict=True)
    T, B, S, H = rconf.batch_length, rconf.batch_size, rconf.stoch_dim, rconf.imag_horizon
    raw = O.synthetic_batch(oconf, seed=4321, first=True)
    obs = O.preprocess(raw, oconf)
    Ie = int(iwae_samples or 1)
    noise = O.make_noise(oconf, seed=999, eval_iwae=iwae_samples)
    onehot = rconf.actor_dist == 'onehot'
    with MultinomialPatch() as mp:
        mp.queue = [noise['u_post'][t] for t in range(T)] + [noise['u_pred'].reshape(-1, S)]
        for i in range(H):
            mp.queue += ([noise['u_act'][i]] if onehot else []) + [noise['u_prior'][i]]
            if not onehot:
                mp.eps_queue.append(noise['eps_act'][i])
        for i in range(T - 1):
            mp.queue += ([noise['u_act_log'][i]] if onehot else []) + [noise['u_prior_log'][i]]
            if not onehot:
                mp.eps_queue.append(noise['eps_act_log'][i])
        with torch.no_grad():
            losses, new_state, metrics, tensors, dream_tensors = model.training_step(
                obs, model.init_state(B * Ie), iwae_samples=iwae_samples, do_image_pred=True, do_dream_tensors=True,
                do_open_loop=do_open_loop)
        assert not mp.queue and not mp.eps_queue
        pred_idx = mp.idx[T].reshape(T, B * Ie, S)
        tail = mp.idx[T + 1 + (2 if onehot else 1) * H:]
        log_act = torch.stack(tail[0::2]) if onehot else torch.zeros(T - 1, B, dtype=torch.long)
        log_lat = torch.stack(tail[1::2] if onehot else tail).reshape(T - 1, B, S)
    out = {'conf_json': np.array(repr(sorted(vars(oconf).items())))}
    for k, v in raw.items():
        out['in_' + k] = v
    for k, v in noise.items():
        out['in_' + k] = v.numpy()
    out['losses'] = np.array([float(l) for l in losses], dtype=np.float64)
    for k, v in metrics.items():
        out['metric_' + k] = np.array(float(v), dtype=np.float64)
    for k, v in tensors.items():
        if k in ('image_rec', 'image_pred'):
            out['tensor_' + k + '_sum'] = np.array(float(v.double().sum()))
            out['tensor_' + k + '_frame'] = v[:1, :1].numpy()
        else:
            out['tensor_' + k] = v.detach().numpy()
    for k, v in dream_tensors.items():
        if k == 'image_pred':
            out['dream_image_pred_sum'] = np.array(float(v.double().sum()))
            out['dream_image_pred_frame'] = v[-1:, :1].numpy()
        else:
            out['dream_' + k] = v.detach().numpy()
    out['idx_post'] = torch.stack(mp.idx[:T]).reshape(T, B * Ie, S).numpy().astype(np.uint8)
    if iwae_samples:
        out['iwae_samples'] = np.array(Ie)
    out['out_state_h'] = new_state[0].numpy()
    out['idx_pred'] = pred_idx.numpy().astype(np.uint8)
    out['idx_log_act'] = log_act.numpy().astype(np.uint8)
    out['idx_log_lat'] = log_lat.numpy().astype(np.uint8)
    path = os.path.join(ROOT, 'tests', 'golden', f'{name}.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, f'{os.path.getsize(path) / 1024:.0f} KiB', {k: float(v) for k, v in metrics.items() if 'logprob' in k})


def run_inference(name, sections, overrides):
    """Dreamer.inference (dreamer.py:92-111) as the acting process calls it (generator.py:317-331): one step, (1,B,...) obs,
    the posterior draw pinned by the same multinomial patch.  Stores inputs, uniforms, action probabilities, the new state
    and policy_value - what an actor running the reference would see when it loads a checkpoint written by the build."""
    torch.manual_seed(0)
    sys.path.insert(0, REF)
    from pydreamer.models import Dreamer
    import torch.distributions as D
    D.Distribution.set_default_validate_args(False)
    rconf = reference_conf(sections, overrides)
    oconf = O.make_conf(**{k: getattr(rconf, k) for k in O.DEFAULTS})
    model = Dreamer(rconf)
    model.load_state_dict(O.make_params(oconf, seed=0), strict=True)
    B, S, C, A = 3, rconf.stoch_dim, rconf.stoch_discrete, rconf.action_dim
    g = torch.Generator().manual_seed(31)
    image_u8 = torch.randint(0, 256, (1, B, 64, 64, 3), generator=g, dtype=torch.uint8)
    image = (image_u8.float() / 255.0 - 0.5).permute(0, 1, 4, 2, 3).contiguous()
    action = torch.nn.functional.one_hot(torch.randint(0, A, (1, B), generator=g), A).float()
    reset = torch.tensor([[True, False, False]])
    h = torch.tanh(torch.randn(B, rconf.deter_dim, generator=g))
    z = torch.nn.functional.one_hot(torch.randint(0, C, (B, S), generator=g), C).float().reshape(B, S * C)
    u = torch.rand(1, B, S, generator=g)
    obs = dict(image=image, action=action, reset=reset, reward=torch.zeros(1, B), terminal=torch.zeros(1, B))
    with MultinomialPatch() as mp, torch.no_grad():
        mp.queue = [u[0]]
        dist, (h1, z1), metrics = model.inference(obs, (h, z))
        assert not mp.queue
        probs = dist.probs if hasattr(dist, 'probs') else None
    out = dict(conf_json=np.array(repr(sorted(vars(oconf).items()))), in_image_u8=image_u8.numpy(), in_action=action.numpy(),
               in_reset=reset.numpy(), in_h=h.numpy(), in_z=z.numpy(), in_u=u.numpy(), action_probs=probs.numpy(),
               out_h=h1.numpy(), out_z=z1.numpy(), policy_value=np.array(float(metrics['policy_value'])),
               state_dict_keys=np.array(list(model.state_dict().keys())))
    path = os.path.join(ROOT, 'tests', 'golden', f'{name}.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, f'{os.path.getsize(path) / 1024:.0f} KiB', 'policy_value', float(metrics['policy_value']))


SMALL_GRADS = ('wm.core.cell.a_mlp.weight', 'wm.core.cell.gru.layers.0.bias_hh', 'wm.core.cell.post_norm.weight',
               'wm.core.cell.prior_mlp.bias', 'wm.encoder.encoder_image.model.0.weight',
               'wm.decoder.image.model.8.weight', 'ac.actor.model.12.weight', 'ac.critic.model.1.weight')

if __name__ == '__main__':
    which = sys.argv[1:] or ['tiny', 'debug', 'dmc']
    if 'tiny' in which:
        t = O.tiny_conf()
        run('tiny', ['defaults', 'atari'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon), full_grads=SMALL_GRADS)
    if 'dmc' in which:
        # BASELINE.json configs[4] family: continuous actions (defaults+dmc, tanh_normal) with actor_grad=reinforce
        # (the dmc section's actor_grad=dynamics asserts in the reference), tiny dims
        t = O.tiny_conf()
        run('tiny_dmc', ['defaults', 'dmc'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=4, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon, actor_grad='reinforce'), steps=1,
            full_grads=('ac.actor.model.12.weight', 'ac.actor.model.12.bias'))
    if 'eval' in which:
        t = O.tiny_conf()
        run_eval('tiny_eval', ['defaults', 'atari'],
                 dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                      cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                      imag_horizon=t.imag_horizon))
        run_eval('tiny_open_loop', ['defaults', 'atari'],
                 dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                      cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                      imag_horizon=t.imag_horizon), do_open_loop=True)
    if 'eval_iwae' in which:
        # evaluate()'s own call shape (train.py:353-359,380-385): iwae_samples=eval_samples WITH do_image_pred /
        # do_dream_tensors (closed loop) and WITH do_open_loop (open loop): the I > 1 reductions of decoders.py:85-106,170-171
        t = O.tiny_conf()
        base = dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                    cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                    imag_horizon=t.imag_horizon)
        run_eval('tiny_eval_iwae', ['defaults', 'atari'], base, iwae_samples=3)
        run_eval('tiny_open_loop_iwae', ['defaults', 'atari'], base, do_open_loop=True, iwae_samples=3)
    if 'amp' in which:
        t = O.tiny_conf()
        run_amp('tiny_amp', ['defaults', 'atari'],
                dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                     cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                     imag_horizon=t.imag_horizon))
    if 'atari' in which:
        # BASELINE.json configs[1]: Atari-literal at full size (B=50,T=50,H=15,deter 600); ~1 min per step on 8 vCPU
        run('atari_literal', ['defaults', 'atari'],
            dict(batch_size=50, batch_length=50, imag_horizon=15, deter_dim=600, action_dim=18), steps=1, slim=True)
    for gt in ('gru_layernorm', 'gru_layernorm_dv2'):
        if gt in which or 'grucells' in which:
            # the LayerNorm GRU cells (rnn.py:95-138), tiny dims, 2 training steps incl. gradients
            t = O.tiny_conf()
            run('tiny_' + gt, ['defaults', 'atari'],
                dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                     cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                     imag_horizon=t.imag_horizon, gru_type=gt), steps=2,
                full_grads=('wm.core.cell.post_norm.weight', 'wm.core.cell.a_mlp.weight', 'ac.actor.model.12.weight'))
    if 'gru_layers' in which:
        # GRUCellStack with several layers (rnn.py:40-67): 3 plain GRU cells of width deter_dim/3 = 32
        t = O.tiny_conf()
        run('tiny_gru_layers3', ['defaults', 'atari'],
            dict(deter_dim=96, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon, gru_layers=3), steps=2,
            full_grads=('wm.core.cell.gru.layers.0.weight_ih', 'wm.core.cell.gru.layers.1.weight_ih',
                        'wm.core.cell.gru.layers.2.weight_hh', 'wm.core.cell.gru.layers.1.bias_hh',
                        'wm.core.cell.z_mlp.weight', 'ac.actor.model.12.weight'))
    if 'no_layernorm' in which:
        # SURVEY 8(a) variant: layer_norm=False (common.py:68-74 NoNorm in every MLP head and in the RSSM cell's three norms)
        t = O.tiny_conf()
        run('tiny_no_layernorm', ['defaults', 'atari'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon, layer_norm=False), steps=2,
            full_grads=('wm.core.cell.z_mlp.weight', 'wm.core.cell.post_mlp_h.weight', 'wm.core.cell.prior_mlp.bias',
                        'wm.decoder.reward.model.model.12.weight', 'ac.actor.model.12.weight', 'ac.critic.model.12.weight'))
    if 'gaussian' in which:
        # SURVEY 8(a) variant: stoch_discrete = 0 - Gaussian latents (rssm.py:103-117,195-203, functions.py:46-56): z is
        # stoch_dim wide, prior / posterior heads emit (mean, std) and Normal.rsample / the Normal KL replace the categorical
        t = O.tiny_conf()
        run('tiny_gaussian_latents', ['defaults', 'atari'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=0,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon), steps=2,
            full_grads=('wm.core.cell.z_mlp.weight', 'wm.core.cell.post_mlp.weight', 'wm.core.cell.prior_mlp.bias',
                        'ac.actor.model.12.weight'))
    if 'corners' in which:
        # corners the HIP path does not build yet; the oracle is pinned for them ahead of the product:
        # Gaussian latents with iwae_samples = 2 (sampled Normal log-density KL, dreamer.py:340-343) and a 2-layer stack of
        # NormGRUCell (rnn.py:40-67 with rnn.py:95-114)
        t = O.tiny_conf()
        base = dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, cnn_depth=t.cnn_depth,
                    action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size, imag_horizon=t.imag_horizon)
        run('tiny_gaussian_iwae', ['defaults', 'atari'], dict(base, stoch_discrete=0, iwae_samples=2), steps=1,
            full_grads=('wm.core.cell.post_mlp.weight',))
        run('tiny_gru_layernorm_layers2', ['defaults', 'atari'],
            dict(base, stoch_discrete=t.stoch_discrete, gru_type='gru_layernorm', gru_layers=2), steps=1,
            full_grads=('wm.core.cell.gru.layers.1.ln_update.weight',))
    if 'variants2' in which:
        # SURVEY 8(a) variants inside the same functions: the normal_tanh actor (functions.py:59-66, a2c.py:43-55; continuous
        # actions with actor_grad=reinforce) and the plain-KL branch kl_balance = 0.5 (dreamer.py:241,334-335)
        t = O.tiny_conf()
        base = dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                    cnn_depth=t.cnn_depth, batch_length=t.batch_length, batch_size=t.batch_size, imag_horizon=t.imag_horizon)
        run('tiny_normal_tanh', ['defaults', 'dmc'], dict(base, action_dim=4, actor_grad='reinforce', actor_dist='normal_tanh'),
            steps=2, full_grads=('ac.actor.model.12.weight', 'ac.actor.model.12.bias', 'ac.critic.model.12.weight'))
        run('tiny_kl_plain', ['defaults', 'atari'], dict(base, action_dim=t.action_dim, kl_balance=0.5), steps=2,
            full_grads=('wm.core.cell.post_mlp.weight', 'wm.core.cell.prior_mlp.weight', 'wm.core.cell.prior_mlp.bias'))
    if 'scalars' in which:
        # every scalar hyper-parameter of the path away from its default at once (defaults.yaml:38-53,94-97): loss weights, KL
        # weight and balance, discount, GAE lambda, entropy weight, learning rates, Adam eps, clip thresholds that BIND
        # (grad_clip 100 / grad_clip_ac 0.2 are below the tiny model's gradient norms), target refresh every step; 3 steps
        t = O.tiny_conf()
        run('tiny_scalars', ['defaults', 'atari'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon, kl_weight=0.3, kl_balance=0.65, image_weight=0.7, reward_weight=2.0,
                 terminal_weight=0.5, gamma=0.95, lambda_gae=0.8, entropy=0.01, target_interval=1, adam_lr=1.0e-3,
                 adam_lr_actor=3.0e-4, adam_lr_critic=2.0e-4, adam_eps=1.0e-6, grad_clip=100, grad_clip_ac=0.2), steps=3,
            full_grads=SMALL_GRADS)
    if 'combo' in which:
        # the structural variants of SURVEY 8(a) TOGETHER: Gaussian latents, a 2-layer stack of late-reset LayerNorm GRU cells,
        # NoNorm MLPs, the auxiliary critic and a tanh_normal actor on continuous actions - the corners were pinned one by one,
        # this pins their interaction
        t = O.tiny_conf()
        run('tiny_combo', ['defaults', 'dmc'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=0, cnn_depth=t.cnn_depth,
                 action_dim=4, batch_length=t.batch_length, batch_size=t.batch_size, imag_horizon=t.imag_horizon,
                 actor_grad='reinforce', actor_dist='tanh_normal', gru_type='gru_layernorm_dv2', gru_layers=2, layer_norm=False,
                 aux_critic=True), steps=2,
            full_grads=('wm.core.cell.gru.layers.1.weight_hh.weight', 'wm.core.cell.post_mlp.weight', 'ac.actor.model.12.weight',
                        'wm.ac_aux.critic.model.12.weight'))
    if 'aux' in which:
        # aux_critic (dreamer.py:267-279,347-358): a critic on the REAL trajectory inside the world model
        t = O.tiny_conf()
        run('tiny_aux_critic', ['defaults', 'atari'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon, aux_critic=True), steps=2,
            full_grads=('wm.ac_aux.critic.model.12.weight', 'wm.core.cell.post_norm.weight'))
    if 'inference' in which:
        t = O.tiny_conf()
        run_inference('tiny_inference', ['defaults', 'atari'],
                      dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                           cnn_depth=t.cnn_depth, action_dim=t.action_dim))
    if 'iwae' in which:
        # iwae_samples > 1 (rssm.py:35-41 batch expansion, dreamer.py:340-343 sampled KL,
        # functions.py:97-102 logavgexp), as a TRAINING step (gradients included), tiny dims, I = 3
        t = O.tiny_conf()
        run('tiny_iwae', ['defaults', 'atari'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon, iwae_samples=3), steps=2, full_grads=SMALL_GRADS)
    if 'probe_gradients' in which:
        # probe_gradients=True with probe_model=none (dreamer.py:60-87,183-186): THREE optimizers (wm, actor, critic), the first loss
        # is loss_model + loss_probe, grad_clip() returns three norms; the probe head's dummy parameter gets a gradient but no step
        t = O.tiny_conf()
        run('tiny_probe_gradients', ['defaults', 'atari'],
            dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim, stoch_discrete=t.stoch_discrete,
                 cnn_depth=t.cnn_depth, action_dim=t.action_dim, batch_length=t.batch_length, batch_size=t.batch_size,
                 imag_horizon=t.imag_horizon, probe_gradients=True), steps=2, full_grads=SMALL_GRADS)
    if 'dmc_native' in which:
        # BASELINE.json configs[4] at its native width: defaults+dmc (deter_dim 2048, tanh_normal actor) with
        # actor_grad=reinforce, action_dim 6, B=50, T=50, H=15; slim fixture (several minutes per step on 8 vCPU)
        run('dmc_native', ['defaults', 'dmc'],
            dict(batch_size=50, batch_length=50, imag_horizon=15, action_dim=6, actor_grad='reinforce'), steps=1, slim=True)
    if 'atari_native' in which:
        # pydreamer's OWN Atari configuration (defaults+atari as shipped: B=32, T=48, deter_dim 1024, H=15; what the reference's
        # README measured and what `bench.py --workload atari-native` runs), action_dim 18; slim fixture
        run('atari_native', ['defaults', 'atari'], dict(action_dim=18), steps=1, slim=True)
    if 'atari_amp' in which:
        # BASELINE.json configs[2]: Atari-literal forward under torch.autocast('cpu', bfloat16) and in fp32 (slim: scalars +
        # posterior indices)
        run_amp('atari_literal_amp', ['defaults', 'atari'],
                dict(batch_size=50, batch_length=50, imag_horizon=15, deter_dim=600, action_dim=18))
    if 'atari_amp_grads' in which:
        run_amp_grads('atari_literal_amp', ['defaults', 'atari'],
                      dict(batch_size=50, batch_length=50, imag_horizon=15, deter_dim=600, action_dim=18))
    if 'dmc_amp_grads' in which:
        run_amp_grads('dmc_native_amp', ['defaults', 'dmc'],
                      dict(batch_size=50, batch_length=50, imag_horizon=15, action_dim=6, actor_grad='reinforce'))
    if 'dmc_amp' in which:
        # BASELINE.json configs[4] as named: DMC continuous actions (defaults+dmc, deter_dim 2048, tanh_normal, action_dim 6,
        # actor_grad=reinforce) at B=50, T=50, H=15 forward under torch.autocast('cpu', bfloat16) and in fp32
        run_amp('dmc_native_amp', ['defaults', 'dmc'],
                dict(batch_size=50, batch_length=50, imag_horizon=15, action_dim=6, actor_grad='reinforce'))
    if 'debug' in which:
        # BASELINE.json configs[0]: defaults+atari+debug on CPU, B=4,T=10,H=5, discrete(6)
        run('debug_literal', ['defaults', 'atari', 'debug'],
            dict(batch_size=4, batch_length=10, imag_horizon=5, action_dim=6), steps=1,
            full_grads=('wm.core.cell.gru.layers.0.bias_hh', 'ac.actor.model.12.weight'))
