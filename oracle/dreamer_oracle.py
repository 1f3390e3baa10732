"""CPU oracle for the DreamerV2 gradient step — TEST INFRASTRUCTURE ONLY.

This file is a plain-torch (CPU, fp32) restatement of the reference hot path
`Dreamer.training_step()` + the trainer section around it.  It exists so that the HIP path can be checked on a
machine where /root/reference does not exist.  It is imported ONLY by `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg.  The product path (`pydreamer_amd/`) never imports it and has no CPU fallback.

Parity pinning: the reference ships no tests or golden vectors for this path, so this oracle is
pinned against outputs of the reference itself, generated in the build container by `oracle/gen_golden.py`
(imports /root/reference, patches torch.multinomial with the inverse-CDF rule below) and committed under
`tests/golden/*.npz`; `tests/test_oracle_golden.py` replays them through this file.

Every function cites the reference lines it follows (paths relative to /root/reference).
Sampling contract: wherever the reference calls torch.multinomial (its only sampler call
site, torch/distributions/categorical.py:143-148), the oracle draws `idx = #{k : cdf_k <= u * cdf_last}` from explicit
uniforms `u`, with `cdf = cumsum(probs)` in fp32.
"""
import math
from argparse import Namespace
from collections import OrderedDict

import numpy as np
import torch
import torch.distributions as D
import torch.nn.functional as F

D.Distribution.set_default_validate_args(False)   # train.py:30

# ---------------------------------------------------------------------------------------------------------------
# config: the keys of config/defaults.yaml the hot path reads (dreamer.py:23-58,237-277; encoders.py:14-36;
# decoders.py:14-46).  Values = `defaults` section; `atari` overrides below (defaults.yaml:190-201).
# ---------------------------------------------------------------------------------------------------------------
DEFAULTS = dict(
    image_size=64, image_channels=3, action_dim=0,
    iwae_samples=1, kl_balance=0.8, kl_weight=1.0, image_weight=1.0, vecobs_weight=1.0, reward_weight=1.0,
    terminal_weight=1.0, adam_lr=3.0e-4, adam_lr_actor=1.0e-4, adam_lr_critic=1.0e-4, adam_eps=1.0e-5,
    batch_length=48, batch_size=32, grad_clip=200, grad_clip_ac=200,
    deter_dim=2048, stoch_dim=32, stoch_discrete=32, hidden_dim=1000, gru_layers=1, gru_type='gru', layer_norm=True,
    cnn_depth=48, reward_decoder_layers=4, terminal_decoder_layers=4,
    gamma=0.995, lambda_gae=0.95, entropy=0.003, target_interval=100, imag_horizon=15,
    actor_grad='reinforce', actor_dist='onehot',
    aux_critic=False, aux_critic_weight=1.0, gamma_aux=0.99, lambda_gae_aux=0.95, target_interval_aux=1000,
    probe_gradients=False,
)
ATARI = dict(action_dim=18, deter_dim=1024, kl_weight=0.1, gamma=0.99, entropy=0.001)


def make_conf(*sections, **overrides):
    c = dict(DEFAULTS)
    for s in sections:
        c.update(s)
    c.update(overrides)
    return Namespace(**c)


def tiny_conf(**overrides):
    """SURVEY.md 8(c) golden config (1): small dims, heads stay 400 wide."""
    base = dict(deter_dim=64, hidden_dim=64, stoch_dim=8, stoch_discrete=8, cnn_depth=8, action_dim=6,
                batch_length=5, batch_size=3, imag_horizon=4)
    base.update(overrides)
    return make_conf(ATARI, **base)


def atari_literal_conf(**overrides):
    """BASELINE.json configs[1]: B=50,T=50,H=15,deter=600,stoch 32x32."""
    base = dict(batch_size=50, batch_length=50, deter_dim=600, action_dim=18)
    base.update(overrides)
    return make_conf(ATARI, **base)


def feature_dim(conf):
    return conf.deter_dim + conf.stoch_dim * (conf.stoch_discrete or 1)       # rssm.py:103: Gaussian latents are stoch_dim wide


# ---------------------------------------------------------------------------------------------------------------
# parameters: the reference's state_dict keys and shapes; closed-form deterministic values
# ---------------------------------------------------------------------------------------------------------------
MLP_HIDDEN = 400   # a2c.py:16, decoders.py:259,289


def _mlp_shapes(prefix, in_dim, out_dim, layers, out, layer_norm=True):
    dim = in_dim
    for i in range(layers):                                 # common.py:43-50
        out[f'{prefix}.{3 * i}.weight'] = (MLP_HIDDEN, dim)
        out[f'{prefix}.{3 * i}.bias'] = (MLP_HIDDEN,)
        if layer_norm:                                      # NoNorm (common.py:68-74) has no parameters
            out[f'{prefix}.{3 * i + 1}.weight'] = (MLP_HIDDEN,)
            out[f'{prefix}.{3 * i + 1}.bias'] = (MLP_HIDDEN,)
        dim = MLP_HIDDEN
    out[f'{prefix}.{3 * layers}.weight'] = (out_dim, dim)   # common.py:51-53
    out[f'{prefix}.{3 * layers}.bias'] = (out_dim,)


def param_shapes(conf):
    d, ch = conf.cnn_depth, conf.image_channels
    D_, Hd, Z, A = conf.deter_dim, conf.hidden_dim, conf.stoch_dim * (conf.stoch_discrete or 1), conf.action_dim
    ZP = conf.stoch_dim * (conf.stoch_discrete or 2)       # width of the prior / posterior parameters (rssm.py:112,117)
    Fd, E = feature_dim(conf), conf.cnn_depth * 32
    s = OrderedDict()
    enc = 'wm.encoder.encoder_image.model'
    for i, (ci, co) in enumerate([(ch, d), (d, 2 * d), (2 * d, 4 * d), (4 * d, 8 * d)]):   # encoders.py:80-89
        s[f'{enc}.{2 * i}.weight'] = (co, ci, 4, 4)
        s[f'{enc}.{2 * i}.bias'] = (co,)
    dec = 'wm.decoder.image.model'
    s[f'{dec}.0.weight'] = (32 * d, Fd)                                                    # decoders.py:127-129
    s[f'{dec}.0.bias'] = (32 * d,)
    for i, (ci, co, k) in enumerate([(32 * d, 4 * d, 5), (4 * d, 2 * d, 5), (2 * d, d, 6), (d, ch, 6)]):   # decoders.py:149-155
        s[f'{dec}.{2 + 2 * i}.weight'] = (ci, co, k, k)
        s[f'{dec}.{2 + 2 * i}.bias'] = (co,)
    _mlp_shapes('wm.decoder.reward.model.model', Fd, 1, conf.reward_decoder_layers, s, conf.layer_norm)
    _mlp_shapes('wm.decoder.terminal.model.model', Fd, 1, conf.terminal_decoder_layers, s, conf.layer_norm)
    c = 'wm.core.cell'                                                                     # rssm.py:103-116
    s[f'{c}.z_mlp.weight'] = (Hd, Z); s[f'{c}.z_mlp.bias'] = (Hd,)
    s[f'{c}.a_mlp.weight'] = (Hd, A)
    if conf.layer_norm:
        s[f'{c}.in_norm.weight'] = (Hd,); s[f'{c}.in_norm.bias'] = (Hd,)
    GL = int(getattr(conf, 'gru_layers', 1))                                               # GRUCellStack, rnn.py:43-57
    ls = D_ // GL
    assert ls * GL == D_, 'Must be divisible'
    for li in range(GL):
        gl, kin = f'{c}.gru.layers.{li}', (Hd if li == 0 else ls)
        if conf.gru_type == 'gru':                                                         # nn.GRUCell, rnn.py:47-48
            s[f'{gl}.weight_ih'] = (3 * ls, kin); s[f'{gl}.weight_hh'] = (3 * ls, ls)
            s[f'{gl}.bias_ih'] = (3 * ls,); s[f'{gl}.bias_hh'] = (3 * ls,)
        elif conf.gru_type == 'gru_layernorm':                                             # NormGRUCell, rnn.py:95-104
            s[f'{gl}.weight_ih.weight'] = (3 * ls, kin); s[f'{gl}.weight_hh.weight'] = (3 * ls, ls)
            for n in ('ln_reset', 'ln_update', 'ln_newval'):
                s[f'{gl}.{n}.weight'] = (ls,); s[f'{gl}.{n}.bias'] = (ls,)
        elif conf.gru_type == 'gru_layernorm_dv2':                                         # NormGRUCellLateReset, rnn.py:117-125
            s[f'{gl}.weight_ih.weight'] = (3 * ls, kin); s[f'{gl}.weight_hh.weight'] = (3 * ls, ls)
            s[f'{gl}.lnorm.weight'] = (3 * ls,); s[f'{gl}.lnorm.bias'] = (3 * ls,)
        else:
            raise ValueError(conf.gru_type)
    s[f'{c}.prior_mlp_h.weight'] = (Hd, D_); s[f'{c}.prior_mlp_h.bias'] = (Hd,)
    if conf.layer_norm:
        s[f'{c}.prior_norm.weight'] = (Hd,); s[f'{c}.prior_norm.bias'] = (Hd,)
    s[f'{c}.prior_mlp.weight'] = (ZP, Hd); s[f'{c}.prior_mlp.bias'] = (ZP,)
    s[f'{c}.post_mlp_h.weight'] = (Hd, D_); s[f'{c}.post_mlp_h.bias'] = (Hd,)
    s[f'{c}.post_mlp_e.weight'] = (Hd, E)
    if conf.layer_norm:
        s[f'{c}.post_norm.weight'] = (Hd,); s[f'{c}.post_norm.bias'] = (Hd,)
    s[f'{c}.post_mlp.weight'] = (ZP, Hd); s[f'{c}.post_mlp.bias'] = (ZP,)
    if conf.aux_critic:                                                                    # dreamer.py:267-277 (a full ActorCritic)
        _mlp_shapes('wm.ac_aux.actor.model', Fd, A if conf.actor_dist == 'onehot' else 2 * A, 4, s, conf.layer_norm)
        _mlp_shapes('wm.ac_aux.critic.model', Fd, 1, 4, s, conf.layer_norm)
        _mlp_shapes('wm.ac_aux.critic_target.model', Fd, 1, 4, s, conf.layer_norm)
    _mlp_shapes('ac.actor.model', Fd, A if conf.actor_dist == 'onehot' else 2 * A, 4, s, conf.layer_norm)   # a2c.py:35-39
    _mlp_shapes('ac.critic.model', Fd, 1, 4, s, conf.layer_norm)
    _mlp_shapes('ac.critic_target.model', Fd, 1, 4, s, conf.layer_norm)
    s['probe_model.dummy'] = (1,)                                                          # probes.py:144
    return s


def make_params(conf, seed=0, dtype=torch.float32):
    """Deterministic closed-form parameter values (numpy RandomState per tensor) — NOT an init scheme of the
    reference; golden fixtures and parity tests load explicit weights on both sides.
    Scales are xavier-like so activations and losses are in the regime of a freshly initialised model."""
    shapes = param_shapes(conf)
    out = OrderedDict()
    for i, (name, shape) in enumerate(shapes.items()):
        rs = np.random.RandomState(seed * 100003 + i)
        if name == 'probe_model.dummy':
            v = np.full(shape, 0.25, dtype=np.float64)
        elif len(shape) == 1:
            is_ln_weight = name.endswith('norm.weight') or (name.endswith('.weight') and len(shape) == 1)
            v = (1.0 if is_ln_weight else 0.0) + 0.05 * rs.uniform(-1, 1, shape)
        else:
            if len(shape) == 4:
                fan_in = shape[1] * shape[2] * shape[3]
                fan_out = shape[0] * shape[2] * shape[3]
            else:
                fan_out, fan_in = shape
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            v = rs.uniform(-lim, lim, shape)
        out[name] = torch.tensor(v, dtype=dtype)
    return out


def group_of(name):
    """Optimizer group of a parameter (dreamer.py:60-66); critic_target has no optimizer."""
    if name.startswith('wm.'):
        return 'wm'
    if name.startswith('probe_model.'):
        return 'probe'
    if name.startswith('ac.actor.'):
        return 'actor'
    if name.startswith('ac.critic.'):
        return 'critic'
    return None


# ---------------------------------------------------------------------------------------------------------------
# synthetic replay batch; same generator for tests and bench
# ---------------------------------------------------------------------------------------------------------------
def synthetic_batch(conf, seed=1234, first=True):
    T, B, A = conf.batch_length, conf.batch_size, conf.action_dim
    rs = np.random.RandomState(seed)
    image_u8 = rs.randint(0, 256, (T, B, conf.image_size, conf.image_size, conf.image_channels)).astype(np.uint8)
    act_idx = rs.randint(0, A, (T, B))
    reward = np.tanh(rs.randn(T, B)).astype(np.float32)                     # clip_rewards: tanh (defaults.yaml:196)
    terminal = (rs.rand(T, B) < 0.005).astype(np.float32)
    reset = np.zeros((T, B), dtype=bool)
    reset[0] = (rs.rand(B) < 1.0 / 200)                                     # reset_interval 200 (defaults.yaml:36)
    if first:
        reset[0, 0] = True
    reset[T // 2, B - 1] = True                                             # exercise a mid-sequence reset (allow_mid_reset)
    return dict(image_u8=image_u8, action_idx=act_idx.astype(np.int64), reward=reward, terminal=terminal, reset=reset)


def preprocess(raw, conf, device='cpu'):
    """preprocessing.py:21-29,135-150: u8 HWC -> float CHW in [-0.5,0.5]; one-hot actions."""
    image = torch.from_numpy(raw['image_u8']).to(device).float().div(255.0).sub(0.5).permute(0, 1, 4, 2, 3).contiguous()
    action = F.one_hot(torch.from_numpy(raw['action_idx']).to(device), conf.action_dim).float()
    return dict(image=image, action=action, reward=torch.from_numpy(raw['reward']).to(device),
                terminal=torch.from_numpy(raw['terminal']).to(device), reset=torch.from_numpy(raw['reset']).to(device))


def make_noise(conf, seed=777, eval_iwae=None):
    """Uniforms in reference call order: T posterior draws (B*I*S), then H x [actor (M), prior (M*S)].
    Gaussian latents (stoch_discrete = 0): the latent arrays (u_post, u_prior, u_pred, u_prior_log) hold STANDARD-NORMAL
    draws instead - the eps of Normal.rsample (z = mean + std * eps).
    eval_iwae = I: the evaluation call shape (train.py:353-359,380-385 pass iwae_samples=eval_samples to a model whose own
    conf.iwae_samples is 1): every per-sample array is sized for I samples, the do_dream_tensors log dream for the B first
    states (dreamer.py:170: states[0, :, 0])."""
    T, B, S, H = conf.batch_length, conf.batch_size, conf.stoch_dim, conf.imag_horizon
    Isamp = int(eval_iwae or conf.iwae_samples)
    M = T * B * Isamp
    rs = np.random.RandomState(seed)
    lat = (lambda *shape: rs.rand(*shape)) if conf.stoch_discrete else (lambda *shape: rs.randn(*shape))
    out = dict(u_post=torch.tensor(lat(T, B * Isamp, S), dtype=torch.float32),
               u_act=torch.tensor(rs.rand(H, M), dtype=torch.float32),
               u_prior=torch.tensor(lat(H, M, S), dtype=torch.float32))
    # continuous actors draw normal noise instead (torch.normal in Normal.sample): x = mean + std * eps.  The noise is
    # scaled by 0.25 so that tanh(x) stays away from +-1: the reference's TanhTransform inverse is an un-clamped atanh,
    # which returns inf (and a NaN loss_actor) once a sampled action saturates in fp32 — a property of the reference
    # that parity data must avoid, not reproduce
    out['eps_act'] = torch.tensor(0.25 * rs.randn(H, M, conf.action_dim), dtype=torch.float32)
    # logging variants (drawn AFTER everything above, so the streams of the plain step are unchanged):
    #   do_image_pred: one prior sample per (t,b) (dreamer.py:383); do_dream_tensors: a (T-1)-step dream from the B first states
    Bi = B * Isamp
    out['u_pred'] = torch.tensor(lat(T, Bi, S), dtype=torch.float32)
    if eval_iwae:
        Bi = B
    out['u_act_log'] = torch.tensor(rs.rand(T - 1, Bi), dtype=torch.float32)
    out['u_prior_log'] = torch.tensor(lat(T - 1, Bi, S), dtype=torch.float32)
    out['eps_act_log'] = torch.tensor(0.25 * rs.randn(T - 1, Bi, conf.action_dim), dtype=torch.float32)
    return out


# ---------------------------------------------------------------------------------------------------------------
# building blocks
# ---------------------------------------------------------------------------------------------------------------
def sample_inverse_cdf(probs, u):
    """idx = #{k : cdf_k <= u*cdf_last}; probs (..., C), u (...)."""
    cdf = torch.cumsum(probs, -1)
    target = u.unsqueeze(-1) * cdf[..., -1:]
    return (cdf <= target).sum(-1).clamp(max=probs.shape[-1] - 1)


def _norm(p, name, x):
    """nn.LayerNorm(eps=1e-3), or NoNorm (common.py:68-74: identity, no parameters) when the model was built with
    layer_norm=False - told apart by the presence of the parameters."""
    if f'{name}.weight' not in p:
        return x
    return F.layer_norm(x, (x.shape[-1],), p[f'{name}.weight'], p[f'{name}.bias'], 1e-3)


def mlp(p, prefix, x, layers):
    """common.py:37-65."""
    lead = x.shape[:-1]
    y = x.reshape(-1, x.shape[-1])
    for i in range(layers):
        y = F.linear(y, p[f'{prefix}.{3 * i}.weight'], p[f'{prefix}.{3 * i}.bias'])
        y = F.elu(_norm(p, f'{prefix}.{3 * i + 1}', y))
    y = F.linear(y, p[f'{prefix}.{3 * layers}.weight'], p[f'{prefix}.{3 * layers}.bias'])
    if y.shape[-1] == 1:
        return y.reshape(lead)            # nn.Flatten(0) + unflatten_batch, common.py:54-57,61-65
    return y.reshape(lead + (y.shape[-1],))


def conv_encoder(p, image):
    """encoders.py:72-96."""
    T, B = image.shape[:2]
    x = image.reshape((-1,) + image.shape[2:])
    for i in range(4):
        x = F.elu(F.conv2d(x, p[f'wm.encoder.encoder_image.model.{2 * i}.weight'],
                           p[f'wm.encoder.encoder_image.model.{2 * i}.bias'], stride=2))
    return x.reshape(T, B, -1)


def conv_decoder(p, features):
    """decoders.py:144-161."""
    lead = features.shape[:-1]
    x = F.linear(features.reshape(-1, features.shape[-1]), p['wm.decoder.image.model.0.weight'],
                 p['wm.decoder.image.model.0.bias'])
    x = x.reshape(x.shape[0], -1, 1, 1)
    for i in range(4):
        x = F.conv_transpose2d(x, p[f'wm.decoder.image.model.{2 + 2 * i}.weight'],
                               p[f'wm.decoder.image.model.{2 + 2 * i}.bias'], stride=2)
        if i < 3:
            x = F.elu(x)
    return x.reshape(lead + x.shape[1:])


def gru_cell(p, x, h):
    """GRUCellStack (rnn.py:40-67): layer i takes the new state of layer i-1 as input and the i-th chunk of the incoming
    state as its own state; the new states are concatenated (rnn.py:59-67)."""
    n = 1
    while any(k.startswith(f'wm.core.cell.gru.layers.{n}.') for k in p):
        n += 1
    outs = []
    for i, hi in enumerate(h.chunk(n, -1)):
        x = gru_layer(p, x, hi, i)
        outs.append(x)
    return torch.cat(outs, -1) if n > 1 else outs[0]


def gru_layer(p, x, h, layer=0):
    """One cell of the stack: nn.GRUCell (gate order r,z,n), NormGRUCell (rnn.py:95-114) or NormGRUCellLateReset
    (rnn.py:117-138), told apart by the parameter names of the cell."""
    c = f'wm.core.cell.gru.layers.{layer}'
    ln = lambda v, n: F.layer_norm(v, (v.shape[-1],), p[f'{c}.{n}.weight'], p[f'{c}.{n}.bias'], 1e-3)
    if f'{c}.lnorm.weight' in p:                               # gru_layernorm_dv2
        gates = ln(F.linear(x, p[f'{c}.weight_ih.weight']) + F.linear(h, p[f'{c}.weight_hh.weight']), 'lnorm')
        reset, update, newval = gates.chunk(3, -1)
        reset = torch.sigmoid(reset)
        update = torch.sigmoid(update - 1)                     # update_bias = -1
        newval = torch.tanh(reset * newval)                    # late reset
        return update * newval + (1 - update) * h
    if f'{c}.ln_reset.weight' in p:                            # gru_layernorm
        i_r, i_u, i_n = F.linear(x, p[f'{c}.weight_ih.weight']).chunk(3, -1)
        h_r, h_u, h_n = F.linear(h, p[f'{c}.weight_hh.weight']).chunk(3, -1)
        reset = torch.sigmoid(ln(i_r + h_r, 'ln_reset'))
        update = torch.sigmoid(ln(i_u + h_u, 'ln_update'))
        newval = torch.tanh(ln(i_n + reset * h_n, 'ln_newval'))
        return update * newval + (1 - update) * h
    gi = F.linear(x, p[f'{c}.weight_ih'], p[f'{c}.bias_ih'])
    gh = F.linear(h, p[f'{c}.weight_hh'], p[f'{c}.bias_hh'])
    i_r, i_z, i_n = gi.chunk(3, -1)
    h_r, h_z, h_n = gh.chunk(3, -1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (h - n) * z + n


def gaussian_params(pp, min_std=0.1, max_std=2.0):
    """functions.py:46-56 diag_normal: mean, std = chunk(2); std = max_std * sigmoid(std) + min_std."""
    mean, std = pp.chunk(2, -1)
    return mean, max_std * torch.sigmoid(std) + min_std


def zdistr(conf, logits):
    """rssm.py:195-203."""
    if not conf.stoch_discrete:
        mean, std = gaussian_params(logits)
        return D.Independent(D.Normal(mean, std), 1)
    lg = logits.reshape(logits.shape[:-1] + (conf.stoch_dim, conf.stoch_discrete)).float()
    return D.Independent(D.OneHotCategoricalStraightThrough(logits=lg), 1)


def st_sample(conf, logits, u, forced_idx=None):
    """OneHotCategoricalStraightThrough.rsample with explicit noise: onehot + (probs - probs.detach()).
    Gaussian latents: Normal.rsample = mean + std * eps with `u` holding eps; the index slot carries zeros."""
    S, C = conf.stoch_dim, conf.stoch_discrete
    if not C:
        mean, std = gaussian_params(logits)
        return mean + std * u, torch.zeros(logits.shape[:-1] + (S,), dtype=torch.long)
    lg = logits.reshape(logits.shape[:-1] + (S, C)).float()
    lg = lg - lg.logsumexp(-1, keepdim=True)            # Categorical normalises logits, then probs = softmax
    probs = torch.softmax(lg, -1)
    idx = sample_inverse_cdf(probs.detach(), u) if forced_idx is None else forced_idx.long()
    onehot = F.one_hot(idx, C).to(probs.dtype)
    sample = onehot + (probs - probs.detach())
    return sample.reshape(logits.shape[:-1] + (S * C,)), idx


def cell_trunk(p, action, h, z):
    c = 'wm.core.cell'
    x = F.linear(z, p[f'{c}.z_mlp.weight'], p[f'{c}.z_mlp.bias']) + F.linear(action, p[f'{c}.a_mlp.weight'])
    return gru_cell(p, F.elu(_norm(p, f'{c}.in_norm', x)), h)


def prior_head(p, h):
    """rssm.py:174-177 / 186-193."""
    c = 'wm.core.cell'
    x = F.linear(h, p[f'{c}.prior_mlp_h.weight'], p[f'{c}.prior_mlp_h.bias'])
    x = F.elu(_norm(p, f'{c}.prior_norm', x))
    return F.linear(x, p[f'{c}.prior_mlp.weight'], p[f'{c}.prior_mlp.bias'])


def cell_forward(p, conf, embed, action, reset_mask, h, z, u, forced_idx=None):
    """RSSMCell.forward, rssm.py:125-153."""
    c = 'wm.core.cell'
    h = h * reset_mask
    z = z * reset_mask
    h = cell_trunk(p, action, h, z)
    x = F.linear(h, p[f'{c}.post_mlp_h.weight'], p[f'{c}.post_mlp_h.bias']) + F.linear(embed, p[f'{c}.post_mlp_e.weight'])
    x = F.elu(_norm(p, f'{c}.post_norm', x))
    post = F.linear(x, p[f'{c}.post_mlp.weight'], p[f'{c}.post_mlp.bias'])
    sample, idx = st_sample(conf, post, u, forced_idx)
    return post, h, sample, idx


def cell_forward_prior(p, conf, action, h, z, u, reset_mask=None):
    """RSSMCell.forward_prior, rssm.py:155-184 (reset_mask=None in the dream; given in the open-loop sequence)."""
    if reset_mask is not None:
        h = h * reset_mask
        z = z * reset_mask
    h = cell_trunk(p, action, h, z)
    prior = prior_head(p, h)
    sample, idx = st_sample(conf, prior, u)
    return prior, h, sample, idx


# ---------------------------------------------------------------------------------------------------------------
# world model (dreamer.py:297-396)
# ---------------------------------------------------------------------------------------------------------------
def nanmean(x):
    """functions.py:149-150."""
    return torch.nansum(x) / (~torch.isnan(x)).sum()


def grad_probe(g, pidx):
    """Two closed-form directions per parameter (no RNG involved): slim full-size fixtures store the projections of every
    gradient onto them next to its norm, which pins the gradient's DIRECTION in 16 bytes (a norm alone does not)."""
    v = g.detach().double().reshape(-1).cpu()
    i = torch.arange(v.numel(), dtype=torch.float64)
    w1 = torch.cos(i * 0.7548776662466927 + float(pidx))
    w2 = torch.cos(i * 0.5698402909980532 + 2.0 * float(pidx) + 0.5)
    return float((v * w1).sum()), float((v * w2).sum())


def logavgexp(x, dim):
    """functions.py:97-102."""
    if x.size(dim) > 1:
        return x.logsumexp(dim=dim) - math.log(x.size(dim))
    return x.squeeze(dim)


def wm_training_step(p, conf, obs, in_state, u_post, forced_idx=None, u_pred=None, do_open_loop=False, iwae_samples=None):
    """WorldModel.training_step (dreamer.py:297-396) with RSSMCore.forward (rssm.py:21-78) inlined.  iwae_samples = I > 1:
    the batch is multiplied by I (rssm.py:35-41: row b*I + i), the KL term becomes the sampled one (dreamer.py:340-343) and
    loss_model = -logavgexp(-loss_tbi) over I (dreamer.py:362-365).  u_post is (T, B*I, S); in_state is (B*I, .)."""
    I = int(iwae_samples or conf.iwae_samples)
    T, B = obs['action'].shape[:2]
    embed = conv_encoder(p, obs['image'])                                     # dreamer.py:307
    h, z = in_state
    expand = lambda x: x.unsqueeze(2).expand(T, B, I, x.shape[-1]).reshape(T, B * I, x.shape[-1])      # rssm.py:35-37
    embeds, actions = expand(embed), expand(obs['action'])
    reset_masks = expand((~obs['reset']).unsqueeze(-1).to(embed.dtype))       # rssm.py:41
    posts, hs, zs, idxs = [], [], [], []
    for t in range(T):                                                        # rssm.py:49-56
        if do_open_loop:                                                      # rssm.py:53: post = prior, no embed
            post, h, z, idx = cell_forward_prior(p, conf, actions[t], h, z, u_post[t], reset_masks[t])
        else:
            post, h, z, idx = cell_forward(p, conf, embeds[t], actions[t], reset_masks[t], h, z, u_post[t],
                                           None if forced_idx is None else forced_idx[t])
        posts.append(post); hs.append(h); zs.append(z); idxs.append(idx)
    posts, hs, zs = torch.stack(posts), torch.stack(hs), torch.stack(zs)
    priors = prior_head(p, hs)                                                # rssm.py:61
    features = torch.cat((hs, zs), -1)                                        # rssm.py:62,83-84
    out_state = (h.detach(), z.detach())                                      # rssm.py:77
    feat_tbi = features.reshape(T, B, I, -1)

    # decoders (decoders.py:50-108)
    decoded = conv_decoder(p, feat_tbi)                                       # (T,B,I,C,H,W)
    target = obs['image'].unsqueeze(2)
    loss_image = 0.5 * torch.square(decoded - target).sum(dim=[-1, -2, -3])   # decoders.py:163-167
    std = 0.3989422804
    mu = mlp(p, 'wm.decoder.reward.model.model', feat_tbi, conf.reward_decoder_layers)
    loss_reward = -D.Normal(mu, torch.ones_like(mu) * std).log_prob(obs['reward'].unsqueeze(2)) * std ** 2   # decoders.py:296-304
    tl = mlp(p, 'wm.decoder.terminal.model.model', feat_tbi, conf.terminal_decoder_layers)
    tdist = D.Bernoulli(logits=tl.float())
    loss_terminal = -tdist.log_prob(obs['terminal'].unsqueeze(2))             # decoders.py:263-269
    loss_reconstr = conf.image_weight * loss_image + conf.reward_weight * loss_reward + conf.terminal_weight * loss_terminal

    # KL (dreamer.py:326-343)
    prior_tbi, post_tbi = priors.reshape(T, B, I, -1), posts.reshape(T, B, I, -1)
    dprior, dpost = zdistr(conf, prior_tbi), zdistr(conf, post_tbi)
    loss_kl_exact = D.kl_divergence(dpost, dprior)
    if I > 1:                                                                 # sampled KL for IWAE, dreamer.py:340-343
        zs_tbi = zs.reshape(dpost.batch_shape + dpost.event_shape)
        loss_kl = dpost.log_prob(zs_tbi) - dprior.log_prob(zs_tbi)
    elif conf.kl_balance == 0.5:
        loss_kl = loss_kl_exact
    else:
        postgrad = D.kl_divergence(dpost, zdistr(conf, prior_tbi.detach()))
        priograd = D.kl_divergence(zdistr(conf, post_tbi.detach()), dprior)
        loss_kl = (1 - conf.kl_balance) * postgrad + conf.kl_balance * priograd
    loss_model_tbi = conf.kl_weight * loss_kl + loss_reconstr                 # dreamer.py:362-365
    loss_model_tb = -logavgexp(-loss_model_tbi, 2)
    loss = loss_model_tb.mean()
    aux = None
    if conf.aux_critic:                                                       # dreamer.py:347-358 (assumes I = 1)
        (_, loss_critic_aux), m_aux, t_aux = ac_training_step(p, conf, feat_tbi[:, :, 0], obs['action'][1:], obs['reward'],
                                                              obs['terminal'], prefix='wm.ac_aux', gamma=conf.gamma_aux,
                                                              lam=conf.lambda_gae_aux)
        loss = loss + conf.aux_critic_weight * loss_critic_aux
        aux = (m_aux, t_aux)

    with torch.no_grad():
        ent_prior = dprior.entropy().mean(2)
        ent_post = dpost.entropy().mean(2)
        lae = lambda x: (-logavgexp(-x, 2)).detach()                          # decoders.py:170,277,312: TBI => TB
        tensors = dict(loss_kl=lae(loss_kl_exact), entropy_prior=ent_prior, entropy_post=ent_post,
                       loss_image=lae(loss_image), image_rec=decoded.mean(2).detach(),
                       loss_reward=lae(loss_reward), reward_rec=mu.mean(2).detach(),
                       loss_terminal=lae(loss_terminal), terminal_rec=tdist.mean.mean(2).detach())
        if aux is not None:
            tensors['policy_value_aux'] = aux[1]['value']
        metrics = dict(loss_model=loss_model_tb.mean(), loss_kl=tensors['loss_kl'].mean(),
                       entropy_prior=ent_prior.mean(), entropy_post=ent_post.mean(),
                       loss_image=tensors['loss_image'].mean(), loss_reward=tensors['loss_reward'].mean(),
                       loss_terminal=tensors['loss_terminal'].mean())
        if aux is not None:
            metrics.update(loss_critic_aux=aux[0]['loss_critic'], policy_value_aux=aux[0]['policy_value_im'])
    extras = dict(post_idx=torch.stack(idxs), post=posts.detach(), prior=priors.detach(), embed=embed.detach())
    if u_pred is not None:                                                    # do_image_pred, dreamer.py:381-394
        with torch.no_grad():      # I > 1: the decoders reduce TBI => TB by -logavgexp(-loss) / mean (decoders.py:170-171,277-278,312-313)
            z_prior, pred_idx = st_sample(conf, priors.detach(), u_pred)      # zdistr(prior).sample()
            fp = torch.cat((hs, z_prior), -1).reshape(T, B, I, -1).detach()   # feature_replace_z
            dec_p = conv_decoder(p, fp)
            li = lae(0.5 * torch.square(dec_p - target).sum(dim=[-1, -2, -3]))
            mu_p = mlp(p, 'wm.decoder.reward.model.model', fp, conf.reward_decoder_layers)
            lr = lae(-D.Normal(mu_p, torch.ones_like(mu_p) * std).log_prob(obs['reward'].unsqueeze(2)) * std ** 2)
            td = D.Bernoulli(logits=mlp(p, 'wm.decoder.terminal.model.model', fp, conf.terminal_decoder_layers).float())
            lt = lae(-td.log_prob(obs['terminal'].unsqueeze(2)))
            tensors.update(logprob_image=li, logprob_reward=lr, logprob_terminal=lt, image_pred=dec_p.mean(2),
                           reward_pred=mu_p.mean(2), terminal_pred=td.mean.mean(2))
            metrics.update(logprob_image=li.mean(), logprob_reward=lr.mean(), logprob_terminal=lt.mean())
            for sig in (-1, 1):                                               # decoders.py:94-100 (extra_metrics)
                m = torch.sign(obs['reward']) == sig
                lp = lr * m / m
                metrics[f'logprob_reward{sig}'] = nanmean(lp)
                tensors[f'logprob_reward{sig}'] = lp
            m = obs['terminal'] > 0                                           # decoders.py:102-105
            lp = lt * m / m
            metrics['logprob_terminal1'] = nanmean(lp)
            tensors['logprob_terminal1'] = lp
            extras['pred_idx'] = pred_idx
    return loss, feat_tbi, (hs.reshape(T, B, I, -1), zs.reshape(T, B, I, -1)), out_state, metrics, tensors, extras


# ---------------------------------------------------------------------------------------------------------------
# imagination (dreamer.py:188-216) and actor-critic (a2c.py:61-149)
# ---------------------------------------------------------------------------------------------------------------
def actor_distribution(conf, y):
    """ActorCritic.forward_actor (a2c.py:43-55) with functions.py:59-78 restated."""
    y = y.float()
    if conf.actor_dist == 'onehot':
        return D.OneHotCategorical(logits=y)
    mean_, std_ = y.chunk(2, -1)
    if conf.actor_dist == 'normal_tanh':                      # functions.py:59-66
        normal = D.Normal(torch.tanh(mean_), 1.0 * torch.sigmoid(std_) + 0.01)
        return D.Independent(normal, 1)
    if conf.actor_dist == 'tanh_normal':                      # functions.py:69-78
        normal = D.Independent(D.Normal(5 * torch.tanh(mean_ / 5), F.softplus(std_) + 0.1), 1)
        dist = D.TransformedDistribution(normal, [D.TanhTransform()])
        dist.entropy = normal.entropy                         # the reference's entropy "HACK"
        return dist
    raise AssertionError(conf.actor_dist)


def sample_continuous(conf, y, eps):
    """Distribution.sample() with torch.normal(mean, std) restated as mean + std * eps."""
    mean_, std_ = y.float().chunk(2, -1)
    if conf.actor_dist == 'normal_tanh':
        return torch.tanh(mean_) + (torch.sigmoid(std_) + 0.01) * eps
    return torch.tanh(5 * torch.tanh(mean_ / 5) + (F.softplus(std_) + 0.1) * eps)


def dream(p, conf, in_state, H, u_act, u_prior, eps_act=None):
    assert conf.actor_grad == 'reinforce'
    h, z = in_state
    feats, actions, act_idx, lat_idx = [], [], [], []
    with torch.no_grad():
        for i in range(H):
            feature = torch.cat((h, z), -1)
            logits = mlp(p, 'ac.actor.model', feature, 4).float()             # a2c.py:43-47
            if conf.actor_dist == 'onehot':
                lg = logits - logits.logsumexp(-1, keepdim=True)
                idx = sample_inverse_cdf(torch.softmax(lg, -1), u_act[i])     # dreamer.py:198-200
                action = F.one_hot(idx, conf.action_dim).float()
            else:
                action = sample_continuous(conf, logits, eps_act[i])
                idx = torch.zeros(action.shape[0], dtype=torch.long)
            feats.append(feature); actions.append(action); act_idx.append(idx)
            _, h, z, zi = cell_forward_prior(p, conf, action, h, z, u_prior[i])   # dreamer.py:205
            lat_idx.append(zi)
        feats.append(torch.cat((h, z), -1))
        feats = torch.stack(feats)
        actions = torch.stack(actions)
        rewards = mlp(p, 'wm.decoder.reward.model.model', feats, conf.reward_decoder_layers)           # Normal.mean
        terminals = torch.sigmoid(mlp(p, 'wm.decoder.terminal.model.model', feats, conf.terminal_decoder_layers).float())
    return feats, actions, rewards, terminals, dict(act_idx=torch.stack(act_idx), lat_idx=torch.stack(lat_idx))


def ac_training_step(p, conf, features, actions, rewards, terminals, prefix='ac', gamma=None, lam=None):
    """ActorCritic.training_step (a2c.py:61-149); prefix 'wm.ac_aux' + (gamma_aux, lambda_gae_aux) = the auxiliary critic."""
    gamma = conf.gamma if gamma is None else gamma
    lam = conf.lambda_gae if lam is None else lam
    reward1, terminal0, terminal1 = rewards[1:], terminals[:-1], terminals[1:]
    with torch.no_grad():
        value_t = mlp(p, f'{prefix}.critic_target.model', features, 4)
    value0t, value1t = value_t[:-1], value_t[1:]
    advantage = -value0t + reward1 + gamma * (1.0 - terminal1) * value1t
    gae, agae = [], None
    for adv, term in zip(reversed(advantage.unbind()), reversed(terminal1.unbind())):
        agae = adv if agae is None else adv + lam * gamma * (1.0 - term) * agae
        gae.append(agae)
    gae.reverse()
    advantage_gae = torch.stack(gae)
    value_target = advantage_gae + value0t
    reality_weight = (1 - terminal0).log().cumsum(dim=0).exp()

    value = mlp(p, f'{prefix}.critic.model', features, 4)
    value0 = value[:-1]
    loss_critic = (0.5 * torch.square(value_target.detach() - value0) * reality_weight).mean()

    logits = mlp(p, f'{prefix}.actor.model', features[:-1], 4).float()
    policy = actor_distribution(conf, logits)
    loss_policy = -policy.log_prob(actions) * advantage_gae.detach()
    policy_entropy = policy.entropy()
    loss_actor = ((loss_policy - conf.entropy * policy_entropy) * reality_weight).mean()

    with torch.no_grad():
        metrics = dict(loss_critic=loss_critic.detach(), loss_actor=loss_actor.detach(),
                       policy_entropy=policy_entropy.mean(), policy_value=value0[0].mean(),
                       policy_value_im=value0.mean(), policy_reward=reward1.mean(), policy_reward_std=reward1.std())
        tensors = dict(value=value.detach(), value_target=value_target.detach(), value_advantage=advantage.detach(),
                       value_advantage_gae=advantage_gae.detach(), value_weight=reality_weight.detach())
    return (loss_actor, loss_critic), metrics, tensors


# ---------------------------------------------------------------------------------------------------------------
# Dreamer.training_step (dreamer.py:113-186) and the trainer section (train.py:165-198)
# ---------------------------------------------------------------------------------------------------------------
class OracleDreamer:
    """Holds leaf parameter tensors (reference state_dict names), optimizers and the target-network counter."""

    def __init__(self, conf, params):
        self.conf = conf
        self.p = OrderedDict((k, v.clone().requires_grad_(group_of(k) is not None)) for k, v in params.items())
        self.train_steps = 0
        self.aux_train_steps = 0

    def group(self, g):
        return [v for k, v in self.p.items() if group_of(k) == g]

    def init_optimizers(self):
        c = self.conf
        mk = lambda g, lr: torch.optim.AdamW(self.group(g), lr=lr, eps=c.adam_eps)   # dreamer.py:60-66
        if getattr(c, 'probe_gradients', False):                                       # dreamer.py:67-71: the probe head has no optimizer
            self.optimizers = (mk('wm', c.adam_lr), mk('actor', c.adam_lr_actor), mk('critic', c.adam_lr_critic))
        else:
            self.optimizers = (mk('wm', c.adam_lr), mk('probe', c.adam_lr), mk('actor', c.adam_lr_actor),
                               mk('critic', c.adam_lr_critic))
        return self.optimizers

    def init_state(self, batch):
        c = self.conf
        return (torch.zeros(batch, c.deter_dim), torch.zeros(batch, c.stoch_dim * (c.stoch_discrete or 1)))

    def training_step(self, obs, in_state, noise, forced_idx=None, do_image_pred=False, do_dream_tensors=False,
                      do_open_loop=False, iwae_samples=None):
        c, p = self.conf, self.p
        T, B = obs['action'].shape[:2]
        I = int(iwae_samples or c.iwae_samples)
        if c.aux_critic:                                                           # a2c.py:76-79 inside wm.ac_aux
            if self.aux_train_steps % c.target_interval_aux == 0:
                with torch.no_grad():
                    for k in list(p):
                        if k.startswith('wm.ac_aux.critic_target.'):
                            p[k].copy_(p[k.replace('critic_target', 'critic')])
            self.aux_train_steps += 1
        loss_model, features, states, out_state, metrics, tensors, extras = \
            wm_training_step(p, c, obs, in_state, noise['u_post'], forced_idx, noise['u_pred'] if do_image_pred else None,
                             do_open_loop, iwae_samples=I)
        loss_probe = torch.square(p['probe_model.dummy'])                          # probes.py:146-150
        in_dream = tuple(x.detach().reshape(-1, x.shape[-1]) for x in states)      # dreamer.py:149
        if self.train_steps % c.target_interval == 0:                              # a2c.py:76-79
            with torch.no_grad():
                for k in list(p):
                    if k.startswith('ac.critic_target.'):
                        p[k].copy_(p[k.replace('critic_target', 'critic')])
        self.train_steps += 1
        feats, actions, rewards, terminals, dx = dream(p, c, in_dream, c.imag_horizon, noise['u_act'], noise['u_prior'],
                                                      noise.get('eps_act'))
        dx['actions'] = actions
        (loss_actor, loss_critic), m_ac, t_ac = ac_training_step(p, c, feats, actions, rewards, terminals)
        metrics.update(m_ac)
        tensors.update(policy_value=t_ac['value'][0].reshape(T, B, I).mean(-1))    # dreamer.py:159
        extras.update(dx)
        extras.update(dream_features=feats, ac_tensors=t_ac)
        if do_dream_tensors:                                                       # dreamer.py:163-180
            with torch.no_grad():
                first = tuple(x.detach()[0, :, 0] for x in states)                 # (T,B,I) => (B)
                f2, a2, r2, t2, dx2 = dream(p, c, first, T - 1, noise['u_act_log'], noise['u_prior_log'],
                                            noise.get('eps_act_log'))
                image_dream = conv_decoder(p, f2)
                _, _, t_ac2 = ac_training_step(p, c, f2, a2, r2, t2)               # log_only=True: same tensors
                extras['dream_tensors'] = dict(action_pred=torch.cat([obs['action'][:1], a2]), reward_pred=r2,
                                               terminal_pred=t2, image_pred=image_dream, **t_ac2)
                extras['dream_log_idx'] = dx2
        if getattr(c, 'probe_gradients', False):                                       # dreamer.py:183-186
            return (loss_model + loss_probe, loss_actor, loss_critic), out_state, metrics, tensors, extras
        return (loss_model, loss_probe, loss_actor, loss_critic), out_state, metrics, tensors, extras

    def backward_clip_step(self, losses):
        """train.py:184-198 with amp disabled."""
        c = self.conf
        for opt in self.optimizers:
            opt.zero_grad()
        for loss in losses:
            loss.backward()
        clip = torch.nn.utils.clip_grad_norm_
        if getattr(c, 'probe_gradients', False):                                       # dreamer.py:81-86
            grad_metrics = dict(grad_norm=clip(self.group('wm'), c.grad_clip),
                                grad_norm_actor=clip(self.group('actor'), c.grad_clip_ac),
                                grad_norm_critic=clip(self.group('critic'), c.grad_clip_ac))
        else:
            grad_metrics = dict(grad_norm=clip(self.group('wm'), c.grad_clip), grad_norm_probe=clip(self.group('probe'), c.grad_clip),
                                grad_norm_actor=clip(self.group('actor'), c.grad_clip_ac),
                                grad_norm_critic=clip(self.group('critic'), c.grad_clip_ac))
        grads = OrderedDict((k, v.grad.detach().clone()) for k, v in self.p.items() if v.grad is not None)
        for opt in self.optimizers:
            opt.step()
        return grad_metrics, grads
