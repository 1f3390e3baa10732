"""CPU (-m "not gpu"): host-side logic — the C-ABI library loads and exports every symbol include/dreamer_hip.h
declares (no compute without a GPU), config surface, state_dict compatibility with the reference's key table,
workspace/acts sizing, sharding bounds, loud failure off-device."""
import ctypes
import os
import re

import pytest
import torch

from oracle import dreamer_oracle as O
from pydreamer_amd import config, dist as DP, hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'dreamer_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(dm_[a-z0-9_]+)\s*\(', hdr)))
    lib = hip.lib()
    assert lib.dm_version() == hip.DM_ABI_VERSION == 13
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(hip.exported_symbols()) == declared, set(declared) ^ set(hip.exported_symbols())


def test_error_reporting_without_gpu():
    """Argument validation happens on the host: bad calls return DM_E_* with a message, never crash."""
    with pytest.raises(hip.DreamerHipError) as e:
        hip.call('dm_gemm_f32', 0, 0, 4, 4, 4, None, 4, None, 4, None, 4, None, None, 0, 0, None, 0, None)
    assert 'null' in str(e.value)
    with pytest.raises(hip.DreamerHipError):
        hip.call('dm_gae_losses', 0, 5, 0.99, 0.95, None, None, None, None, None, None, None, None)
    if not torch.cuda.is_available():
        with pytest.raises(hip.DreamerHipError) as e:
            hip.call('dm_device_check')
        assert 'gfx950' in str(e.value) or 'HIP device' in str(e.value)


def test_struct_layouts_match_header():
    assert ctypes.sizeof(hip.dm_shape) == 16 * 4
    assert ctypes.sizeof(hip.dm_reduce_item) == 32
    assert ctypes.sizeof(hip.dm_mlp_params) == 8 * (9 + 9 + 8 + 8) + 8          # + precision, reserved_ (ABI v3)
    assert ctypes.sizeof(hip.dm_mlp_grads) == 8 * (9 + 9 + 8 + 8)
    assert hip.dm_mlp_params.precision.offset == 8 * 34
    assert ctypes.sizeof(hip.dm_conv_params) == 8 * 10
    assert ctypes.sizeof(hip.dm_rssm_params) == 8 * 58        # + 12 GRUCellStack layer slots (ABI v4) + 18 LayerNorm ones (v6)
    names = hip.rssm_param_names('gru', 3)
    assert len(names) == hip.DM_RSSM_NPARAMS == 58 and names[28:36] == [f'gru.layers.{i}.{n}' for i in (1, 2) for n in
                                                                         ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
    assert names[36:] == [None] * 22 and hip.rssm_param_names('gru')[28:] == [None] * 30
    names = hip.rssm_param_names('gru_layernorm', 2)
    assert names[22:28] == [f'gru.layers.0.ln_{n}.{w}' for n in ('reset', 'update', 'newval') for w in ('weight', 'bias')]
    assert names[28:32] == ['gru.layers.1.weight_ih.weight', 'gru.layers.1.weight_hh.weight', None, None]
    assert names[40:46] == [f'gru.layers.1.ln_{n}.{w}' for n in ('reset', 'update', 'newval') for w in ('weight', 'bias')]
    assert names[32:40] == [None] * 8 and names[46:] == [None] * 12
    names = hip.rssm_param_names('gru_layernorm_dv2', 4)
    assert names[52:58] == ['gru.layers.3.lnorm.weight', 'gru.layers.3.lnorm.bias', None, None, None, None]


def test_config_surface():
    c = config.load_config('defaults', 'atari')
    assert (c.batch_size, c.batch_length, c.deter_dim, c.kl_weight, c.gamma, c.entropy, c.action_dim) == (32, 48, 1024, 0.1, 0.99, 0.001, 18)
    lit = config.atari_literal()
    assert (lit.batch_size, lit.batch_length, lit.imag_horizon, lit.deter_dim, lit.stoch_dim, lit.stoch_discrete) == (50, 50, 15, 600, 32, 32)
    dbg = config.load_config('defaults', 'atari', 'debug', batch_size=4, batch_length=10, imag_horizon=5, action_dim=6)
    assert (dbg.batch_size, dbg.imag_horizon, dbg.action_dim) == (4, 5, 6)
    with pytest.raises(KeyError):
        config.load_config('defaults', nonexistent_key=1)
    # oracle and product agree on every shared key
    oc = O.make_conf(O.ATARI)
    for k, v in vars(oc).items():
        assert getattr(c, k) == v, k


def test_state_dict_keys_match_reference_table():
    """oracle.param_shapes was asserted key-for-key against the reference's state_dict by gen_golden.py."""
    from pydreamer_amd.models import Dreamer
    for oconf in (O.tiny_conf(), O.atari_literal_conf(), O.tiny_conf(gru_type='gru_layernorm'),
                  O.tiny_conf(gru_type='gru_layernorm_dv2'), O.tiny_conf(aux_critic=True), O.tiny_conf(gru_layers=2),
                  O.atari_literal_conf(gru_layers=3), O.tiny_conf(layer_norm=False), O.tiny_conf(layer_norm=False, aux_critic=True),
                  O.tiny_conf(stoch_discrete=0), O.atari_literal_conf(stoch_discrete=0),
                  O.tiny_conf(gru_type='gru_layernorm', gru_layers=2), O.tiny_conf(gru_type='gru_layernorm_dv2', gru_layers=4),
                  O.tiny_conf(stoch_discrete=0, aux_critic=True, layer_norm=False, gru_layers=2, actor_dist='tanh_normal', action_dim=4)):
        shapes = O.param_shapes(oconf)
        conf = config.load_config('defaults', 'atari', **vars(oconf))
        with torch.device('meta'):
            sd = Dreamer(conf).state_dict()
        assert list(sd.keys()) == list(shapes.keys())
        for k, s in shapes.items():
            assert tuple(sd[k].shape) == tuple(s), k


def test_unsupported_configs_fail_loudly():
    from pydreamer_amd.models import Dreamer
    for kw in (dict(aux_critic=True, iwae_samples=2), dict(gru_layers=5, deter_dim=1000), dict(gru_layers=3, deter_dim=1002),
               dict(gru_type='bogus'), dict(actor_dist='bogus'), dict(actor_grad='dynamics'),
               dict(image_size=32)):
        conf = config.load_config('defaults', 'atari', **kw)
        with pytest.raises(NotImplementedError):
            Dreamer(conf)


def test_workspace_and_acts_sizing():
    lib = hip.lib()
    tiny = hip.make_shape(T=5, B=3, I=1, H=4, D=64, Hd=64, S=8, C=8, E=256, A=6, mlp_hidden=400, mlp_layers=4,
                          cnn_depth=8, img=64, img_ch=3, flags=0)
    lit = hip.make_shape(T=50, B=50, I=1, H=15, D=600, Hd=1000, S=32, C=32, E=1536, A=18, mlp_hidden=400, mlp_layers=4,
                         cnn_depth=48, img=64, img_ch=3, flags=0)
    assert 64 << 20 <= hip.workspace_bytes(tiny) < 256 << 20
    big = hip.workspace_bytes(lit)
    assert 3 << 30 < big < 8 << 30, big          # dominated by the decoder layer-3 column matrix (2.9 GB) + grads
    enc = int(lib.dm_conv_encoder_acts_floats(ctypes.byref(lit))) * 4
    dec = int(lib.dm_conv_decoder_acts_floats(ctypes.byref(lit))) * 4
    rssm = int(lib.dm_rssm_acts_floats(ctypes.byref(lit))) * 4
    # patch matrices are implicit; the encoder's layer-1 patch matrix (461 MB) is gone too since layer 1 is a direct kernel
    assert 0.6e9 < enc < 1.0e9 and 0.5e9 < dec < 1.5e9 and 0.1e9 < rssm < 0.3e9, (enc, dec, rssm)
    assert int(lib.dm_mlp_acts_floats(2500, 400, 4)) >= 2500 * (400 * 2 + 2) * 4


def test_shard_bounds_cover_batch():
    for B in (50, 7, 8, 1):
        for world in (1, 2, 4, 8):
            spans = [DP.shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert [hi - lo for lo, hi in (DP.shard_bounds(50, 8, r) for r in range(8))] == [7, 7, 6, 6, 6, 6, 6, 6]


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under pydreamer_amd/ may reference it."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'pydreamer_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in text and 'from oracle' not in text and 'dreamer_oracle' not in text, f


def test_init_weights_tf2_matches_reference_rule():
    """A22 (functions.py:81-94, applied to `wm` only, dreamer.py:283-284): xavier-uniform conv / linear weights with zero
    biases, GRUCell xavier weight_ih + orthogonal weight_hh + zero biases, LayerNorm left at (1, 0); `ac` keeps torch's
    default Linear init (kaiming-uniform(a=sqrt 5) weight, uniform(+-1/sqrt(fan_in)) bias)."""
    import math
    import torch
    from pydreamer_amd import config
    from pydreamer_amd.models import Dreamer, LinearP, ConvP, GRUCellP, LayerNormP
    torch.manual_seed(3)
    conf = config.load_config('defaults', 'atari', deter_dim=64, hidden_dim=64, stoch_dim=8, stoch_discrete=8, cnn_depth=8,
                              action_dim=6)
    m = Dreamer(conf)
    n_lin = n_conv = 0
    for mod in m.wm.modules():
        if isinstance(mod, (LinearP, ConvP)):
            w = mod.weight.data
            rf = w[0][0].numel() if w.dim() > 2 else 1
            fan_in, fan_out = w.shape[1] * rf, w.shape[0] * rf
            bound = math.sqrt(6.0 / (fan_in + fan_out))
            assert float(w.abs().max()) <= bound + 1e-7, type(mod).__name__
            if w.numel() >= 2000:                       # uniform(-b, b): std = b / sqrt(3)
                assert abs(float(w.std()) - bound / math.sqrt(3)) < 0.1 * bound
            if mod.bias is not None:
                assert float(mod.bias.data.abs().max()) == 0.0
            n_lin += isinstance(mod, LinearP)
            n_conv += isinstance(mod, ConvP)
        if isinstance(mod, GRUCellP):
            hh = mod.weight_hh.data.double()             # (3D, D) with orthonormal columns
            eye = hh.t() @ hh
            assert float((eye - torch.eye(eye.shape[0], dtype=torch.double)).abs().max()) < 1e-5
            b = math.sqrt(6.0 / (mod.weight_ih.shape[0] + mod.weight_ih.shape[1]))
            assert float(mod.weight_ih.data.abs().max()) <= b + 1e-7
            assert float(mod.bias_ih.data.abs().max()) == 0.0 and float(mod.bias_hh.data.abs().max()) == 0.0
        if isinstance(mod, LayerNormP):
            assert torch.equal(mod.weight.data, torch.ones_like(mod.weight)) and float(mod.bias.data.abs().max()) == 0.0
    assert n_conv == 8 and n_lin >= 15
    # the actor-critic is NOT re-initialised: torch.nn.Linear defaults, so biases are non-zero and weights are bounded by
    # 1/sqrt(fan_in) (kaiming_uniform with a = sqrt(5))
    for head in (m.ac.actor, m.ac.critic, m.ac.critic_target):
        lin = head.model[0]
        bound = 1.0 / math.sqrt(lin.weight.shape[1])
        assert float(lin.weight.data.abs().max()) <= bound + 1e-7
        assert float(lin.bias.data.abs().max()) > 0.0 and float(lin.bias.data.abs().max()) <= bound + 1e-7
    assert not any(p.requires_grad for p in m.ac.critic_target.parameters())


@pytest.mark.skipif(not os.path.isdir('/root/reference/pydreamer'), reason='the reference exists in the build container only')
def test_reference_loads_build_state_dict():
    """Checkpoint interop, model side (tools.py:164-197, generator.py:109): the REAL reference Dreamer loads the state_dict
    written by this package (strict: identical keys and shapes) and, with those weights, reproduces the inference golden;
    and this package loads the reference's state_dict.  Runs only where /root/reference exists; nothing is read from it on
    the GPU box."""
    import ast
    import sys
    import numpy as np
    import torch
    import yaml
    from argparse import Namespace
    from oracle import dreamer_oracle as O
    from oracle.gen_golden import MultinomialPatch, reference_conf
    from pydreamer_amd.models import Dreamer
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'tiny_inference.npz'))
    oconf = O.make_conf(**dict(ast.literal_eval(str(g['conf_json']))))
    conf = config.load_config('defaults', 'atari', **{k: getattr(oconf, k) for k in vars(oconf)})
    ours = Dreamer(conf)                                     # parameters on CPU: holding and exchanging them needs no GPU
    ours.load_state_dict(O.make_params(oconf, seed=0), strict=True)
    sys.path.insert(0, '/root/reference')
    from pydreamer.models import Dreamer as RefDreamer
    import torch.distributions as D
    D.Distribution.set_default_validate_args(False)
    t = O.tiny_conf()
    rconf = reference_conf(['defaults', 'atari'], dict(deter_dim=t.deter_dim, hidden_dim=t.hidden_dim, stoch_dim=t.stoch_dim,
                                                       stoch_discrete=t.stoch_discrete, cnn_depth=t.cnn_depth,
                                                       action_dim=t.action_dim))
    ref = RefDreamer(rconf)
    ref.load_state_dict(ours.state_dict(), strict=True)      # the reference accepts the build's checkpoint as is
    ours.load_state_dict(ref.state_dict(), strict=True)      # ... and the other way round
    u8 = torch.from_numpy(g['in_image_u8'])
    obs = dict(image=(u8.float() / 255.0 - 0.5).permute(0, 1, 4, 2, 3).contiguous(), action=torch.from_numpy(g['in_action']),
               reset=torch.from_numpy(g['in_reset']), reward=torch.zeros(1, 3), terminal=torch.zeros(1, 3))
    with MultinomialPatch() as mp, torch.no_grad():
        mp.queue = [torch.from_numpy(g['in_u'])[0]]
        dist, (h1, z1), metrics = ref.inference(obs, (torch.from_numpy(g['in_h']), torch.from_numpy(g['in_z'])))
    np.testing.assert_allclose(dist.probs.numpy(), g['action_probs'], rtol=0, atol=1e-7)
    np.testing.assert_allclose(h1.numpy(), g['out_h'], rtol=0, atol=1e-7)


def test_graft_entry_build_passes():
    """The driver's build check (make is a no-op on an up-to-date tree; the ABI assertion and symbol walk are what is tested)."""
    import __graft_entry__ as g
    g.build()


def test_runtime_switch_defaults():
    """Library-wide switches that change WHICH kernels run (never what they compute): the bf16-storage operand path is on, the
    LDS-weight-stationary persistent posterior chain is on (where the shape qualifies) unless the environment says otherwise."""
    lib = hip.lib()
    if not any(os.environ.get(k) for k in ('DM_BF16_NO_TWINS', 'DM_RSSM_LDS')):
        assert lib.dm_bf16_twins_enable(-1) == 1
        assert lib.dm_rssm_lds_enable(-1) == 1
        assert lib.dm_bptt_fold_enable(-1) == 1 or os.environ.get('DM_BPTT_FOLD')
    assert lib.dm_rssm_lds_status() == 0 and lib.dm_rssm_lds_status_ack() == 0 and lib.dm_rssm_lds_gave_up() == 0
    assert lib.dm_rccl_available() in (0, 1)          # the native exchange step binds RCCL with dlopen: no link-time dependency
    if lib.dm_rccl_available():
        assert lib.dm_rccl_version() > 20000
    assert lib.dm_bf16_twins_enable(0) == 0 and lib.dm_bf16_twins_enable(1) == 1
    if not os.environ.get('DM_ROLLOUT_NO_FUSE_ACT'):      # round 6: the rollout's action draw rides in the actor kernel by default
        assert lib.dm_rollout_fuse_act_enable(-1) == 1 and lib.dm_rollout_fuse_act_enable(0) == 0 and lib.dm_rollout_fuse_act_enable(1) == 1
    assert lib.dm_gemm_dma_enable(-1) == 1 and lib.dm_gemm_dma_enable(0) == 0 and lib.dm_gemm_dma_enable(1) == 1
    assert lib.dm_dec_l4_bwd_direct_enable(-1) == 1 and lib.dm_dec_l4_bwd_direct_enable(0) == 0 and lib.dm_dec_l4_bwd_direct_enable(1) == 1


def test_scheduling_switches_and_windows():
    """Host-side scheduling decisions that never change what is computed: parameter gradients on the library's side stream and
    the early head window are on by default; the head windows are the same in every execution order (ActorCritic.split_steps);
    joining / disarming the side stream with nothing deferred is a no-op that needs no device; rollout marks are validated."""
    import ctypes
    from pydreamer_amd import models as M
    if not any(os.environ.get(k) for k in ('DM_WGRAD_SIDE', 'DM_HEADS_EARLY')):
        assert M._WGRAD_SIDE and M._HEADS_EARLY
    S = M.ActorCritic.split_steps
    assert [S(j) for j in (2, 6, 8, 9, 10, 16, 50)] == [2, 6, 8, 5, 6, 10, 31]      # J < 9: one window; else 5/8 of the steps first
    assert all(0 < S(j) <= j for j in range(1, 200))
    lib = hip.lib()
    assert lib.dm_wgrad_side_arm(0) == 0
    assert lib.dm_wgrad_side_join(None) == 0
    assert lib.dm_dream_rollout_marks(0, None, None) == 0
    assert lib.dm_dream_rollout_marks(5, (ctypes.c_int * 5)(), (ctypes.c_void_p * 5)()) != 0       # at most four marks
    assert b'marks' in lib.dm_last_error()
    assert lib.dm_dream_rollout_marks(1, (ctypes.c_int * 1)(3), (ctypes.c_void_p * 1)(None)) != 0   # null event
