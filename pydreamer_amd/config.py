"""Config surface of the hot path: the keys of pydreamer's `config/defaults.yaml` that Dreamer / WorldModel /
ActorCritic read (reference: dreamer.py:23-58,237-277; encoders.py:14-36; decoders.py:14-46; launch.py:16-41).

`load_config('defaults', 'atari', batch_size=50)` mirrors `launch.py --configs defaults atari --batch_size 50`:
named sections are merged in order, then overrides, giving an `argparse.Namespace` the modules read by attribute.
A user-supplied YAML with the same flat section->key layout can be merged with `yaml_path=`.
"""
from argparse import Namespace

SECTIONS = {
    'defaults': dict(
        # features
        image_key='image', image_size=64, image_channels=3, image_categorical=False, action_dim=0, clip_rewards=None,
        # training
        reset_interval=200, iwae_samples=1, kl_balance=0.8, kl_weight=1.0, image_weight=1.0, vecobs_weight=1.0,
        reward_weight=1.0, terminal_weight=1.0, adam_lr=3.0e-4, adam_lr_actor=1.0e-4, adam_lr_critic=1.0e-4,
        adam_eps=1.0e-5, keep_state=True, batch_length=48, batch_size=32, device='cuda:0', grad_clip=200,
        grad_clip_ac=200, image_decoder_min_prob=0, amp=False, probe_gradients=False,
        # model
        model='dreamer', deter_dim=2048, stoch_dim=32, stoch_discrete=32, hidden_dim=1000, gru_layers=1, gru_type='gru',
        layer_norm=True, vecobs_size=0, image_encoder='cnn', cnn_depth=48, image_encoder_layers=0, image_decoder='cnn',
        image_decoder_layers=0, reward_input=False, reward_decoder_layers=4, reward_decoder_categorical=None,
        terminal_decoder_layers=4,
        # probe
        probe_model='none',
        # actor critic
        gamma=0.995, lambda_gae=0.95, entropy=0.003, target_interval=100, imag_horizon=15, actor_grad='reinforce',
        actor_dist='onehot',
        # aux critic
        aux_critic=False, aux_critic_weight=1.0, gamma_aux=0.99, lambda_gae_aux=0.95, target_interval_aux=1000,
    ),
    'atari': dict(action_dim=18, clip_rewards='tanh', deter_dim=1024, kl_weight=0.1, gamma=0.99, entropy=0.001),
    'dmc': dict(action_dim=12, entropy=1.0e-4, actor_grad='dynamics', actor_dist='tanh_normal', clip_rewards='tanh'),
    'debug': dict(device='cpu', batch_length=15, batch_size=5, imag_horizon=3),
}


def load_config(*sections, yaml_path=None, **overrides):
    tables = {k: dict(v) for k, v in SECTIONS.items()}
    if yaml_path is not None:
        import yaml
        with open(yaml_path) as f:
            for name, table in (yaml.safe_load(f) or {}).items():
                tables.setdefault(name, {}).update(table or {})
    conf = {}
    for s in sections or ('defaults',):
        if s not in tables:
            raise KeyError(f'unknown config section {s!r}; known: {sorted(tables)}')
        conf.update(tables[s])
    unknown = [k for k in overrides if k not in conf]
    if unknown:
        raise KeyError(f'unknown config keys {unknown}')
    conf.update(overrides)
    return Namespace(**conf)


def atari_literal(**overrides):
    """BASELINE.json configs[1]: Atari defaults at B=50, T=50, H=15, deter=600, stoch 32x32."""
    base = dict(batch_size=50, batch_length=50, deter_dim=600, action_dim=18)
    base.update(overrides)
    return load_config('defaults', 'atari', **base)
