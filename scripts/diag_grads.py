"""GPU diagnostic (not a test): per-parameter gradient error of the HIP path vs the oracle for several small configs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import dreamer_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import test_gpu_training_step as T

cases = [
    dict(name='tiny seed0 free', kw={}, seed=0, forced=False),
    dict(name='tiny seed1 free', kw={}, seed=1, forced=False),
    dict(name='tiny seed1 forced', kw={}, seed=1, forced=True),
    dict(name='B4 T6 seed0 free', kw=dict(batch_size=4, batch_length=6), seed=0, forced=False),
    dict(name='B4 T6 seed1 forced klb0.5', kw=dict(batch_size=4, batch_length=6, kl_balance=0.5), seed=1, forced=True),
    dict(name='B4 T5 seed1 forced', kw=dict(batch_size=4), seed=1, forced=True),
    dict(name='B3 T6 seed1 forced', kw=dict(batch_length=6), seed=1, forced=True),
    dict(name='B8 T8 seed2 free', kw=dict(batch_size=8, batch_length=8), seed=2, forced=False),
]
for c in cases:
    oconf = O.tiny_conf(**c['kw'])
    r = T._run_pair(oconf, 1, forced=c['forced'], seed=c['seed'])[0]
    errs = sorted(((T._rel_l2(r['gh'][k], g), k, float(g.norm())) for k, g in r['go'].items()), reverse=True)
    print(f"== {c['name']}: losses hip {[round(float(x),6) for x in r['lh']]} oracle {[round(float(x),6) for x in r['lo']]}")
    for e, k, n in errs[:6]:
        print(f'   {e:.3e}  |g|={n:.3e}  {k}')
