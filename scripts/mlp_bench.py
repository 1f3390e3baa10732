"""GPU microbenchmark (not a test): dm_mlp_head_fwd / dm_mlp_head_bwd on the step's head shapes.
Run twice to A/B the row-panel kernels (csrc/panel.hip) against GEMM + LayerNorm launches:
    python scripts/mlp_bench.py                              # panel path for rows >= 16384
    DM_PANEL_MIN_ROWS=1000000000 python scripts/mlp_bench.py # GEMM + LayerNorm launches everywhere"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd.models import MLP

CASES = [  # rows, in_dim, out_dim, acts, what
    (40000, 1624, 1, True, 'critic (J*M rows)'),
    (40000, 1624, 1, False, 'critic_target / dream reward, terminal (no acts)'),
    (37500, 1624, 18, True, 'actor over all H*M rows'),
    (2500, 1624, 1, True, 'wm reward / terminal head'),
    (2500, 1624, 18, True, 'actor, one rollout step'),
    (5600, 1624, 1, True, 'critic at the 7-column shard'),
]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def main():
    ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
    print('DM_PANEL_MIN_ROWS =', os.environ.get('DM_PANEL_MIN_ROWS', '(default 16384)'))
    for rows, in_dim, out_dim, acts, what in CASES:
        torch.manual_seed(0)
        m = MLP(in_dim, out_dim, 400, 4).to('cuda')
        m.precision = 1 if '--bf16' in sys.argv else 0
        x = torch.randn(rows, in_dim, device='cuda')
        dout = torch.randn(rows, out_dim, device='cuda')
        a = torch.empty(m.acts_floats(rows), device='cuda') if acts else None
        fwd_flop = 2.0 * rows * (in_dim * 400 + 3 * 400 * 400 + 400 * out_dim)
        t_f = timeit(lambda: m.fwd(x, in_dim, rows, ws, acts=a, save_acts=acts))
        line = f'{what:50s} rows {rows:6d} fwd {t_f:8.1f} us {fwd_flop / t_f / 1e6:6.1f} TF/s'
        if acts:
            t_b = timeit(lambda: m.bwd(x, in_dim, rows, a, dout, ws))
            bwd_flop = 2.0 * fwd_flop - 2.0 * rows * in_dim * 400       # no dx for the first layer
            line += f' | bwd {t_b:8.1f} us {bwd_flop / t_b / 1e6:6.1f} TF/s'
        print(line, flush=True)


if __name__ == '__main__':
    main()
