"""GPU microbenchmark (not a test): where the whole-MLP forward kernel (csrc/mlp_chain.hip) spends its time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd.models import MLP
from scripts.mlp_bench import timeit

ws = torch.empty(1 << 30, dtype=torch.uint8, device='cuda')
for rows in (2500, 352, 16):
    for in_dim, layers, acts in ((1624, 4, True), (1624, 4, False), (16, 4, False), (1624, 1, False), (400, 1, False), (16, 1, False)):
        torch.manual_seed(0)
        m = MLP(in_dim, 18, 400, layers).to('cuda')
        x = torch.randn(rows, in_dim, device='cuda')
        a = torch.empty(m.acts_floats(rows), device='cuda') if acts else None
        t = timeit(lambda: m.fwd(x, in_dim, rows, ws, acts=a, save_acts=acts), reps=20)
        fl = 2.0 * rows * (in_dim * 400 + (layers - 1) * 160000 + 400 * 18)
        print(f'rows {rows:5d} in_dim {in_dim:5d} layers {layers} acts {int(acts)}: {t:7.1f} us  {fl / t / 1e6:6.1f} TF/s', flush=True)
