"""GPU probe (not a test): where the whole-MLP forward kernel's time goes - call time against the number of layers and the input
width at a fixed row count, through the C-ABI (MLP.fwd).  Slope over in_dim = time per 32-k pair of layer 0; slope over layers =
time per hidden layer (13 pairs + one LayerNorm epilogue); intercept = output layer + launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd.models import MLP

def t(rows, in_dim, layers, out_dim=18, reps=30, save=False):
    m = MLP(in_dim, out_dim, 400, layers).to('cuda')
    x = torch.randn(rows, in_dim, device='cuda')
    ws = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    for _ in range(3):
        m.fwd(x, in_dim, rows, ws, save_acts=save)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        m.fwd(x, in_dim, rows, ws, save_acts=save)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for rows in (16, 2500, 8000):
    print(f'rows {rows}:')
    for in_dim, layers in ((400, 1), (400, 2), (400, 3), (400, 4), (800, 1), (1600, 1), (3200, 1), (1600, 4), (600, 4)):
        print(f'  in_dim {in_dim:5d} layers {layers}: {t(rows, in_dim, layers):8.1f} us', flush=True)

for in_dim in (600, 1624, 400):      # the price of saving the activations for backward
    print(f'rows 2500 in_dim {in_dim} layers 4: save_acts False {t(2500, in_dim, 4):.1f} us, True {t(2500, in_dim, 4, save=True):.1f} us', flush=True)
