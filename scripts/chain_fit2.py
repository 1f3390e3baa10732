import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd.models import MLP
def t(rows, in_dim, layers, save, out_dim=18, reps=30):
    m = MLP(in_dim, out_dim, 400, layers).to('cuda')
    x = torch.randn(rows, in_dim, device='cuda')
    ws = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    for _ in range(3): m.fwd(x, in_dim, rows, ws, save_acts=save)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): m.fwd(x, in_dim, rows, ws, save_acts=save)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for rows in (2500,):
    for in_dim, layers in ((600, 4), (1624, 4), (400, 4)):
        for save in (False, True):
            print(f'rows {rows} in_dim {in_dim} layers {layers} save_acts {save}: {t(rows, in_dim, layers, save):8.1f} us', flush=True)
