import sys, torch
sys.path.insert(0, '/root/repo')
from pydreamer_amd import hip
for rows, n in ((40000, 400), (40000, 1624), (2500, 1800), (37500, 400)):
    x = torch.randn(rows, n, device='cuda'); out = torch.empty(n, device='cuda'); ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
    f = lambda: hip.call('dm_colsum', rows, n, hip.fptr(x), n, hip.fptr(out), hip.ptr(ws), ws.numel(), hip.stream())
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): f()
    e1.record(); torch.cuda.synchronize()
    print(rows, n, 'us per colsum (partial + final):', round(e0.elapsed_time(e1) * 10, 2), 'max err', float((out.double() - x.double().sum(0)).abs().max()))
