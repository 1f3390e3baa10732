"""How the 64 x 64 LDS-DMA product's time depends on the NUMBER of tiles (K = 1000, NT): the quantisation / tail a stream-K or
tail-split scheduler could recover on the 2 500-row rollout products.  us per launch, TF/s, us per (tiles / 256)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd import hip

def bench(M, N, K, reps=30):
    A = torch.rand(M, K, device='cuda') * 2 - 1
    B = torch.rand(N, K, device='cuda') * 2 - 1
    C = torch.empty(M, N, device='cuda')
    ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')
    def run():
        hip.call('dm_gemm_f32', 0, 0, M, N, K, hip.fptr(A), K, hip.fptr(B), K, hip.fptr(C), N, None, None, 0, 0, hip.ptr(ws), ws.numel(), hip.stream())
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best

K = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
print(f'K = {K}, 64 x 64 tiles (DM_GEMM_TILE=3 forced by the caller)')
print('      M      N  tiles  per CU     us    TF/s  us per (tiles/256)')
for M, N in [(1024, 1024), (2048, 1024), (2560, 1024), (2500, 1024), (2500, 1000), (2048, 1536), (2560, 1280), (2048, 2048), (2304, 2048), (2560, 1792), (2500, 1800),
             (2560, 1856), (2560, 2048), (3072, 2048), (4096, 2048), (4096, 4096)]:
    t = ((M + 63) // 64) * ((N + 63) // 64)
    us = bench(M, N, K)
    print(f'{M:7d}{N:7d}{t:7d}{t / 256:8.2f}{us:8.1f}{2e-6 * M * N * K / us:8.1f}{us / (t / 256):10.2f}')
