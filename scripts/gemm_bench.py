"""GPU microbenchmark (not a test): dm_gemm_f32 TFLOP/s on the step's GEMM shapes and on a 4096^3 reference shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd import hip

SHAPES = [
    # (al, bl, M, N, K, what)
    (0, 0, 4096, 4096, 4096, 'reference square NT'),
    (0, 1, 4096, 4096, 4096, 'reference square NN'),
    (1, 1, 4096, 4096, 4096, 'reference square TN'),
    (0, 0, 40000, 400, 1624, 'AC head layer0 fwd'),
    (0, 0, 40000, 400, 400, 'AC head layer1-3 fwd'),
    (0, 1, 40000, 400, 400, 'AC head dX'),
    (1, 1, 400, 1624, 40000, 'AC head dW0'),
    (1, 1, 400, 400, 40000, 'AC head dW1-3'),
    (0, 0, 2500, 1000, 1024, 'dream z_mlp'),
    (0, 0, 2500, 1800, 1000, 'dream gru ih'),
    (0, 0, 2500, 1800, 600, 'dream gru hh'),
    (0, 0, 2500, 400, 1624, 'dream actor l0 / wm heads'),
    (0, 0, 50, 1000, 1024, 'rssm z_mlp step'),
    (0, 0, 50, 1800, 1000, 'rssm gru ih step'),
    (0, 1, 50, 1000, 1800, 'rssm bptt dgi@Wih'),
    (0, 0, 2402500, 48, 48, 'enc L1'),
    (0, 0, 490000, 96, 768, 'enc L2'),
    (0, 0, 90000, 192, 1536, 'enc L3'),
    (0, 0, 10000, 384, 3072, 'enc L4'),
    (1, 1, 96, 768, 490000, 'enc L2 dW'),
    (1, 1, 48, 48, 2402500, 'enc L1 dW'),
    (0, 1, 490000, 768, 96, 'enc L2 dXcol'),
    (0, 1, 2500, 4800, 1536, 'dec L1 Ycol'),
    (0, 1, 62500, 2400, 192, 'dec L2 Ycol'),
    (0, 1, 422500, 1728, 96, 'dec L3 Ycol'),
    (0, 1, 2250000, 108, 48, 'dec L4 Ycol'),
    (0, 0, 422500, 96, 1728, 'dec L3 dX'),
    (1, 1, 96, 1728, 422500, 'dec L3 dW'),
    (1, 1, 48, 144, 2250000, 'dec L4 dW (cout padded to 4)'),
    # data-parallel shard (B = 7 of 50 columns): imagination products at M = T*B/8 = 350 rows
    (0, 0, 350, 1000, 1024, 'shard dream z_mlp'),
    (0, 0, 350, 1800, 1000, 'shard dream gru ih'),
    (0, 0, 350, 400, 1624, 'shard actor l0'),
    (0, 0, 350, 400, 400, 'shard actor l1-3'),
]

def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default='')       # comma-separated indices into SHAPES
    ap.add_argument('--reps', type=int, default=10)
    ap.add_argument('--bf16', action='store_true')       # bf16 operands / fp32 accumulate (conf.amp)
    args = ap.parse_args()
    gflags = hip.DM_GEMM_BF16 if args.bf16 else 0
    shapes = [SHAPES[int(i)] for i in args.only.split(',')] if args.only else SHAPES
    ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
    for al, bl, M, N, K, what in shapes:
        A = torch.randn((M, K) if al == 0 else (K, M), device='cuda')
        B = torch.randn((N, K) if bl == 0 else (K, N), device='cuda')
        C = torch.empty(M, N, device='cuda')
        def run():
            hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(A), A.shape[1], hip.fptr(B), B.shape[1], hip.fptr(C), N,
                     None, None, 0, gflags, hip.ptr(ws), ws.numel(), hip.stream())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        reps = args.reps
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f'{what:28s} al={al} bl={bl} {M:8d}x{N:5d}x{K:8d}  {ms*1e3:9.1f} us  {2.0*M*N*K/ms/1e9:7.1f} TF/s', flush=True)
        del A, B, C

if __name__ == '__main__':
    main()
