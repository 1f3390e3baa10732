"""GPU microbenchmark (not a test): dm_gemm_f32 on the step's shapes - time AND error against an fp64 product, for the
arithmetic mode the process runs in (default: fp32 MFMA; --bf16: bf16 operands; DM_GEMM_NO_PIPE=1: the single-stage loop instead of the software-pipelined kernel).  One JSON line per shape; profiles/r03_gemm_modes.txt keeps the
round-3 table (which also has the split-bf16 mode, removed in round 4)."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd import hip
from gemm_bench import SHAPES


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=8)
    ap.add_argument('--bf16', action='store_true')
    ap.add_argument('--only', default='')
    ap.add_argument('--dist', default='normal', choices=('normal', 'positive'))     # positive: no cancellation (worst case for a biased product)
    args = ap.parse_args()
    mode = 'bf16' if args.bf16 else 'native'
    gflags = hip.DM_GEMM_BF16 if args.bf16 else 0
    shapes = [SHAPES[int(i)] for i in args.only.split(',')] if args.only else SHAPES
    ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(0)
    for al, bl, M, N, K, what in shapes:
        shp_a, shp_b = ((M, K) if al == 0 else (K, M)), ((N, K) if bl == 0 else (K, N))
        A = torch.randn(shp_a, device='cuda', generator=g)
        B = torch.randn(shp_b, device='cuda', generator=g)
        if args.dist == 'positive':
            A.abs_(); B.abs_()
        C = torch.empty(M, N, device='cuda')
        def run():
            hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(A), A.shape[1], hip.fptr(B), B.shape[1], hip.fptr(C), N,
                     None, None, 0, gflags, hip.ptr(ws), ws.numel(), hip.stream())
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.reps
        err = None
        if M * N <= 64e6:          # fp64 reference (rocBLAS dgemm through torch): error of THIS mode's result
            A64 = (A if al == 0 else A.t()).double()
            B64 = (B if bl == 0 else B.t()).double()
            R = A64 @ B64.t()
            d = (C.double() - R)
            err = dict(rel_l2=float(d.norm() / R.norm()), max_abs=float(d.abs().max()), ref_rms=float(R.pow(2).mean().sqrt()),
                       mean_signed=float(d.mean() / R.abs().mean()))
            del A64, B64, R, d
        print(json.dumps(dict(mode=mode, dist=args.dist, what=what, al=al, bl=bl, M=M, N=N, K=K, us=ms * 1e3,
                              tflops=2.0 * M * N * K / ms / 1e9, err=err)), flush=True)
        del A, B, C


if __name__ == '__main__':
    main()
