"""GPU microbenchmark (not a test): dm_gemm_bf16h (operands STORED as bf16, 64-k tiles) against dm_gemm_f32 with DM_GEMM_BF16
(fp32 storage, rounded on the way into LDS) on the step's shapes.  One line per shape; DM_GEMM_TILE=1..5 forces a tile."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd import hip

SHAPES = [
    (0, 0, 4096, 4096, 4096, 'square NT'), (0, 1, 4096, 4096, 4096, 'square NN'), (1, 1, 4096, 4096, 4096, 'square TN'),
    (0, 0, 2500, 1800, 1000, 'dream gru ih'), (0, 0, 2500, 1800, 600, 'dream gru hh'), (0, 0, 2500, 1024, 1000, 'dream prior'),
    (0, 0, 2500, 1000, 600, 'dream prior_h'), (0, 0, 40000, 400, 1624, 'AC head l0'), (0, 0, 40000, 400, 400, 'AC head l1-3'),
    (0, 1, 2500, 1000, 1800, 'dgrad gru ih'), (1, 1, 1800, 1000, 2500, 'wgrad gru ih'), (1, 1, 400, 1624, 40000, 'wgrad head l0'),
    (0, 0, 490000, 96, 768, 'enc L2 (as plain)'), (0, 0, 10000, 384, 3072, 'enc L4 (as plain)'), (1, 1, 96, 768, 490000, 'enc L2 dW (as plain)'),
]


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ws = torch.empty(512 << 20, dtype=torch.uint8, device='cuda')
    g = torch.Generator(device='cuda').manual_seed(0)
    for al, bl, M, N, K, what in SHAPES:
        A = torch.randn((M, K) if al == 0 else (K, M), device='cuda', generator=g)
        B = torch.randn((N, K) if bl == 0 else (K, N), device='cuda', generator=g)
        Ah, Bh = A.bfloat16().contiguous(), B.bfloat16().contiguous()
        C = torch.empty(M, N, device='cuda')
        Ch = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
        C2 = torch.empty(M, N, device='cuda')
        def run_h():
            hip.call('dm_gemm_bf16h', al, bl, M, N, K, hip.ptr(Ah), Ah.shape[1], hip.ptr(Bh), Bh.shape[1], hip.fptr(C), N,
                     hip.ptr(Ch), None, 0, hip.ptr(ws), ws.numel(), hip.stream())
        def run_f():
            hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(A), A.shape[1], hip.fptr(B), B.shape[1], hip.fptr(C2), N,
                     None, None, 0, hip.DM_GEMM_BF16, hip.ptr(ws), ws.numel(), hip.stream())
        ms_h, ms_f = timeit(run_h), timeit(run_f)
        err = twin = None
        if M * N <= 64e6:
            R = (Ah.float() if al == 0 else Ah.float().t()) @ (Bh.float() if bl == 0 else Bh.float().t()).t()
            err = float((C - R).norm() / R.norm())
            twin = bool(torch.equal(Ch, C.bfloat16()))
            err_f = float((C2 - R).norm() / R.norm())
        print(json.dumps(dict(what=what, al=al, bl=bl, M=M, N=N, K=K, us_h=round(ms_h * 1e3, 1), us_f32store=round(ms_f * 1e3, 1),
                              tf_h=round(2.0 * M * N * K / ms_h / 1e9, 1), tf_f32store=round(2.0 * M * N * K / ms_f / 1e9, 1),
                              rel_err_vs_fp32_of_bf16_inputs=err, twin_exact=twin)), flush=True)
        del A, B, Ah, Bh, C, Ch, C2


if __name__ == '__main__':
    main()
