"""GPU microbenchmark (not a test): what a CU-masked stream (hip.cu_masked_stream) really gets - one 4096^3 product and
one 50-row chain product timed on streams with different masks, alone and with a second masked stream busy beside it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pydreamer_amd import hip


def timed(stream, fn, reps=10):
    with torch.cuda.stream(stream):
        fn(); fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            fn()
        e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device('cuda:0')
    ws = {}
    def product(M, N, K):
        A, B, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev), torch.empty(M, N, device=dev)
        def run():
            s = torch.cuda.current_stream().cuda_stream
            w = ws.setdefault(s, torch.empty(256 << 20, dtype=torch.uint8, device=dev))
            hip.call('dm_gemm_f32', 0, 0, M, N, K, hip.fptr(A), K, hip.fptr(B), K, hip.fptr(C), N, None, None, 0, 0,
                     hip.ptr(w), w.numel(), hip.stream())
        return run
    big, small = product(4096, 4096, 4096), product(50, 1800, 1000)
    torch.cuda.synchronize()
    masks = {'torch stream': None, 'all 256': [0xFFFFFFFF] * 8, 'bits 8-31 of each word (224)': [0xFFFFFF00] * 8,
             'bits 0-7 of each word (64)': [0xFF] * 8, 'words 0-3 (128)': [0xFFFFFFFF] * 4 + [0] * 4,
             'word 0 (32)': [0xFFFFFFFF] + [0] * 7, '2 words only, all ones (64 bits)': [0xFFFFFFFF] * 2}
    streams = {}
    for name, m in masks.items():
        st = torch.cuda.Stream(dev) if m is None else hip.cu_masked_stream(m, dev)
        streams[name] = st
        print(f'{name:36s} 4096^3 {timed(st, big):9.1f} us   50x1800x1000 {timed(st, small, 50):7.1f} us', flush=True)
    # the chain product on its reserved CUs while the complement stream runs big products
    a, b = streams['bits 0-7 of each word (64)'], streams['bits 8-31 of each word (224)']
    for name, (sa, sb) in {'masked 64 | masked 224': (a, b), 'torch prio -1 | torch': (torch.cuda.Stream(dev, priority=-1), torch.cuda.Stream(dev))}.items():
        with torch.cuda.stream(sb):
            for _ in range(40):
                big()
        t = timed(sa, small, 100)
        torch.cuda.synchronize()
        print(f'{name:36s} chain product beside big products: {t:7.1f} us', flush=True)


main()
