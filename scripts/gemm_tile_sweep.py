"""GPU microbenchmark (not a test): every plain product of the step's shape table (profiles/r05_gemm_shapes_dma.txt) on every tile
of the menu (DM_GEMM_TILE=1..5 in a subprocess each: the switch is read once per process), LDS-DMA loop on.  Prints us per shape
and tile and the library's own choice - the data the tile cost model in gemm.hip is checked against."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def shapes(path):
    out = []
    for line in open(path).read().split('\n')[1:]:
        f = line.split()
        if len(f) < 10 or f[1] == '0' or f[5] != '0':
            continue
        kind, M, N, K = int(f[0]), int(f[1]), int(f[2]), int(f[3])
        if K % 4 or min(M, N) < 4:
            continue
        out.append(((kind >> 1) & 1, kind & 1, M, N, K, float(f[6])))
    return sorted(set(out))


def child(path):
    import torch
    from pydreamer_amd import hip
    ws = torch.empty(256 << 20, dtype=torch.uint8, device='cuda')
    for al, bl, M, N, K, n in shapes(path):
        A = torch.randn((M, K) if al == 0 else (K, M), device='cuda')
        B = torch.randn((N, K) if bl == 0 else (K, N), device='cuda')
        C = torch.empty(M, N, device='cuda')
        run = lambda: hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(A), A.shape[1], hip.fptr(B), B.shape[1], hip.fptr(C), N,
                               None, None, 0, 0, hip.ptr(ws), ws.numel(), hip.stream())
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        print(al, bl, M, N, K, n, e0.elapsed_time(e1) / 20 * 1e3, flush=True)


if __name__ == '__main__':
    path = os.path.join(ROOT, 'profiles', 'r05_gemm_shapes_dma.txt')
    if len(sys.argv) > 1 and sys.argv[1] == 'child':
        child(path)
        sys.exit(0)
    res = {}
    for tile in (0, 1, 2, 3, 4, 5):
        env = dict(os.environ)
        if tile:
            env['DM_GEMM_TILE'] = str(tile)
        out = subprocess.run([sys.executable, __file__, 'child'], env=env, capture_output=True, text=True).stdout
        for line in out.split('\n'):
            f = line.split()
            if len(f) == 7:
                res.setdefault(tuple(f[:6]), {})[tile] = float(f[6])
    names = ['auto', '128x128', '128x64', '64x64', '128x96', '96x128']
    print('al bl M N K n/step ' + ' '.join(names) + ' best regret_us_per_step')
    regret = 0.0
    for k, v in sorted(res.items(), key=lambda kv: -float(kv[0][5]) * kv[1].get(0, 0)):
        best = min((v[t], t) for t in v if t)
        r = float(k[5]) * (v.get(0, 0) - best[0])
        regret += r
        print(' '.join(k), ' '.join(f'{v.get(t, float("nan")):8.1f}' for t in range(6)), names[best[1]], f'{r:7.1f}')
    print('total regret us/step', regret)
