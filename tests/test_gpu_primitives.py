"""-m gpu: every C-ABI primitive against a plain torch fp32/fp64 restatement of the same op.

Tolerances are stated per test; integer outputs (sampled indices) are compared exactly except where the
uniform lies within 1e-5 of a CDF boundary (softmax is not bit-reproducible across exp implementations).
"""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = 'cuda'


def _ws(nbytes=256 << 20):
    return torch.empty(nbytes, dtype=torch.uint8, device=DEV)


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _close(a, b, rtol, atol, what=''):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = err > tol
    assert not bad.any(), f'{what}: {int(bad.sum())}/{bad.numel()} mismatches, max err {float(err.max()):.3e}, ' \
                          f'max ref {float(b.abs().max()):.3e}'


# ------------------------------------------------------------------------------------------- GEMM
GEMM_CASES = [
    # (al, bl, M, N, K)
    (0, 0, 50, 1000, 1024), (0, 0, 50, 1800, 600), (0, 0, 2500, 400, 1624), (0, 0, 4097, 130, 70),
    (0, 0, 7, 1, 400), (0, 0, 33, 18, 18), (0, 0, 3000, 1536, 96),
    (0, 1, 50, 1024, 1000), (0, 1, 2500, 1624, 400), (0, 1, 129, 67, 33), (0, 1, 40000, 400, 1),
    (1, 1, 400, 1624, 2500), (1, 1, 48, 48, 30000), (1, 1, 1000, 18, 250), (1, 1, 65, 131, 259), (1, 1, 1, 400, 5000),
    (1, 0, 100, 90, 80),
]


@pytest.mark.parametrize('al,bl,M,N,K', GEMM_CASES)
def test_gemm_layouts(hip, al, bl, M, N, K):
    """C = A(m,k) B(n,k) for every operand layout, ragged sizes, split-K; asymmetric random operands.
    Tolerance: fp32 fmaf-chain vs fp64 reference, 2e-6*sqrt(K)*|a||b| scale."""
    A = _rand(M, K, seed=1)
    B = _rand(N, K, seed=2)
    Ad = A if al == 0 else A.t().contiguous()
    Bd = B if bl == 0 else B.t().contiguous()
    C = torch.full((M, N), float('nan'), device=DEV)
    ws = _ws()
    hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(Ad), Ad.shape[1], hip.fptr(Bd), Bd.shape[1], hip.fptr(C), N,
             None, None, 0, 0, hip.ptr(ws), ws.numel(), hip.stream())
    ref = A.double() @ B.double().t()
    _close(C, ref, 0, 3e-6 * np.sqrt(K) * 4, f'gemm {al}{bl} {M}x{N}x{K}')


@pytest.mark.parametrize('al,bl,M,N,K', [(0, 0, 300, 200, 160), (0, 1, 257, 96, 100), (1, 0, 132, 131, 64),
                                          (1, 1, 96, 260, 4000), (0, 0, 2500, 400, 1624), (1, 1, 400, 400, 40000)])
def test_gemm_bf16_operands(hip, al, bl, M, N, K):
    """DM_GEMM_BF16 (a per-call flag): operands rounded to bf16 (RNE), fp32 accumulation.  Reference = the same product of the
    bf16-ROUNDED operands in fp64, so the only difference left is fp32 summation order: same tolerance as the fp32 test.
    Every operand layout (the row-contiguous ones are transposed into the [row][k] LDS image), ragged edges, split-K."""
    A = _rand(M, K, seed=21)
    B = _rand(N, K, seed=22)
    Ad = A if al == 0 else A.t().contiguous()
    Bd = B if bl == 0 else B.t().contiguous()
    C = torch.full((M, N), float('nan'), device=DEV)
    ws = _ws()
    hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(Ad), Ad.shape[1], hip.fptr(Bd), Bd.shape[1], hip.fptr(C), N,
             None, None, 0, hip.DM_GEMM_BF16, hip.ptr(ws), ws.numel(), hip.stream())
    C32 = torch.empty_like(C)      # the next call, without the flag, is an fp32 product again: nothing sticks
    hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(Ad), Ad.shape[1], hip.fptr(Bd), Bd.shape[1], hip.fptr(C32), N,
             None, None, 0, 0, hip.ptr(ws), ws.numel(), hip.stream())
    torch.cuda.synchronize()
    _close(C32, A.double() @ B.double().t(), 0, 3e-6 * np.sqrt(K) * 4, f'fp32 gemm after a bf16 one {al}{bl} {M}x{N}x{K}')
    ref = A.bfloat16().double() @ B.bfloat16().double().t()
    _close(C, ref, 0, 3e-6 * np.sqrt(K) * 4, f'bf16 gemm {al}{bl} {M}x{N}x{K}')
    # and it really is a different (coarser) product than fp32
    full = A.double() @ B.double().t()
    assert (C.double().cpu() - full.cpu()).abs().max() > 1e-4


@pytest.mark.parametrize('al,bl,M,N,K', [(0, 0, 300, 200, 136), (0, 0, 2500, 1800, 1000), (0, 1, 257, 96, 72), (1, 0, 128, 333, 200),
                                         (1, 1, 96, 768, 4999), (1, 1, 400, 1624, 2500), (0, 0, 70, 40, 8), (0, 0, 4096, 1024, 4096)])
def test_gemm_bf16_storage(hip, al, bl, M, N, K):
    """dm_gemm_bf16h: operands STORED as bf16 (64-k tiles, 16-byte chunks), fp32 accumulation; reference = fp64 product of the
    same bf16 values; the optional bf16 twin of the result is exactly RNE(C).  Ragged M / N / K, every layout pair."""
    g = torch.Generator().manual_seed(11)
    A = torch.randn((M, K) if al == 0 else (K, M), generator=g).bfloat16().to(DEV)
    B = torch.randn((N, K) if bl == 0 else (K, N), generator=g).bfloat16().to(DEV)
    bias = torch.randn(N, generator=g).to(DEV)
    C = torch.empty(M, N, device=DEV)
    Ch = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    ws = torch.empty(64 << 20, dtype=torch.uint8, device=DEV)
    hip.call('dm_gemm_bf16h', al, bl, M, N, K, hip.ptr(A), A.shape[1], hip.ptr(B), B.shape[1], hip.fptr(C), N, hip.ptr(Ch),
             hip.fptr(bias), 0, hip.ptr(ws), ws.numel(), hip.stream())
    ref = (A.double() if al == 0 else A.double().t()) @ (B.double() if bl == 0 else B.double().t()).t() + bias.double()
    _close(C, ref, 0, 3e-6 * np.sqrt(K) * 4, f'bf16-storage gemm {al}{bl} {M}x{N}x{K}')
    assert torch.equal(Ch, C.bfloat16())


DMA_CASES = [
    # (al, bl, M, N, K, flags): 16-byte-load shapes on every tile of the menu (64x64, 128x64, 128x128, 128x96, 96x128), ragged
    # edges in M / N / K, split-K weight-gradient shapes, a single k-tile, fewer k-tiles than LDS stages
    (0, 0, 2500, 1800, 1000, 0), (0, 0, 2500, 1000, 600, 2), (0, 0, 4096, 4096, 512, 0), (0, 0, 40000, 400, 400, 3),
    (0, 0, 3000, 192, 864, 0), (0, 0, 300, 200, 160, 0), (0, 0, 70, 40, 8, 0), (0, 0, 129, 67, 36, 1), (0, 0, 5, 3, 4, 0),
    (0, 1, 2500, 1624, 400, 0), (0, 1, 2500, 4800, 1536, 0), (0, 1, 4096, 2048, 300, 2), (0, 1, 257, 96, 100, 0), (0, 1, 131, 8, 33, 0),
    (1, 1, 400, 1624, 2500, 0), (1, 1, 96, 1728, 42250, 0), (1, 1, 1536, 4800, 2500, 0), (1, 1, 48, 48, 30000, 0), (1, 1, 64, 132, 259, 1),
    (1, 1, 4, 400, 5000, 0), (1, 0, 100, 92, 80, 0), (1, 0, 2048, 1024, 777 * 4, 0),
]


@pytest.mark.parametrize('al,bl,M,N,K,flags', DMA_CASES)
def test_gemm_dma_equals_register_staged_loop(hip, al, bl, M, N, K, flags):
    """gemm_dma_kernel (LDS-DMA operand pipeline, round 5) against gemm_f32_kernel (register-staged loop) on the same call:
    same tiles, same k -> MFMA-step map, same epilogue => BIT-IDENTICAL results (torch.equal), with and without the bias /
    addend / accumulate / ELU epilogue; and both against the fp64 product."""
    A = _rand(M, K, seed=31)
    B = _rand(N, K, seed=32)
    Ad = A if al == 0 else A.t().contiguous()
    Bd = B if bl == 0 else B.t().contiguous()
    bias = _rand(N, seed=33) if flags else None
    add = _rand(M, N, seed=34) if flags & 1 else None
    C0 = _rand(M, N, seed=35)
    ws = _ws()
    outs = []
    try:
        for on in (2, 0):      # 2: the LDS-DMA loop for every k extent (by default it serves products of >= 14 k-tiles)
            assert hip.lib().dm_gemm_dma_enable(on) == on
            C = C0.clone()
            hip.call('dm_gemm_f32', al, bl, M, N, K, hip.fptr(Ad), Ad.shape[1], hip.fptr(Bd), Bd.shape[1], hip.fptr(C), N,
                     hip.fptr(bias) if bias is not None else None, hip.fptr(add) if add is not None else None, N, flags,
                     hip.ptr(ws), ws.numel(), hip.stream())
            torch.cuda.synchronize()
            outs.append(C)
    finally:
        hip.lib().dm_gemm_dma_enable(1)
    assert torch.equal(outs[0], outs[1]), f'{int((outs[0] != outs[1]).sum())} of {outs[0].numel()} elements differ'
    ref = A.double() @ B.double().t()
    if bias is not None:
        ref = ref + bias.double()
    if add is not None:
        ref = ref + add.double()
    if flags & 1:
        ref = ref + C0.double()
    if flags & 2:
        ref = F.elu(ref)
    _close(outs[0], ref, 2e-6, 3e-6 * np.sqrt(K) * 4 + 1e-5, f'dma gemm {al}{bl} {M}x{N}x{K}')


def test_gemm_epilogue_and_strides(hip):
    """bias + addend + accumulate + ELU, with sub-matrix leading dimensions (feature-matrix slices)."""
    M, N, K, ldc, lda = 70, 200, 100, 264, 164
    Abig = _rand(M, lda, seed=3)
    A = Abig[:, 60:60 + K]                      # offset view: base aligned (60*4 bytes = 240 = 16*15)
    B = _rand(N, K, seed=4)
    bias = _rand(N, seed=5)
    add = _rand(M, N + 8, seed=6)
    Cbig = _rand(M, ldc, seed=7)
    C0 = Cbig.clone()
    ws = _ws()
    a_ptr = ctypes.c_void_p(Abig.data_ptr() + 60 * 4)
    c_ptr = ctypes.c_void_p(Cbig.data_ptr() + 64 * 4)
    hip.call('dm_gemm_f32', 0, 0, M, N, K, a_ptr, lda, hip.fptr(B), K, c_ptr, ldc, hip.fptr(bias), hip.fptr(add), N + 8,
             hip.DM_GEMM_ACCUM | hip.DM_GEMM_ELU, hip.ptr(ws), ws.numel(), hip.stream())
    ref = F.elu(A.double() @ B.double().t() + bias.double() + add[:, :N].double() + C0[:, 64:64 + N].double())
    _close(Cbig[:, 64:64 + N], ref, 1e-5, 1e-4, 'gemm epilogue')
    # untouched columns stay untouched
    assert torch.equal(Cbig[:, :64], C0[:, :64]) and torch.equal(Cbig[:, 64 + N:], C0[:, 64 + N:])


def test_gemm_unaligned_k(hip):
    """K=18 (action_dim) rows are not 16-byte aligned: scalar load path."""
    M, N, K = 50, 1000, 18
    A, B = _rand(M, K, seed=8), _rand(N, K, seed=9)
    C = torch.empty(M, N, device=DEV)
    ws = _ws()
    hip.call('dm_gemm_f32', 0, 0, M, N, K, hip.fptr(A), K, hip.fptr(B), K, hip.fptr(C), N, None, None, 0, 0, hip.ptr(ws),
             ws.numel(), hip.stream())
    _close(C, A.double() @ B.double().t(), 0, 2e-5, 'gemm K=18')


def test_gemm_onehot_rows_exact(hip):
    """One-hot A rows (straight-through samples): result must equal the ordered sum of the selected weight columns
    bit-for-bit (fmaf chain with exact zeros), SURVEY section 0.8."""
    M, S, C, N = 50, 32, 32, 1000
    g = torch.Generator().manual_seed(10)
    idx = torch.randint(0, C, (M, S), generator=g)
    A = F.one_hot(idx, C).float().reshape(M, S * C).to(DEV)
    W = _rand(N, S * C, seed=11)
    out = torch.empty(M, N, device=DEV)
    hip.call('dm_gemm_f32', 0, 0, M, N, S * C, hip.fptr(A), S * C, hip.fptr(W), S * C, hip.fptr(out), N, None, None, 0, 0,
             None, 0, hip.stream())
    cols = (idx + torch.arange(S) * C).to(DEV)          # (M,S) selected columns, ascending in k
    ref = torch.zeros(M, N, device=DEV)
    for s in range(S):
        ref = ref + W[:, cols[:, s]].t()
    _close(out, ref, 0, 1e-5, 'one-hot gemm')


# ------------------------------------------------------------------------------------------- LayerNorm+ELU
@pytest.mark.parametrize('rows,n', [(50, 1000), (2500, 400), (3, 64), (130, 1001), (10, 1500), (7, 1024)])
def test_ln_elu_fwd_bwd(hip, rows, n):
    x = _rand(rows, n, seed=1, scale=2.0)
    gamma = 1 + 0.1 * _rand(n, seed=2)
    beta = 0.1 * _rand(n, seed=3)
    dy = _rand(rows, n, seed=4)
    y = torch.empty_like(x)
    stats = torch.empty(rows, 2, device=DEV)
    hip.call('dm_ln_elu_fwd', rows, n, hip.fptr(x), n, hip.fptr(gamma), hip.fptr(beta), 1e-3, hip.fptr(y), n,
             hip.fptr(stats), hip.stream())
    xd = x.double().requires_grad_(True)
    gd = gamma.double().requires_grad_(True)
    bd = beta.double().requires_grad_(True)
    yr = F.elu(F.layer_norm(xd, (n,), gd, bd, 1e-3))
    _close(y, yr, 1e-5, 1e-5, 'ln_elu fwd')
    yr.backward(dy.double())
    dx = torch.empty_like(x)
    dg = torch.empty(n, device=DEV)
    db = torch.empty(n, device=DEV)
    ws = _ws(64 << 20)
    hip.call('dm_ln_elu_bwd', rows, n, hip.fptr(x), n, hip.fptr(y), n, hip.fptr(stats), hip.fptr(gamma), hip.fptr(dy), n,
             hip.fptr(dx), n, hip.fptr(dg), hip.fptr(db), hip.ptr(ws), ws.numel(), hip.stream())
    _close(dx, xd.grad, 1e-4, 2e-5, 'ln_elu dx')
    _close(dg, gd.grad, 1e-4, 1e-4 * np.sqrt(rows), 'ln_elu dgamma')
    _close(db, bd.grad, 1e-4, 1e-4 * np.sqrt(rows), 'ln_elu dbeta')


@pytest.mark.parametrize('rows,n', [(2500, 1800), (200000, 48), (1000003, 3), (40000, 1), (70000, 64), (300, 7)])
def test_colsum(hip, rows, n):
    """Bias-gradient column sums: wide (GEMM outputs), narrow dense (conv channels over millions of pixels), ragged."""
    x = _rand(rows, n, seed=5)
    out = torch.empty(n, device=DEV)
    ws = _ws(64 << 20)
    hip.call('dm_colsum', rows, n, hip.fptr(x), n, hip.fptr(out), hip.ptr(ws), ws.numel(), hip.stream())
    _close(out, x.double().sum(0), 1e-5, 2e-4 * np.sqrt(rows / 2500), 'colsum')


# ------------------------------------------------------------------------------------------- GRU
def test_gru_gates(hip):
    """Against torch.nn.GRUCell (rnn.py:48-49) given the same pre-activations, forward and backward."""
    rows, D, Hd = 50, 600, 1000
    cell = torch.nn.GRUCell(Hd, D).double()
    x = _rand(rows, Hd, seed=1).double().cpu().requires_grad_(True)
    h = _rand(rows, D, seed=2).double().cpu().requires_grad_(True)
    hn = cell(x, h)
    dh = _rand(rows, D, seed=3)
    hn.backward(dh.double().cpu())
    gi = (x @ cell.weight_ih.t() + cell.bias_ih).detach().float().to(DEV).contiguous()
    gh = (h @ cell.weight_hh.t() + cell.bias_hh).detach().float().to(DEV).contiguous()
    hin = h.detach().float().to(DEV)
    out = torch.empty(rows, D, device=DEV)
    hip.call('dm_gru_gates_fwd', rows, D, hip.fptr(gi), hip.fptr(gh), hip.fptr(hin), D, hip.fptr(out), D, hip.stream())
    _close(out, hn, 1e-5, 1e-5, 'gru fwd')
    dgi = torch.empty_like(gi)
    dgh = torch.empty_like(gh)
    dhin = torch.empty(rows, D, device=DEV)
    hip.call('dm_gru_gates_bwd', rows, D, hip.fptr(gi), hip.fptr(gh), hip.fptr(hin), D, hip.fptr(dh), D, hip.fptr(dgi),
             hip.fptr(dgh), hip.fptr(dhin), D, hip.stream())
    # dx = dgi @ W_ih ; dh = dh*u + dgh @ W_hh
    dx = dgi.double().cpu() @ cell.weight_ih.detach()
    dhh = dhin.double().cpu() + dgh.double().cpu() @ cell.weight_hh.detach()
    _close(dx, x.grad, 1e-4, 1e-5, 'gru dx')
    _close(dhh, h.grad, 1e-4, 1e-5, 'gru dh')


# ------------------------------------------------------------------------------------------- sampler
def _oracle_sample(logits, u):
    """The shared inverse-CDF rule on CPU fp32: idx = #{k: cdf_k <= u*cdf_last}."""
    p = torch.softmax(logits.float().cpu(), -1)
    cdf = torch.cumsum(p, -1)
    target = u.float().cpu().unsqueeze(-1) * cdf[..., -1:]
    idx = (cdf <= target).sum(-1).clamp(max=logits.shape[-1] - 1)
    margin = (cdf - target).abs().min(-1).values
    return idx, margin


@pytest.mark.parametrize('rows,groups,C', [(50, 32, 32), (2500, 1, 18), (7, 8, 8)])
def test_sample_onehot(hip, rows, groups, C):
    logits = _rand(rows, groups * C, seed=1, scale=2.0)
    g = torch.Generator().manual_seed(2)
    u = torch.rand(rows, groups, generator=g).to(DEV)
    onehot = torch.empty(rows, groups * C, device=DEV)
    idx = torch.empty(rows, groups, dtype=torch.int32, device=DEV)
    hip.call('dm_sample_onehot', rows, groups, C, hip.fptr(logits), groups * C, hip.fptr(u), None, hip.fptr(onehot),
             groups * C, hip.ptr(idx), hip.stream())
    ref, margin = _oracle_sample(logits.reshape(rows, groups, C), u)
    got = idx.cpu().long()
    diff = got != ref
    # any disagreement must sit on a CDF boundary (|cdf - u*total| < 1e-5)
    assert (margin[diff] < 1e-5).all(), f'{int(diff.sum())} index mismatches away from CDF boundaries'
    assert diff.float().mean() < 1e-3
    oh = onehot.reshape(rows, groups, C).cpu()
    assert torch.equal(oh, F.one_hot(got, C).float())
    # forced indices bypass the sampler bit-exactly
    forced = ref.int().to(DEV).contiguous()
    hip.call('dm_sample_onehot', rows, groups, C, hip.fptr(logits), groups * C, None, hip.ptr(forced), hip.fptr(onehot),
             groups * C, hip.ptr(idx), hip.stream())
    assert torch.equal(idx.cpu().long(), ref)


def test_sample_onehot_edges(hip):
    """u=0 picks the first category with non-zero mass; u just below 1 picks the last; a delta distribution is exact."""
    C = 32
    logits = torch.zeros(3, C, device=DEV)
    logits[2, 5] = 200.0
    u = torch.tensor([[0.0], [0.99999], [0.5]], device=DEV)
    onehot = torch.empty(3, C, device=DEV)
    idx = torch.empty(3, 1, dtype=torch.int32, device=DEV)
    hip.call('dm_sample_onehot', 3, 1, C, hip.fptr(logits), C, hip.fptr(u), None, hip.fptr(onehot), C, hip.ptr(idx),
             hip.stream())
    assert idx.cpu().flatten().tolist() == [0, C - 1, 5]


# ------------------------------------------------------------------------------------------- KL / ST
def test_kl_balance(hip):
    rows, S, C = 250, 32, 32
    post = _rand(rows, S * C, seed=1, scale=1.5)
    prior = _rand(rows, S * C, seed=2, scale=1.5)
    kl = torch.empty(rows, device=DEV)
    ep = torch.empty(rows, device=DEV)
    eq = torch.empty(rows, device=DEV)
    hip.call('dm_kl_balance_fwd', rows, S, C, hip.fptr(post), hip.fptr(prior), hip.fptr(kl), hip.fptr(ep), hip.fptr(eq),
             hip.stream())
    import torch.distributions as D

    def dist(x):
        return D.Independent(D.OneHotCategoricalStraightThrough(logits=x.reshape(rows, S, C)), 1)
    a = post.double().cpu().requires_grad_(True)
    b = prior.double().cpu().requires_grad_(True)
    klr = D.kl_divergence(dist(a), dist(b))
    _close(kl, klr, 1e-5, 1e-5, 'kl')
    _close(ep, dist(a).entropy(), 1e-5, 1e-5, 'entropy post')
    _close(eq, dist(b).entropy(), 1e-5, 1e-5, 'entropy prior')
    # balanced gradients (dreamer.py:337-339): 0.2*KL(post||sg prior) + 0.8*KL(sg post||prior), mean over rows
    bal = 0.8
    loss = ((1 - bal) * D.kl_divergence(dist(a), dist(b.detach())) + bal * D.kl_divergence(dist(a.detach()), dist(b))).mean()
    loss.backward()
    dpost = torch.empty_like(post)
    dprior = torch.empty_like(prior)
    hip.call('dm_kl_balance_bwd', rows, S, C, hip.fptr(post), hip.fptr(prior), (1 - bal) / rows, bal / rows,
             hip.fptr(dpost), hip.fptr(dprior), hip.stream())
    _close(dpost, a.grad, 1e-4, 1e-8, 'dpost')
    _close(dprior, b.grad, 1e-4, 1e-8, 'dprior')


def test_st_softmax_bwd(hip):
    rows, S, C = 50, 32, 32
    logits = _rand(rows, S * C, seed=1)
    dz = _rand(rows, S * C, seed=2)
    base = _rand(rows, S * C, seed=3)
    out = base.clone()
    hip.call('dm_st_softmax_bwd', rows, S, C, hip.fptr(logits), S * C, hip.fptr(dz), S * C, hip.fptr(out), S * C, 1,
             hip.stream())
    x = logits.double().cpu().requires_grad_(True)
    p = torch.softmax(x.reshape(rows, S, C), -1)
    (p * dz.double().cpu().reshape(rows, S, C)).sum().backward()
    _close(out, base.double().cpu() + x.grad, 1e-4, 1e-6, 'st bwd accum')


def test_mask_rows(hip):
    x = _rand(50, 600, seed=1)
    reset = (torch.arange(50) % 3 == 0).to(torch.uint8).to(DEV)
    y = torch.empty(50, 600, device=DEV)
    hip.call('dm_mask_rows', 50, 600, hip.fptr(x), 600, hip.ptr(reset), hip.fptr(y), 600, hip.stream())
    assert torch.equal(y, x * (1 - reset.float()).unsqueeze(1))


# ------------------------------------------------------------------------------------------- conv gathers
@pytest.mark.parametrize('hb,c,k', [(14, 8, 4), (13, 16, 5), (30, 8, 6), (64, 3, 6), (5, 32, 5), (31, 8, 4)])
def test_im2col_col2im(hip, hb, c, k):
    n = 5
    hs = (hb - k) // 2 + 1
    big = _rand(n, hb, hb, c, seed=1)
    col = torch.empty(n * hs * hs, k * k * c, device=DEV)
    hip.call('dm_im2col_s2', n, hb, hb, c, k, hip.fptr(big), 0, hip.fptr(col), hip.stream())
    unf = F.unfold(big.permute(0, 3, 1, 2), k, stride=2)             # (n, c*k*k, L) with (c,ky,kx) order
    ref = unf.reshape(n, c, k, k, hs * hs).permute(0, 4, 2, 3, 1).reshape(n * hs * hs, k * k * c)
    assert torch.equal(col, ref.contiguous())
    # col2im is the adjoint (fold); plus bias/ELU epilogue
    colr = _rand(n * hs * hs, k * k * c, seed=2)
    bias = _rand(c, seed=3)
    out = torch.empty(n, hb, hb, c, device=DEV)
    hip.call('dm_col2im_s2', n, hb, hb, c, k, hip.fptr(colr), hip.fptr(bias), hip.DM_C2I_ELU, None, hip.fptr(out),
             hip.stream())
    cf = colr.reshape(n, hs * hs, k, k, c).permute(0, 4, 2, 3, 1).reshape(n, c * k * k, hs * hs)
    fold = F.fold(cf.double(), (hb, hb), k, stride=2).permute(0, 2, 3, 1)
    _close(out, F.elu(fold + bias.double()), 1e-5, 1e-5, 'col2im')


def test_im2col_nchw(hip):
    n, c, hb, k = 4, 3, 64, 4
    hs = 31
    big = _rand(n, c, hb, hb, seed=1)
    col = torch.empty(n * hs * hs, c * k * k, device=DEV)
    hip.call('dm_im2col_s2', n, hb, hb, c, k, hip.fptr(big), 1, hip.fptr(col), hip.stream())
    ref = F.unfold(big, k, stride=2).permute(0, 2, 1).reshape(n * hs * hs, c * k * k)
    assert torch.equal(col, ref.contiguous())


# ------------------------------------------------------------------------------------------- losses / GAE
def test_head_loss(hip):
    import torch.distributions as D
    rows = 2500
    out = _rand(rows, seed=1, scale=2.0)
    tgt_r = torch.tanh(_rand(rows, seed=2))
    tgt_t = (torch.rand(rows, generator=torch.Generator().manual_seed(3)) < 0.1).float().to(DEV)
    for kind, tgt in ((0, tgt_r), (1, tgt_t)):
        loss = torch.empty(rows, device=DEV)
        dout = torch.empty(rows, device=DEV)
        mean = torch.empty(rows, device=DEV)
        hip.call('dm_head_loss', kind, rows, hip.fptr(out), hip.fptr(tgt), 1.0 / rows, 0.0, hip.fptr(loss),
                 hip.fptr(dout), hip.fptr(mean), hip.stream())
        o = out.double().cpu().requires_grad_(True)
        if kind == 0:
            std = 0.3989422804
            ref = -D.Normal(o, torch.ones_like(o) * std).log_prob(tgt.double().cpu()) * std ** 2
            refmean = o
        else:
            ref = -D.Bernoulli(logits=o).log_prob(tgt.double().cpu())
            refmean = torch.sigmoid(o)
        ref.mean().backward()
        _close(loss, ref, 1e-5, 1e-6, f'head loss {kind}')
        _close(dout, o.grad, 1e-5, 1e-9, f'head dout {kind}')
        _close(mean, refmean, 1e-5, 1e-6, f'head mean {kind}')


def test_gae_and_ac_losses(hip):
    """a2c.py:81-131 restated with torch ops."""
    H, M, A = 15, 2500, 18
    J = H + 1
    gamma, lam, ent_w = 0.99, 0.95, 1e-3
    reward = _rand(J, M, seed=1)
    terminal = torch.sigmoid(_rand(J, M, seed=2) - 3)
    value_t = _rand(J, M, seed=3)
    value = _rand(J, M, seed=4)
    logits = _rand(H, M, A, seed=5)
    act = torch.randint(0, A, (H, M), generator=torch.Generator().manual_seed(6)).int().to(DEV)
    adv = torch.empty(H, M, device=DEV)
    agae = torch.empty(H, M, device=DEV)
    vtgt = torch.empty(H, M, device=DEV)
    wgt = torch.empty(H, M, device=DEV)
    hip.call('dm_gae_losses', H, M, gamma, lam, hip.fptr(reward), hip.fptr(terminal), hip.fptr(value_t), hip.fptr(adv),
             hip.fptr(agae), hip.fptr(vtgt), hip.fptr(wgt), hip.stream())
    r, tm, vt = reward.double().cpu(), terminal.double().cpu(), value_t.double().cpu()
    advantage = -vt[:-1] + r[1:] + gamma * (1 - tm[1:]) * vt[1:]
    acc, out = None, []
    for a_, t_ in zip(reversed(advantage.unbind()), reversed(tm[1:].unbind())):
        acc = a_ if acc is None else a_ + lam * gamma * (1 - t_) * acc
        out.append(acc)
    out.reverse()
    agae_ref = torch.stack(out)
    w_ref = (1 - tm[:-1]).log().cumsum(0).exp()
    _close(adv, advantage, 1e-5, 1e-5, 'advantage')
    _close(agae, agae_ref, 1e-5, 2e-5, 'advantage_gae')
    _close(vtgt, agae_ref + vt[:-1], 1e-5, 2e-5, 'value_target')
    _close(wgt, w_ref, 1e-5, 1e-6, 'reality weight')
    # critic
    rows = H * M
    lossc = torch.empty(rows, device=DEV)
    dval = torch.empty(rows, device=DEV)
    v0 = value[:-1].contiguous()
    hip.call('dm_critic_loss', rows, hip.fptr(v0), hip.fptr(vtgt), hip.fptr(wgt), 1.0 / rows, hip.fptr(lossc),
             hip.fptr(dval), hip.stream())
    v = v0.double().cpu().requires_grad_(True)
    lc = 0.5 * (vtgt.double().cpu() - v) ** 2 * wgt.double().cpu()
    lc.mean().backward()
    _close(lossc.reshape(H, M), lc, 1e-5, 1e-6, 'critic loss')
    _close(dval.reshape(H, M), v.grad, 1e-5, 1e-10, 'critic dvalue')
    # actor
    import torch.distributions as D
    lossa = torch.empty(rows, device=DEV)
    ent = torch.empty(rows, device=DEV)
    dlog = torch.empty(rows, A, device=DEV)
    hip.call('dm_actor_loss', rows, A, hip.fptr(logits), hip.ptr(act), hip.fptr(agae), hip.fptr(wgt), ent_w, 1.0 / rows,
             hip.fptr(lossa), hip.fptr(ent), hip.fptr(dlog), hip.stream())
    lg = logits.double().cpu().requires_grad_(True)
    dist = D.OneHotCategorical(logits=lg)
    onehot = F.one_hot(act.long().cpu(), A).double()
    la = (-dist.log_prob(onehot) * agae.double().cpu() - ent_w * dist.entropy()) * wgt.double().cpu()
    la.mean().backward()
    _close(lossa.reshape(H, M), la, 1e-5, 1e-6, 'actor loss')
    _close(ent.reshape(H, M), dist.entropy(), 1e-5, 1e-6, 'policy entropy')
    _close(dlog.reshape(H, M, A), lg.grad, 1e-4, 1e-10, 'actor dlogits')


def test_multi_sum(hip):
    xs = [_rand(n, seed=i) for i, n in enumerate((1, 37500, 2500, 12345))]
    items = (hip.dm_reduce_item * len(xs))()
    for i, x in enumerate(xs):
        items[i].x = x.data_ptr()
        items[i].n = x.numel()
        items[i].scale = 1.0 / x.numel()
    out = torch.empty(len(xs), device=DEV)
    hip.call('dm_multi_sum', len(xs), items, hip.fptr(out), hip.stream())
    ref = torch.stack([x.double().mean() for x in xs])
    _close(out, ref, 1e-5, 1e-6, 'multi_sum')


# ------------------------------------------------------------------------------------------- optimizer
def test_norm_clip_adamw(hip):
    """Three AdamW steps with clipping against torch.optim.AdamW + clip_grad_norm_ (dreamer.py:60-87)."""
    n = 1_000_003
    p0 = _rand(n + 1, seed=1)[:n].clone()
    ref_p = torch.nn.Parameter(p0.clone().cpu())
    opt = torch.optim.AdamW([ref_p], lr=3e-4, eps=1e-5)
    p = p0.clone()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    norm = torch.empty(2, device=DEV)
    ws = _ws(1 << 20)
    for step in range(1, 4):
        gcpu = (_rand(n, seed=10 + step) * (3.0 if step == 2 else 0.05)).cpu()
        ref_p.grad = gcpu.clone()
        total = torch.nn.utils.clip_grad_norm_([ref_p], 200.0)
        opt.step()
        g = gcpu.to(DEV)
        hip.call('dm_multi_tensor_norm_clip', hip.fptr(g), n, 200.0, hip.fptr(norm), hip.ptr(ws), ws.numel(), hip.stream())
        hip.call('dm_scale_inplace', hip.fptr(g), n, ctypes.c_void_p(norm.data_ptr() + 4), hip.stream())
        hip.call('dm_adamw_step', hip.fptr(p), hip.fptr(g), hip.fptr(m), hip.fptr(v), n, 3e-4, 0.9, 0.999, 1e-5, 0.01, step,
                 None, hip.stream())
        _close(norm[0], gcpu.double().norm(), 2e-6, 0, 'grad norm vs fp64')
        _close(norm[0], total, 1e-4, 0, 'grad norm vs torch fp32')
        _close(g, ref_p.grad, 1e-4, 1e-9, 'clipped grad')
        _close(p, ref_p.detach(), 2e-6, 2e-7, f'adamw step {step}')


def test_copy_axpby(hip):
    x = _rand(1000, seed=1)
    y = _rand(1000, seed=2)
    y0 = y.clone()
    hip.call('dm_axpby', 1000, 2.0, hip.fptr(x), 0.5, hip.fptr(y), hip.stream())
    _close(y, 2 * x + 0.5 * y0, 1e-6, 1e-6, 'axpby')
    hip.call('dm_copy_params', hip.fptr(y), hip.fptr(x), 1000, hip.stream())
    assert torch.equal(x, y)
