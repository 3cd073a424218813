"""bf16 MFMA contraction against torch fp32 matmul of the same bf16-rounded operands.
With small-integer operands every product and sum is exact in fp32, so those cases are bit-exact
and catch any wrong lane map / transposed tile; random operands are held to 1e-2 (bf16 outputs)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ints(shape, dev, seed, lo=-3, hi=4):
    g = torch.Generator(device='cpu').manual_seed(seed)
    return torch.randint(lo, hi, shape, generator=g).to(torch.bfloat16).to(dev)


def _operands(M, N, K, a_rm, b_rm, dev, seed, ints=True):
    if ints:
        a, b = _ints((M, K), dev, seed), _ints((N, K), dev, seed + 1)
    else:
        g = torch.Generator(device='cpu').manual_seed(seed)
        a = (torch.rand((M, K), generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
        b = (torch.rand((N, K), generator=g) * 2 - 1).to(torch.bfloat16).to(dev)
    A = a.t().contiguous() if a_rm else a
    B = b.t().contiguous() if b_rm else b
    return a, b, A, B


@pytest.mark.parametrize("a_rm,b_rm", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 448, 448), (1000, 72, 192), (8, 8, 64), (384, 1024, 512)])
def test_slab_exact_on_integers(dev, a_rm, b_rm, M, N, K):
    from recommendation_amd import ops
    a, b, A, B = _operands(M, N, K, a_rm, b_rm, dev, seed=M + N + K)
    out = torch.full((1, M, N), float('nan'), dtype=torch.float32, device=dev)
    ops.gemm_bf16(A, B, ops.MI_EPI_SLAB, a_rm=a_rm, b_rm=b_rm, out=out)
    want = a.float() @ b.float().t()
    assert torch.equal(out[0], want)


@pytest.mark.parametrize("b_rm", [False, True])
@pytest.mark.parametrize("M,N,K", [(8192, 1024, 64), (8192, 1024, 128), (8192, 1024, 192), (8200, 1024, 448),
                                   (16384, 448, 1024), (32768, 256, 320)])
def test_big_tile_exact_on_integers(dev, b_rm, M, N, K):
    """Shapes that take the 256 x 256 eight-wave tile (at least 128 tiles, at most 15 % padding columns), whose K loop
    runs across tile boundaries with its last two K tiles peeled: one, two, three and many K tiles, a tile over the
    last row, a tile over the last column."""
    from recommendation_amd import ops
    a, b, A, B = _operands(M, N, K, False, b_rm, dev, seed=M + N + K)
    out = torch.full((1, M, N), float('nan'), dtype=torch.float32, device=dev)
    ops.gemm_bf16(A, B, ops.MI_EPI_SLAB, b_rm=b_rm, out=out)
    want = a.float() @ b.float().t()
    assert torch.equal(out[0], want)
    # and through the bf16 epilogue (bias + ReLU), still exact on these integers (|sum| < 2^8 would be needed for
    # bf16: so compare after rounding the reference the same way)
    bias = _ints((N,), dev, 5).float()
    o2 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(A, B, ops.MI_EPI_BIAS_ACT, b_rm=b_rm, flags=ops.MI_EPI_RELU, out=o2, bias=bias)
    assert torch.equal(o2, torch.relu(want + bias).to(torch.bfloat16))


def test_identity_times_asymmetric(dev):
    """A = I picks out B's rows: a swapped row/column map in the result cannot hide."""
    from recommendation_amd import ops
    n = 128
    A = torch.eye(n, dtype=torch.bfloat16, device=dev)
    Bm = (torch.arange(n, device=dev)[:, None] * 2 + torch.arange(n, device=dev)[None, :] % 7).to(torch.bfloat16)
    out = torch.empty((1, n, n), dtype=torch.float32, device=dev)
    ops.gemm_bf16(A, Bm, ops.MI_EPI_SLAB, out=out)           # C = I * Bm^T
    assert torch.equal(out[0], Bm.float().t())


@pytest.mark.parametrize("splits", [2, 4, 16])
def test_split_k_slabs_sum_to_the_product(dev, splits):
    from recommendation_amd import ops
    M, N, K = 448, 448, 64 * 16
    a, b, A, B = _operands(M, N, K, True, True, dev, seed=9)
    out = torch.zeros((splits, M, N), dtype=torch.float32, device=dev)
    ops.gemm_bf16(A, B, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, k_splits=splits, out=out)
    assert torch.equal(out.sum(0), a.float() @ b.float().t())
    per = K // splits
    assert torch.equal(out[1], a[:, per:2 * per].float() @ b[:, per:2 * per].float().t())


def test_bias_relu_epilogue(dev):
    from recommendation_amd import ops
    M, N, K = 512, 1024, 448
    a, b, A, B = _operands(M, N, K, False, False, dev, seed=3, ints=False)
    bias = torch.linspace(-1, 1, N, device=dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(A, B, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias)
    want = torch.relu(a.float() @ b.float().t() + bias)
    err = (out.float() - want).abs().max() / want.abs().max()
    assert err <= 1e-2
    # bf16 rounding of an fp32-accumulated sum: half an ulp of bf16 (2^-9) of the value
    assert ((out.float() - want).abs() <= want.abs() * 2 ** -8 + 1e-6).all()


@pytest.mark.parametrize("M,N,K,rows", [(1200, 128, 128, 100), (512, 1024, 64, 8), (1000, 72, 192, 20)])
def test_bias_matrix_epilogue(dev, M, N, K, rows):
    """A bias that depends on the row group: bias[m // rows][n] (one bias row per sample when the
    product has one row per (sample, position)); interior, last-column and last-row tiles."""
    from recommendation_amd import ops
    a, b, A, B = _operands(M, N, K, False, False, dev, seed=M, ints=False)
    groups = -(-M // rows)
    bias = torch.randn((groups, N + 4), device=dev)[:, :N]            # rows 16-byte aligned, not contiguous
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(A, B, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=out, bias=bias, bias_rows=rows)
    want = torch.relu(a.float() @ b.float().t() + bias.repeat_interleave(rows, dim=0)[:M])
    assert ((out.float() - want).abs() <= want.abs() * 2 ** -8 + 1e-6).all()


def test_cross_forward_epilogue(dev):
    from recommendation_amd import ops
    M, d = 384, 448
    g = torch.Generator(device='cpu').manual_seed(1)
    x0 = (torch.randn((M, d), generator=g)).to(torch.bfloat16).to(dev)
    xl = (torch.randn((M, d), generator=g)).to(torch.bfloat16).to(dev)
    W = (torch.randn((d, d), generator=g) / d ** 0.5).to(torch.bfloat16).to(dev)
    bias = torch.randn(d, generator=g).to(dev)
    u = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
    xn = torch.empty((M, d), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(xl, W, ops.MI_EPI_CROSS_FWD, out=xn, out1=u, bias=bias, in0=x0, in1=xl)
    u_want = xl.float() @ W.float().t() + bias
    x_want = x0.float() * u_want + xl.float()
    assert (u.float() - u_want).abs().max() <= 1e-2 * u_want.abs().max()
    assert (xn.float() - x_want).abs().max() <= 1e-2 * x_want.abs().max()


def test_mask_and_cross_backward_epilogues(dev):
    from recommendation_amd import ops
    M, N, K = 256, 448, 1024                       # dX[M, N] = dY[M, K] * W[K, N]  (W reduction-major)
    g = torch.Generator(device='cpu').manual_seed(2)
    dY = _ints((M, K), dev, 5, -2, 3)
    W = _ints((K, N), dev, 6, -2, 3)
    H = torch.randn((M, N), generator=g).to(torch.bfloat16).to(dev)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(dY, W, ops.MI_EPI_MASK, b_rm=True, out=out, in0=H)
    want = (dY.float() @ W.float()) * (H.float() > 0)
    assert torch.equal(out.float(), want.to(torch.bfloat16).float())
    # cross backward: v = C + R; out = v; out1 = v*x0
    R, x0 = _ints((M, N), dev, 7), _ints((M, N), dev, 8)
    dx = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    du = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(dY, W, ops.MI_EPI_CROSS_BWD, b_rm=True, out=dx, in1=R, in0=x0, out1=du)
    v = dY.float() @ W.float() + R.float()
    assert torch.equal(dx.float(), v.to(torch.bfloat16).float())
    assert torch.equal(du.float(), (v * x0.float()).to(torch.bfloat16).float())
    dx2 = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    ops.gemm_bf16(dY, W, ops.MI_EPI_CROSS_BWD, b_rm=True, out=dx2)
    assert torch.equal(dx2.float(), (dY.float() @ W.float()).to(torch.bfloat16).float())


def test_cross_input_grad(dev):
    from recommendation_amd import ops
    M, N, L = 300, 448, 3
    dx = [_ints((M, N), dev, 20 + l) for l in range(L + 1)]
    u = [_ints((M, N), dev, 30 + l) for l in range(L)]
    out = torch.full((M, N), float('nan'), dtype=torch.float32, device=dev)
    ops.cross_input_grad(dx, u, out)
    want = dx[0].float()
    for l in range(L):
        want = want + dx[l + 1].float() * u[l].float()
    assert torch.equal(out, want)


def test_rejects_bad_shapes(dev):
    from recommendation_amd import ops
    from recommendation_amd._lib import MiError
    A = torch.zeros((64, 100), dtype=torch.bfloat16, device=dev)
    out = torch.zeros((1, 64, 64), dtype=torch.float32, device=dev)
    with pytest.raises(MiError):
        ops.gemm_bf16(A, A, ops.MI_EPI_SLAB, out=out)          # K = 100 is not a multiple of 64


@pytest.mark.parametrize("shape", [(1024, 448), (448, 448), (130, 70), (64, 64), (1, 5)])
def test_transpose_bf16_bit_exact(dev, shape):
    from recommendation_amd import ops
    R, C = shape
    src = torch.randn((R, C), device=dev).to(torch.bfloat16)
    dst = torch.zeros((C, R), dtype=torch.bfloat16, device=dev)
    ops.transpose_bf16(src, dst)
    assert torch.equal(dst, src.t().contiguous())


@pytest.mark.parametrize("M,N,slices", [(4096, 1024, 64), (8200, 128, 256), (3000, 64, 7), (512, 448, 8)])
def test_colsum_bf16_partials(dev, M, N, slices):
    """Per-slice column sums of a bf16 matrix (both thread layouts: 64 column threads, and 16 for
    matrices of up to 128 columns); small integers, so the fp32 sums are exact."""
    from recommendation_amd import ops
    g = torch.Generator().manual_seed(M + N)
    X = torch.randint(-4, 5, (M, N), generator=g).to(torch.bfloat16).to(dev)
    part = torch.full((slices, N), float('nan'), device=dev)
    ops.colsum_bf16(X, slices, part)
    rps = -(-M // slices)
    want = torch.stack([X[s * rps:(s + 1) * rps].float().sum(0) for s in range(slices)])
    assert torch.equal(part, want)


@pytest.mark.parametrize("n,S", [(1024, 256), (516, 256), (128, 1024), (1024 * 1024, 8), (448, 3)])
def test_slab_adam_sums_slabs_and_applies_tf_adam(dev, n, S):
    """Both reductions (a thread walking all slabs of its four elements; 16 lanes sharing the slabs
    of a short vector): integer slabs so the gradient is exact, then TensorFlow-form Adam."""
    from recommendation_amd import ops
    g = torch.Generator().manual_seed(n + S)
    slab = torch.randint(-3, 4, (S, n), generator=g).float().to(dev)
    w = torch.randn(n, generator=g).to(dev)
    m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    w0 = w.clone()
    grad = torch.empty(n, device=dev)
    wb = torch.empty(n, dtype=torch.bfloat16, device=dev)
    lr_t, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    ops.slab_adam(slab, S, n, n, w, m, v, b1, b2, eps, lr_t=lr_t, w_bf16=wb, grad_out=grad)
    gw = slab.sum(0)
    assert torch.equal(grad, gw)
    f32 = torch.float32
    one = torch.tensor(1.0, dtype=f32)                       # (1 - beta) formed in fp32, as TensorFlow's kernel does
    m_want = gw * (one - torch.tensor(b1, dtype=f32)).to(dev)
    v_want = gw * gw * (one - torch.tensor(b2, dtype=f32)).to(dev)
    w_want = w0 - (m_want * lr_t) / (v_want.sqrt() + eps)
    assert torch.allclose(m, m_want, rtol=2e-6, atol=0) and torch.allclose(v, v_want, rtol=2e-6, atol=0)
    assert torch.allclose(w, w_want, rtol=2e-6, atol=1e-7)
    assert torch.equal(wb, w.to(torch.bfloat16))
