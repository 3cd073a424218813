"""The whole-network entry points (mi_cross_fwd / mi_cross_bwd, mi_din_lau_fwd / mi_din_lau_bwd) against the same
launch sequences issued one by one through the single entry points: same bits in every output and every parameter.
(The sequences themselves are held to the numpy oracles by test_dcn_gpu.py and test_din_gpu.py, which run through
the composites.)"""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda", 0)


def _dense(value, dev, matrix):
    from recommendation_amd.dcn import _Dense
    return _Dense(value, dev, matrix)


def _clone(p, dev):
    q = _dense(p.w.clone(), dev, p.h is not None)
    q.m.copy_(p.m)
    q.v.copy_(p.v)
    return q


def _same(a, b):
    for name in ('w', 'm', 'v', 'h', 'hT', 'grad'):
        x, y = getattr(a, name, None), getattr(b, name, None)
        if x is not None:
            assert torch.equal(x, y), name


@pytest.mark.parametrize("du_ready", [False, True])
@pytest.mark.parametrize("L,W,B", [(3, 128, 256), (1, 64, 64), (2, 448, 1024)])
def test_cross_network_calls_equal_the_layer_by_layer_sequence_bitwise(dev, L, W, B, du_ready):
    from recommendation_amd import ops
    gen = torch.Generator().manual_seed(3)
    Ws = [_dense(torch.randn((W, W), generator=gen) / np.sqrt(W) * 0.5, dev, True) for _ in range(L)]
    bs = [_dense(torch.randn((W,), generator=gen) * 0.05, dev, False) for _ in range(L)]
    for p in Ws + bs:                       # a step that is not the first: non-trivial Adam state
        p.m.copy_(torch.randn(p.m.shape, generator=gen).to(dev) * 1e-3)
        p.v.copy_(torch.rand(p.v.shape, generator=gen).to(dev) * 1e-4)
    Ws2, bs2 = [_clone(p, dev) for p in Ws], [_clone(p, dev) for p in bs]
    x0 = (torch.randn((B, W), generator=gen) * 0.5).to(dev).to(torch.bfloat16)
    dxL = (torch.randn((B, W), generator=gen) * 0.1).to(dev).to(torch.bfloat16)
    sched = torch.tensor([1e-3, 0.9, 0.999, 1.0], device=dev)
    b1, b2, eps = 0.9, 0.999, 1e-8
    bf = dict(dtype=torch.bfloat16, device=dev)
    # --- the sequence, launch by launch
    x = [x0] + [torch.empty((B, W), **bf) for _ in range(L)]
    u = [torch.empty((B, W), **bf) for _ in range(L)]
    for l in range(L):
        ops.gemm_bf16(x[l], Ws[l].h, ops.MI_EPI_CROSS_FWD, out=x[l + 1], out1=u[l], bias=bs[l].w, in0=x0, in1=x[l])
    dx = [torch.empty((B, W), **bf) for _ in range(L)] + [dxL.clone()]
    du = [torch.empty((B, W), **bf) for _ in range(2)]
    ops.mul_bf16(dx[L], x0, du[L & 1])
    s = ops.gemm_wgrad_splits(W, W, B)
    vs = max(1, min(256, B // 64))
    slab = torch.empty((max(s * W * W, vs * W),), device=dev)
    for l in reversed(range(L)):
        d = du[(l + 1) & 1]
        if l > 0:
            ops.gemm_bf16(d, Ws[l].hT, ops.MI_EPI_CROSS_BWD, out=dx[l], in1=dx[l + 1], out1=du[l & 1], in0=x0)
        else:
            ops.gemm_bf16(d, Ws[0].hT, ops.MI_EPI_CROSS_BWD, out=dx[0], in1=dx[1])
        ops.gemm_bf16(d, x[l], ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=W, N=W, K=B, k_splits=s,
                      out=slab[:s * W * W].view(s, W, W))
        ops.slab_adam(slab, s, W * W, W * W, Ws[l].w, Ws[l].m, Ws[l].v, b1, b2, eps, lr_dev=sched, w_bf16=Ws[l].h,
                      grad_out=Ws[l].grad)
        ops.transpose_bf16(Ws[l].h, Ws[l].hT)
        part = slab[:vs * W].view(vs, W)
        ops.colsum_bf16(d, vs, part, N=W)
        ops.slab_adam(slab, vs, W, W, bs[l].w, bs[l].m, bs[l].v, b1, b2, eps, lr_dev=sched, grad_out=bs[l].grad)
    dx0 = torch.empty((B, W), device=dev)
    ops.cross_input_grad(dx, u, dx0)
    # --- the two calls
    net = ops.CrossNetwork(Ws2, bs2, W, B, dev, keep_grads=True)
    xL = net.forward(x0)
    assert torch.equal(xL, x[L])
    for l in range(L):
        assert torch.equal(net.part(ops.MI_CROSS_PART_U, l), u[l])
    net.part(ops.MI_CROSS_PART_DX, L).copy_(dxL)
    if du_ready:
        ops.mul_bf16(net.part(ops.MI_CROSS_PART_DX, L), x0, net.part(ops.MI_CROSS_PART_DU, L & 1))
    dx0_c = torch.empty((B, W), device=dev)
    net.backward(x0, dx0_c, b1, b2, eps, lr_dev=sched, du_last_ready=du_ready)
    assert torch.equal(dx0_c, dx0)
    for a, b in zip(Ws + bs, Ws2 + bs2):
        _same(a, b)
    # the product kernel against torch's own bf16 product (one rounding of the exact product either way)
    want = (dxL.float() * x0.float()).to(torch.bfloat16)
    got = torch.empty_like(want)
    ops.mul_bf16(dxL, x0, got)
    assert torch.equal(got, want)


def test_cross_network_calls_check_their_arguments(dev):
    from recommendation_amd import ops
    gen = torch.Generator().manual_seed(4)
    W, B = 64, 64
    Ws = [_dense(torch.randn((W, W), generator=gen), dev, True)]
    bs = [_dense(torch.randn((W,), generator=gen), dev, False)]
    net = ops.CrossNetwork(Ws, bs, W, B, dev)
    x0 = torch.zeros((B, W), dtype=torch.bfloat16, device=dev)
    net.ws = net.ws[:net.ws.numel() - 256]                     # a workspace that is too small
    with pytest.raises(ops.MiError, match="workspace"):
        net.forward(x0)
    with pytest.raises(ops.MiError):
        ops.CrossNetwork(Ws * 9, bs * 9, W, B, dev)            # more layers than the entry point takes
    with pytest.raises(ops.MiError):
        net.part(ops.MI_CROSS_PART_U, 5)


@pytest.mark.parametrize("factored", [True, False])
def test_din_unit_calls_equal_the_launch_by_launch_sequence_bitwise(dev, factored):
    """One training step of DIN through mi_din_lau_fwd / mi_din_lau_bwd (din.py) against the same step with the
    unit issued launch by launch through the single entry points."""
    from recommendation_amd import ops
    from recommendation_amd.din import DIN
    rows, B, L, D = 5000, 64, 10, 64
    a = DIN(rows, dev, max_len=L, seed=2, keep_grads=True, factored=factored)
    b = DIN(rows, dev, max_len=L, seed=2, keep_grads=True, factored=factored)
    rng = np.random.default_rng(5)
    hist = torch.from_numpy(rng.integers(0, rows, size=(B, L))).to(dev)
    qid = torch.from_numpy(rng.integers(0, rows, size=(B,))).to(dev)
    lengths = torch.from_numpy(rng.integers(0, L + 1, size=(B,)).astype(np.int32)).to(dev)
    labels = torch.from_numpy((rng.random(B) < 0.5).astype(np.float32)).to(dev)
    if not factored:
        for step in range(2):
            la = a.train_step(hist, qid, lengths, labels).clone()
            lb = _din_step_launch_by_launch(b, hist, qid, lengths, labels).clone()
            assert torch.equal(la, lb)
            assert torch.equal(a.feat, b.feat) and torch.equal(a.dE, b_dE(b))
        for name, p in a.dense_params().items():
            _same(p, b.dense_params()[name])
        assert torch.equal(a.E, b.E) and torch.equal(a.E_acc, b.E_acc)
        return
    # The factored unit runs as one launch per direction (mi_din_lau_fwd_fused / mi_din_lau_bwd_fused): its forward
    # and the history rows of dE are bit-identical to the sequence; the sums over positions / samples (the small
    # parameters' gradients, the candidate's row) are added in another order, so the first step's gradients are held
    # to 1e-5 of their scale and the parameters after it to one Adam step of the disagreeing elements.
    from errlog import held
    la = a.train_step(hist, qid, lengths, labels).clone()
    lb = _din_step_launch_by_launch(b, hist, qid, lengths, labels).clone()
    assert torch.equal(la, lb) and torch.equal(a.feat, b.feat)
    L1 = L + 1
    assert torch.equal(a.dE.view(B, L1, D)[:, :L], b_dE(b).view(B, L1, D)[:, :L])
    held("DIN composite vs sequence: candidate rows of dE", float((a.dE - b_dE(b)).abs().max()) / float(b_dE(b).abs().max()), 1e-5)
    for name, p in a.dense_params().items():
        q = b.dense_params()[name]
        scale = float(q.grad.abs().max()) + 1e-30
        held("DIN composite vs sequence: grad " + name, float((p.grad - q.grad).abs().max()) / scale, 1e-5)
        held("DIN composite vs sequence: param " + name, float((p.w - q.w).abs().max()), 2.5 * a.lr)
    held("DIN composite vs sequence: table", float((a.E - b.E).abs().max()), 2.5 * a.emb_lr)


def b_dE(m):
    return m._seq['dE']


def _din_step_launch_by_launch(m, hist, qid, lengths, labels):
    """din.py's training step with the activation unit written out as its single launches (own buffers)."""
    from recommendation_amd import ops
    B, L = hist.shape
    D, H, H2 = m.dim, m.h1p, m.h2p
    M = B * L
    dev = m.device
    m._buffers(B, hist)
    if not hasattr(m, '_seq'):
        bf = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        ac = (2 if m.factored else 4) * D
        vs = max(1, min(1024, M // 64))
        split = {(H, ac): ops.gemm_wgrad_splits(H, ac, M), (H2, H): ops.gemm_wgrad_splits(H2, H, M)}
        q = dict(A=torch.empty((M, ac), **bf), dA=torch.empty((M, ac), **bf), H1=torch.empty((M, H), **bf),
                 dZ1=torch.empty((M, H), **bf), H2=torch.empty((M, H2), **bf), dH2=torch.empty((M, H2), **bf),
                 a=torch.empty((M,), **f32), da=torch.empty((M,), **f32), dE=torch.empty((B * (L + 1), D), **f32),
                 vs=vs, split=split,
                 slab=torch.empty((max(max(s * k[0] * k[1] for k, s in split.items()), vs * (H + 4), vs * (H2 + 4)),), **f32))
        if m.factored:
            q.update(rowbias=torch.empty((B, H), **f32), S=torch.empty((B, H), **f32), dq=torch.empty((B, D), **f32),
                     qpart=torch.empty((ops.din_qterm_part_floats(B, H, D),), **f32), gWf=torch.empty((H, 2 * D), **f32),
                     gq=torch.empty((D * H + H,), **f32))
        m._seq = q
    q = m._seq
    A, a, feat = q['A'], q['a'], m.feat

    def apply(p, n_slabs, n, stride):
        ops.slab_adam(q['slab'], n_slabs, n, stride, p.w, p.m, p.v, m.beta1, m.beta2, m.eps, lr_dev=m.sched, w_bf16=p.h,
                      grad_out=p.grad)

    def weight_step(p, dY, X):
        n_out, k_in = p.w.shape
        s = q['split'][(n_out, k_in)]
        ops.gemm_bf16(dY, X, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=n_out, N=k_in, K=dY.shape[0], k_splits=s,
                      out=q['slab'][:s * n_out * k_in].view(s, n_out, k_in))
        apply(p, s, n_out * k_in, n_out * k_in)
        ops.transpose_bf16(p.h, p.hT)

    def bias_step(p, dY):
        n = p.w.numel()
        ops.colsum_bf16(dY, q['vs'], q['slab'][:q['vs'] * n].view(q['vs'], n), N=n)
        apply(p, q['vs'], n, n)

    ops.din_lau_build(m.E, hist, qid, lengths, A, m.ids_all, q_out=feat[:, D:], err=m.err)
    if m.factored:
        ops.din_qterm_fwd(feat[:, D:], m.Wq, m.b1.w, q['rowbias'])
        ops.gemm_bf16(A, m.W1f, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=q['H1'], bias=q['rowbias'], bias_rows=L)
    else:
        ops.gemm_bf16(A, m.W1.h, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=q['H1'], bias=m.b1.w)
    ops.gemm_bf16(q['H1'], m.W2.h, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=q['H2'], bias=m.b2.w)
    ops.head_fwd(q['H2'], m.w3.w[:H2], m.w3.w[H2:H2 + 1], None, 0.0, a, None, None)
    ops.din_lau_pool(A, a, m.ids_all, D, feat)
    ops.linear_fwd(feat, m.head.w[:2 * D].view(2 * D, 1), m.head.w[2 * D:], out=m.logit.view(B, 1))
    ops.bce_logits(m.logit, labels, 1.0 / B, m.loss_rows, m.g)
    ops.reduce_sum(m.loss_rows, 1.0 / B, out=m.loss)
    ops.adam_schedule_step(m.sched, m.lr, m.beta1, m.beta2)
    dfeat, dW, db = ops.linear_bwd(feat, m.head.w[:2 * D].view(2 * D, 1), m.g.view(B, 1))
    m.head.grad[:2 * D] = dW.view(-1)
    m.head.grad[2 * D:] = db
    ops.adam_tf_dense(m.head.w, m.head.m, m.head.v, m.head.grad, 0.0, m.beta1, m.beta2, m.eps, lr_dev=m.sched)
    ops.din_lau_pool_bwd(A, dfeat, m.ids_all, D, q['da'])
    part = q['slab'][:q['vs'] * (H2 + 4)].view(q['vs'], H2 + 4)
    part.zero_()
    ops.head_bwd(q['da'], m.w3.w[:H2], q['H2'], q['dH2'], q['vs'], part)
    apply(m.w3, q['vs'], H2 + 4, H2 + 4)
    ops.gemm_bf16(q['dH2'], m.W2.hT, ops.MI_EPI_MASK, out=q['dZ1'], in0=q['H1'])
    weight_step(m.W2, q['dH2'], q['H1'])
    bias_step(m.b2, q['dH2'])
    if m.factored:
        qv = feat[:, D:]
        ops.colsum_bf16(q['dZ1'], B, q['S'], N=H)
        ops.din_qterm_bwd(qv, q['S'], m.Wq, dfeat[:, D:], q['dq'], q['qpart'])
        stride = D * H + H
        chunks = q['qpart'].numel() // stride
        ops.slab_adam(q['qpart'], chunks, stride, stride, None, None, None, 0.0, 0.0, 0.0, grad_out=q['gq'])
        ops.slab_adam(q['gq'][D * H:], 1, H, H, m.b1.w, m.b1.m, m.b1.v, m.beta1, m.beta2, m.eps, lr_dev=m.sched,
                      grad_out=m.b1.grad)
        ops.gemm_bf16(q['dZ1'], m.W1fT, ops.MI_EPI_MASK, out=q['dA'])
        s = q['split'][(H, 2 * D)]
        ops.gemm_bf16(q['dZ1'], A, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=H, N=2 * D, K=M, k_splits=s,
                      out=q['slab'][:s * H * 2 * D].view(s, H, 2 * D))
        ops.slab_adam(q['slab'], s, H * 2 * D, H * 2 * D, None, None, None, 0.0, 0.0, 0.0, grad_out=q['gWf'])
        ops.din_w1_step(q['gWf'], 1, H * 2 * D, q['gq'], 1, stride, m.W1.w, m.W1.m, m.W1.v, m.beta1, m.beta2, m.eps,
                        m.W1f, m.W1fT, m.Wq, m.WqT, lr_dev=m.sched, grad_out=m.W1.grad)
        ops.din_lau_scatter_fact(A, q['dA'], a, dfeat, q['dq'], qv, m.ids_all, D, q['dE'])
    else:
        ops.gemm_bf16(q['dZ1'], m.W1.hT, ops.MI_EPI_MASK, out=q['dA'])
        weight_step(m.W1, q['dZ1'], A)
        bias_step(m.b1, q['dZ1'])
        ops.din_lau_scatter(A, q['dA'], a, dfeat, dfeat[:, D:], m.ids_all, D, q['dE'])
    m.plan.prepare(m.ids_all.view(-1))
    ops.bwd_apply_adagrad_rowwise(m.plan, q['dE'], m.E, m.E_acc, m.emb_lr, m.emb_eps, grad_ld=D)
    return m.loss


@pytest.mark.parametrize("B,L,idx", [(64, 10, torch.int64), (32, 100, torch.int32), (8, 112, torch.int64), (8, 1, torch.int32)])
def test_din_fused_forward_equals_the_separate_launches_bitwise(dev, B, L, idx):
    """mi_din_lau_fwd_fused (one workgroup per sample, layers out of LDS) against mi_din_lau_build + mi_din_qterm_fwd +
    two mi_gemm_bf16 + mi_head_fwd + mi_din_lau_pool on the same inputs: every output bit for bit, with empty
    histories, masked positions (-1), ids out of range (reported, rows zero) and a candidate out of range."""
    from recommendation_amd import ops
    rows, D, H, H2 = 3000, 64, 128, 64
    assert ops.din_lau_fwd_fused_ok(D, H, H2, L)
    assert not ops.din_lau_fwd_fused_ok(D, H, H2, 113) and not ops.din_lau_fwd_fused_ok(D, H, H2, L, factored=False)
    g = torch.Generator(device='cpu').manual_seed(11 + B)
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)
    E = (torch.randn((rows, D), generator=g) * 0.3).to(dev)
    W1f = (torch.randn((H, 2 * D), generator=g) * 0.1).to(dev).to(torch.bfloat16)
    Wq = (torch.randn((H, D), generator=g) * 0.1).to(dev)
    b1 = (torch.randn((H,), generator=g) * 0.1).to(dev)
    W2 = (torch.randn((H2, H), generator=g) * 0.1).to(dev).to(torch.bfloat16)
    b2 = (torch.randn((H2,), generator=g) * 0.1).to(dev)
    w3 = torch.zeros((H2 + 4,), **f32)
    w3[:H2 + 1] = (torch.randn((H2 + 1,), generator=g) * 0.2).to(dev)
    hist = torch.randint(0, rows, (B, L), generator=g)
    hist[0, 0] = -1                      # a masked position inside the history
    if B > 2 and L > 2:
        hist[2, 1] = rows + 5            # out of range: reported, the row contributes nothing
    qid = torch.randint(0, rows, (B,), generator=g)
    qid[B - 1] = rows                    # candidate out of range: q = 0
    lengths = torch.randint(0, L + 1, (B,), generator=g).to(torch.int32)
    lengths[0] = L
    if B > 2:
        lengths[1] = 0                   # an empty history
        lengths[2] = L
    hist, qid, lengths = hist.to(idx).to(dev), qid.to(idx).to(dev), lengths.to(dev)
    M = B * L

    def outputs():
        return dict(A=torch.full((M, 2 * D), 7.0, **bf), H1=torch.full((M, H), 7.0, **bf), H2=torch.full((M, H2), 7.0, **bf),
                    a=torch.full((M,), 7.0, **f32), ids_all=torch.full((B, L + 1), 7, dtype=torch.int32, device=dev),
                    feat=torch.full((B, 2 * D), 7.0, **f32), err=torch.zeros((1,), dtype=torch.int32, device=dev))
    x, y = outputs(), outputs()
    ops.din_lau_fwd_fused(E, hist, qid, lengths, W1f, Wq, b1, W2, b2, w3, x['A'], x['H1'], x['H2'], x['a'], x['ids_all'],
                          x['feat'], torch.empty((B, H), **f32), err=x['err'])
    rowbias = torch.empty((B, H), **f32)
    ops.din_lau_build(E, hist, qid, lengths, y['A'], y['ids_all'], q_out=y['feat'][:, D:], err=y['err'])
    ops.din_qterm_fwd(y['feat'][:, D:], Wq, b1, rowbias)
    ops.gemm_bf16(y['A'], W1f, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=y['H1'], bias=rowbias, bias_rows=L)
    ops.gemm_bf16(y['H1'], W2, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=y['H2'], bias=b2)
    ops.head_fwd(y['H2'], w3[:H2], w3[H2:H2 + 1], None, 0.0, y['a'], None, None)
    ops.din_lau_pool(y['A'], y['a'], y['ids_all'], D, y['feat'])
    torch.cuda.synchronize()
    for name in ('ids_all', 'err', 'A', 'H1', 'H2', 'a', 'feat'):
        assert torch.equal(x[name], y[name]), name
    assert int(x['err'][0]) != 0


@pytest.mark.parametrize("B,L", [(64, 10), (32, 100), (8, 112), (8, 1)])
def test_din_fused_backward_equals_the_separate_launches(dev, B, L):
    """mi_din_lau_bwd_fused against mi_din_lau_pool_bwd + mi_head_bwd + the two masked mi_gemm_bf16 +
    mi_din_lau_scatter_fact on the forward of the same batch: dZ2, dZ1 and the history rows of dE bit for bit; the
    sums (S, the head's and the second bias' gradient, the candidate's row) are added in another order: 1e-5 of their
    scale."""
    from recommendation_amd import ops
    rows, D, H, H2 = 3000, 64, 128, 64
    g = torch.Generator(device='cpu').manual_seed(23 + L)
    f32 = dict(dtype=torch.float32, device=dev)
    bf = dict(dtype=torch.bfloat16, device=dev)

    def rnd(*shape, scale=0.1):
        return (torch.randn(shape, generator=g) * scale).to(dev)
    E = rnd(rows, D, scale=0.3)
    W1f, Wq, b1 = rnd(H, 2 * D).to(torch.bfloat16), rnd(H, D), rnd(H)
    W2, b2 = rnd(H2, H).to(torch.bfloat16), rnd(H2)
    w3 = torch.zeros((H2 + 4,), **f32)
    w3[:H2 + 1] = rnd(H2 + 1, scale=0.2)
    W2T, W1fT = W2.t().contiguous(), W1f.t().contiguous()
    hist = torch.randint(0, rows, (B, L), generator=g)
    hist[0, 0] = -1
    qid = torch.randint(0, rows, (B,), generator=g)
    lengths = torch.randint(0, L + 1, (B,), generator=g).to(torch.int32)
    lengths[0] = L
    lengths[1] = 0
    hist, qid, lengths = hist.to(dev), qid.to(dev), lengths.to(dev)
    M = B * L
    A, H1t, H2t = torch.empty((M, 2 * D), **bf), torch.empty((M, H), **bf), torch.empty((M, H2), **bf)
    a, ids_all = torch.empty((M,), **f32), torch.empty((B, L + 1), dtype=torch.int32, device=dev)
    feat = torch.empty((B, 2 * D), **f32)
    ops.din_lau_fwd_fused(E, hist, qid, lengths, W1f, Wq, b1, W2, b2, w3, A, H1t, H2t, a, ids_all, feat,
                          torch.empty((B, H), **f32))
    dfeat = rnd(B, 2 * D, scale=1.0)
    # fused
    P = ops.din_lau_bwd_part_floats()
    slices = 8 if B % 8 == 0 else 1
    x = dict(dZ2=torch.full((M, H2), 7.0, **bf), dZ1=torch.full((M, H), 7.0, **bf), dE=torch.full((B * (L + 1), D), 7.0, **f32),
             S=torch.empty((B, H), **f32), part=torch.empty((B, P), **f32), sums=torch.empty((slices, P), **f32))
    ops.din_lau_bwd_fused(A, H1t, H2t, a, ids_all, feat, dfeat, w3, W2T, W1fT, x['dZ2'], x['dZ1'], x['dE'], x['S'],
                          x['part'], x['sums'])
    # separate launches
    da, dA = torch.empty((M,), **f32), torch.empty((M, 2 * D), **bf)
    y = dict(dZ2=torch.empty((M, H2), **bf), dZ1=torch.empty((M, H), **bf), dE=torch.empty((B * (L + 1), D), **f32),
             S=torch.empty((B, H), **f32))
    vs = max(1, min(1024, M // 64))
    part = torch.zeros((vs, H2 + 4), **f32)
    ops.din_lau_pool_bwd(A, dfeat, ids_all, D, da)
    ops.head_bwd(da, w3[:H2], H2t, y['dZ2'], vs, part)
    ops.gemm_bf16(y['dZ2'], W2T, ops.MI_EPI_MASK, out=y['dZ1'], in0=H1t)
    ops.colsum_bf16(y['dZ1'], B, y['S'], N=H)
    ops.gemm_bf16(y['dZ1'], W1fT, ops.MI_EPI_MASK, out=dA)
    ops.din_lau_scatter_fact(A, dA, a, dfeat, dfeat[:, D:], feat[:, D:], ids_all, D, y['dE'])
    b2part = torch.empty((vs, H2), **f32)
    ops.colsum_bf16(y['dZ2'], vs, b2part, N=H2)
    torch.cuda.synchronize()
    assert torch.equal(x['dZ2'], y['dZ2']) and torch.equal(x['dZ1'], y['dZ1'])
    xe, ye = x['dE'].view(B, L + 1, D), y['dE'].view(B, L + 1, D)
    assert torch.equal(xe[:, :L], ye[:, :L])

    def close(u, v, what):
        scale = float(v.abs().max()) + 1e-30
        err = float((u - v).abs().max()) / scale
        from errlog import held
        held("DIN fused backward B=%d L=%d %s" % (B, L, what), err, 1e-5)
    close(xe[:, L], ye[:, L], "candidate row")
    close(x['S'], y['S'], "S")
    close(x['sums'][:, :H2 + 1].sum(0), part[:, :H2 + 1].sum(0), "head gradient")
    close(x['sums'][:, H2 + 4:].sum(0), b2part.sum(0), "b2 gradient")
    assert float(x['sums'][:, H2 + 1:H2 + 4].abs().max()) == 0.0
