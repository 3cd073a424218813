"""DCN-v2 on the bf16 matrix-core path against the numpy restatement evaluated in fp64.
Tolerance 1e-2 relative to each tensor's largest magnitude (bf16 operands and activations,
fp32 accumulation), as stated for the bf16 paths; index handling stays bit-exact.

Forward quantities are compared with the exact (unrounded) restatement.  Gradients are compared
with the restatement that rounds the STORED tensors to bf16 where the device does: against exact
arithmetic a ReLU gate whose pre-activation lies within bf16 rounding of zero opens on one side
and not the other, and each such flip is a 100 % error in one element of dz - 0.3 % of the gates
flip at these sizes, which moves every downstream sum by a few per cent whatever the kernels do."""
import numpy as np
import pytest
import torch

from oracle import dcn as odcn
from oracle import fm as ofm

pytestmark = pytest.mark.gpu

ROWS = [1460, 583, 3001, 305, 24, 3, 10, 4, 18, 15, 105, 27, 2173, 12, 50, 7]
DIM, ND = 16, 13            # d = 16*16+13 = 269 -> padded to 320


def _bf16_round(a):
    return torch.from_numpy(a).to(torch.bfloat16).float().numpy()


def _state(seed, mlp, n_cross=3):
    st = odcn.State(ROWS, DIM, ND, n_cross=n_cross, mlp=mlp, seed=seed)
    # start from weights that bf16 holds exactly, so that the first step compares like with like
    st.Wc = [_bf16_round(w) for w in st.Wc]
    st.Wm = [_bf16_round(w) for w in st.Wm]
    return st


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


@pytest.mark.parametrize("B,mlp", [(256, (128, 64)), (1024, (256, 128, 64))])
def test_forward_logits(dev, B, mlp):
    from recommendation_amd.dcn import DCNv2
    st = _state(1, mlp)
    m = DCNv2(ROWS, DIM, ND, dev, mlp=mlp)
    m.load_state(st)
    ids, dense, labels = ofm.synthetic_batch(B, ROWS, ND, seed=2, dist='zipf')
    z = m.forward(torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev), torch.from_numpy(labels).to(dev))
    want, k = odcn.forward(st, ids, dense, keep=True)
    # the input row: gathered embeddings and dense features, rounded once to bf16; padding stays zero
    x0 = m.x[0].float().cpu().numpy()
    assert np.array_equal(x0[:, :m.d], _bf16_round(k['xs'][0].astype(np.float32)))
    assert not x0[:, m.d:].any()
    for l in range(3):
        assert rel(m.x[l + 1].float().cpu().numpy()[:, :m.d], k['xs'][l + 1]) <= 1e-2
        assert not m.x[l + 1].float().cpu().numpy()[:, m.d:].any()
    assert rel(z.cpu().numpy(), want) <= 1e-2
    per, sig = ofm.bce_with_logits(want, labels.astype(np.float64))
    assert rel(m.loss_rows.cpu().numpy(), per) <= 1e-2
    assert rel(m.g.cpu().numpy(), (sig - labels) / B) <= 1e-2


def test_gradients_and_update(dev):
    from recommendation_amd.dcn import DCNv2
    B, mlp = 512, (256, 128)
    st = _state(3, mlp)
    m = DCNv2(ROWS, DIM, ND, dev, mlp=mlp, lr=1e-3, emb_lr=1e-2, keep_grads=True)
    m.load_state(st)
    ids, dense, labels = ofm.synthetic_batch(B, ROWS, ND, seed=4, dist='zipf')
    loss_exact, gx = odcn.grads(st, ids, dense, labels)[:2]          # exact arithmetic: printed, bounded by 5e-2
    loss_want, gr, uniq, dV, k = odcn.grads(st, ids, dense, labels, storage='bf16')
    assert abs(loss_want - loss_exact) <= 1e-2 * abs(loss_exact)
    V_before = m.V.clone()
    loss = m.train_step(torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev),
                        torch.from_numpy(labels).to(dev), check_ids=True)
    assert abs(float(loss.item()) - loss_want) <= 1e-2 * abs(loss_want)
    d = m.d
    got = m.dense_params()
    for l in range(3):
        assert rel(got['cross%d/W' % l].grad.cpu().numpy()[:d, :d], gr['cross%d/W' % l]) <= 1e-2, l
        assert rel(got['cross%d/b' % l].grad.cpu().numpy()[:d], gr['cross%d/b' % l]) <= 1e-2, l
        ex = rel(got['cross%d/W' % l].grad.cpu().numpy()[:d, :d], gx['cross%d/W' % l])
        exb = rel(got['cross%d/b' % l].grad.cpu().numpy()[:d], gx['cross%d/b' % l])
        print("cross%d W / b gradient vs the EXACT oracle: %.2e / %.2e" % (l, ex, exb))
        assert ex <= 5e-2 and exb <= 5e-2
        assert not got['cross%d/W' % l].grad.cpu().numpy()[d:, :].any()      # padding gets no gradient
        assert not got['cross%d/W' % l].grad.cpu().numpy()[:, d:].any()
    for i in range(len(mlp)):
        W = got['mlp%d/W' % i].grad.cpu().numpy()
        assert rel(W[:, :gr['mlp%d/W' % i].shape[1]], gr['mlp%d/W' % i]) <= 1e-2, i
        assert rel(got['mlp%d/b' % i].grad.cpu().numpy(), gr['mlp%d/b' % i]) <= 1e-2, i
        ex = rel(W[:, :gx['mlp%d/W' % i].shape[1]], gx['mlp%d/W' % i])
        exb = rel(got['mlp%d/b' % i].grad.cpu().numpy(), gx['mlp%d/b' % i])
        print("mlp%d W / b gradient vs the EXACT oracle: %.2e / %.2e" % (i, ex, exb))
        assert ex <= 5e-2 and exb <= 5e-2
    K = mlp[-1]
    wb = got['out/wb'].grad.cpu().numpy()
    assert rel(wb[:K], gr['out/w']) <= 1e-2
    assert abs(wb[K] - gr['out/b'][0]) <= 1e-2 * max(abs(gr['out/b'][0]), np.abs(gr['out/w']).max())
    assert rel(m.dx0.cpu().numpy()[:, :d], k['dx0']) <= 1e-2
    # embeddings: exactly the touched rows moved, by row-wise Adagrad of the summed gradient
    moved = (m.V != V_before).any(dim=1).cpu().numpy()
    assert np.array_equal(np.nonzero(moved)[0], uniq)
    acc = (dV * dV).mean(axis=1)
    want_V = st.V[uniq] - 1e-2 * dV / (np.sqrt(acc) + 1e-8)[:, None]
    step = 1e-2                                             # |update| <= emb_lr per element
    assert np.abs(m.V.cpu().numpy()[uniq] - want_V).max() <= 2e-2 * step * np.sqrt(DIM)
    # dense parameters after TensorFlow-form Adam's first step: |update| = lr (sign of the gradient)
    for name, p in got.items():
        if name == 'out/wb':
            continue
        g = gr[name]
        w0 = st.dense_params()[name]
        big = np.abs(g) > 1e-2 * np.abs(g).max()
        w1 = p.w.cpu().numpy()
        w1 = w1[:g.shape[0], :g.shape[1]] if g.ndim == 2 else w1[:g.shape[0]]
        assert np.allclose((w0 - w1)[big], 1e-3 * np.sign(g)[big], rtol=2e-2, atol=1e-7), name


def test_training_reduces_loss_like_the_oracle(dev):
    """A few steps on a fixed batch: the device loss follows the oracle's trajectory."""
    from recommendation_amd.dcn import DCNv2
    B, mlp = 256, (128, 64)
    st = _state(5, mlp)
    m = DCNv2(ROWS, DIM, ND, dev, mlp=mlp, lr=1e-3, emb_lr=1e-2)
    m.load_state(st)
    ids, dense, labels = ofm.synthetic_batch(B, ROWS, ND, seed=6, dist='zipf')
    t = [torch.from_numpy(a).to(dev) for a in (ids, dense, labels)]
    got, want = [], []
    for _ in range(8):
        got.append(float(m.train_step(*t).item()))
        want.append(float(odcn.train_step(st, ids, dense, labels, lr=1e-3, emb_lr=1e-2, storage='bf16')))
    assert want[-1] < want[0]
    assert np.allclose(got, want, rtol=1e-2)


def test_graph_replay_equals_eager_steps_bitwise(dev):
    """The step replayed as a HIP graph (two alternating batches, three rounds) leaves the same bits as the eager
    step: every dense parameter, the tables, the losses."""
    from recommendation_amd.dcn import DCNv2
    B, mlp = 256, (128, 64)
    batches = []
    for seed in (11, 12):
        ids, dense, labels = ofm.synthetic_batch(B, ROWS, ND, seed=seed, dist='zipf')
        batches.append([torch.from_numpy(a).to(dev) for a in (ids, dense, labels)])
    models = []
    for graph in (False, True):
        m = DCNv2(ROWS, DIM, ND, dev, mlp=mlp, lr=1e-3, emb_lr=1e-2)
        m.load_state(_state(9, mlp))
        losses = [float(m.train_step(*batches[i % 2], graph=graph).item()) for i in range(6)]
        models.append((m, losses))
    (a, la), (b, lb) = models
    assert la == lb
    assert torch.equal(a.V, b.V) and torch.equal(a.V_acc, b.V_acc)
    assert torch.equal(a.w_arena, b.w_arena) and torch.equal(a.m_arena, b.m_arena) and torch.equal(a.v_arena, b.v_arena)
    assert a.steps == b.steps == 6


def test_rejects_ragged_batch(dev):
    from recommendation_amd.dcn import DCNv2
    from recommendation_amd._lib import MiError
    m = DCNv2(ROWS, DIM, ND, dev, mlp=(64,))
    ids, dense, labels = ofm.synthetic_batch(100, ROWS, ND, seed=7)
    with pytest.raises(MiError):
        m.forward(torch.from_numpy(ids).to(dev), torch.from_numpy(dense).to(dev))
