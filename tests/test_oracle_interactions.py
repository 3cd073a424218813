"""The numpy restatements of DCN-v2 and the DIN unit are themselves pinned: their analytic
gradients must agree with central finite differences of their own forward, and bf16_round must be
round-to-nearest-even."""
import numpy as np
import torch

from oracle import dcn as odcn
from oracle import din as odin
from oracle import fm as ofm


def _fd(loss_fn, arr, idx, h=1e-3):
    old = arr[idx]
    arr[idx] = old + h
    lp = loss_fn()
    arr[idx] = old - h
    lm = loss_fn()
    arr[idx] = old
    return (lp - lm) / (2 * h)


def test_dcn_gradients_match_finite_differences():
    rows = [50, 7, 120, 3]
    st = odcn.State(rows, 4, 3, n_cross=2, mlp=(16, 8), seed=1)
    ids, dense, labels = ofm.synthetic_batch(6, rows, 3, seed=2)
    _, gr, uniq, dV, _ = odcn.grads(st, ids, dense, labels)

    def loss():
        per, _ = ofm.bce_with_logits(odcn.forward(st, ids, dense), labels.astype(np.float64))
        return per.mean()

    for name, idx in [('cross0/W', (1, 2)), ('cross1/W', (0, 5)), ('cross1/b', (3,)), ('mlp0/W', (2, 5)),
                      ('mlp1/b', (1,)), ('out/w', (2,)), ('out/b', (0,))]:
        assert abs(_fd(loss, st.dense_params()[name], idx) - gr[name][idx]) <= 1e-5 + 1e-3 * abs(gr[name][idx]), name
    for i in (0, len(uniq) // 2, len(uniq) - 1):
        assert abs(_fd(loss, st.V, (uniq[i], 1)) - dV[i, 1]) <= 1e-5 + 1e-3 * abs(dV[i, 1])


def test_din_gradients_match_finite_differences():
    st = odin.State(50, dim=4, h1=6, h2=5, seed=1)
    hist, qid, lens, labels = odin.synthetic_batch(5, 7, 50, seed=2)
    hist[0, 0] = -1
    _, gr, uniq, dE, _ = odin.grads(st, hist, qid, lens, labels)

    def loss():
        per, _ = ofm.bce_with_logits(odin.forward(st, hist, qid, lens), labels.astype(np.float64))
        return per.mean()

    for name, idx in [('lau/W1', (1, 9)), ('lau/b1', (3,)), ('lau/W2', (2, 5)), ('lau/b2', (0,)), ('lau/w3', (1,)),
                      ('lau/b3', (0,)), ('out/W', (6,)), ('out/b', (0,))]:
        assert abs(_fd(loss, st.dense_params()[name], idx) - gr[name][idx]) <= 1e-5 + 1e-3 * abs(gr[name][idx]), name
    for i in (0, len(uniq) // 2, len(uniq) - 1):
        assert abs(_fd(loss, st.E, (uniq[i], 1)) - dE[i, 1]) <= 1e-5 + 1e-3 * abs(dE[i, 1])


def test_bf16_round_is_round_to_nearest_even():
    a = (np.random.default_rng(0).standard_normal(100000) * 3).astype(np.float32)
    a[:6] = [0, 1.00390625, 1.01171875, -2.5, 3e-40, 65504.0]
    want = torch.from_numpy(a).to(torch.bfloat16).float().numpy()
    assert np.array_equal(odcn.bf16_round(a), want)
    assert odcn.bf16_round(a.astype(np.float64)).dtype == np.float64


def test_din_factored_first_layer_is_the_same_function():
    """q (Wa+Wc)^T + k (Wb-Wc)^T + (q*k) Wd^T == [q | k | q-k | q*k] W1^T: loss and every gradient agree
    to rounding in exact arithmetic (the factored form only changes which tensors are stored)."""
    from oracle import din as odin
    st = odin.State(500, seed=1)
    hist, qid, lens, labels = odin.synthetic_batch(16, 12, 500, seed=2)
    hist[3, 1] = -1
    l0, g0, u0, d0, _ = odin.grads(st, hist, qid, lens, labels)
    l1, g1, u1, d1, _ = odin.grads(st, hist, qid, lens, labels, factored=True)
    assert abs(l0 - l1) <= 1e-12
    for k in g0:
        assert np.abs(g0[k] - g1[k]).max() <= 1e-12 * max(1.0, np.abs(g0[k]).max())
    assert np.array_equal(u0, u1) and np.abs(d0 - d1).max() <= 1e-12


def test_threaded_fm_step_equals_the_plain_step():
    """oracle.fm.train_step_mt (the all-cores CPU baseline) must give train_step's results."""
    import copy
    from concurrent.futures import ThreadPoolExecutor
    from oracle import fm as ofm
    rows = [50, 7, 3000, 3, 1000, 24]
    for opt in ('sgd', 'adagrad_rowwise', 'adam_tf'):
        a = ofm.State(rows, 16, 13, seed=1, opt=opt)
        b = copy.deepcopy(a)
        with ThreadPoolExecutor(3) as pool:
            for step in range(3):
                ids, dense, labels = ofm.synthetic_batch(500, rows, 13, seed=step, dist='zipf')
                la, za = ofm.train_step(a, ids, dense, labels, lr=1e-2)
                lb, zb = ofm.train_step_mt(b, ids, dense, labels, pool, 3, lr=1e-2)
                assert abs(la - lb) <= 1e-12 * abs(la) and np.allclose(za, zb, rtol=1e-12, atol=0)
        for name in ('V', 'w1', 'wd'):
            assert np.allclose(getattr(a, name), getattr(b, name), rtol=1e-6, atol=1e-9), (opt, name)
        assert abs(float(a.bias) - float(b.bias)) <= 1e-7


def test_threaded_embedding_step_equals_the_per_row_oracle():
    """oracle.embedding.step_mt (what bench.py times on the host cores) against the per-row functions, for the
    three optimizers, one id per bag and multi-hot bags."""
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from oracle import embedding as oe
    rng = np.random.default_rng(8)
    T, D, B = 3, 8, 50
    rows = [40, 7, 200]
    toffs = np.concatenate([[0], np.cumsum(rows)]).astype(np.int64)
    for L in (1, 5):
        ids = np.stack([rng.integers(0, r, size=(B, L)) for r in rows], axis=1).astype(np.int64)
        grad = rng.standard_normal((B, T * D))
        for opt in ('sgd', 'adagrad_rowwise', 'adam_tf'):
            W0 = rng.standard_normal((int(toffs[-1]), D))
            Wa, Wb = W0.copy(), W0.copy()
            sa, sb = np.full(len(W0), 0.1), np.full(len(W0), 0.1)
            ma, va = rng.standard_normal(W0.shape) * 0.1, rng.random(W0.shape) * 0.1
            mb, vb = ma.copy(), va.copy()
            want_out = oe.embag_fwd(Wa, toffs, T, ids, bag_len=L)
            grads = oe.row_grads(toffs, T, ids, grad, D, bag_len=L)
            if opt == 'sgd':
                oe.apply_sgd(Wa, grads, 0.05)
            elif opt == 'adagrad_rowwise':
                oe.apply_adagrad_rowwise(Wa, sa, grads, 0.05, 1e-8)
            else:
                oe.apply_adam_tf(Wa, ma, va, grads, 1e-2, 0.9, 0.999, 1e-8)
            with ThreadPoolExecutor(3) as pool:
                got_out = oe.step_mt(Wb, sb, mb, vb, toffs, T, ids, grad, opt, pool, 3, lr=0.05, eps=1e-8, lr_t=1e-2)
            assert np.allclose(got_out, want_out, rtol=1e-12, atol=1e-12)
            assert np.allclose(Wb, Wa, rtol=1e-6, atol=1e-9), (opt, L)      # (1 - beta) is formed in fp32 by both
            if opt == 'adagrad_rowwise':
                assert np.allclose(sb, sa, rtol=1e-12)
            if opt == 'adam_tf':
                assert np.allclose(mb, ma, rtol=1e-6) and np.allclose(vb, va, rtol=1e-6)
