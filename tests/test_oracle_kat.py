"""Pins the numpy oracle (CPU only).

The reference ships no tests and TensorFlow cannot be imported here, so the oracle is pinned
by hand-derived known answers: the one snippet in the reference's own tree that states an
expected result (recommendation/test.py:19-28, the accuracy construction) and values worked
out by hand from recommendation/code/r_model.py.  Parity versus a live TensorFlow run is
UNPINNED and says so in oracle/__init__.py.
"""
import math

import numpy as np
import pytest

from oracle import colour_model as cm
from oracle import embedding as oe
from oracle import fm as ofm


def test_kat1_accuracy_snippet_from_reference_scratch_file():
    # recommendation/test.py:19-28
    labels = [5, 2, 9, -1]
    preds = [[9, 3, 5, 2, 1], [8, 9, 0, 6, 5], [1, 9, 3, 4, 5], [1, 2, 3, 4, 5]]
    acc, per_row = cm.accuracy(labels, preds, 5)
    assert per_row.tolist() == [[1.0], [0.0], [1.0], [0.0]]
    assert acc == 0.5


def test_kat2_one_hot_plus():
    # r_model.py:82-96
    assert cm.input_to_one_hot_plus([3, -1, 0], 4).tolist() == [[0, 0, 0, 1], [0, 0, 0, 0], [1, 0, 0, 0]]
    assert cm.input_to_one_hot_plus([-1, -1], 3).tolist() == [[0, 0, 0], [0, 0, 0]]


def test_kat3_n_hot_and_its_validation():
    # r_model.py:68-80 with tf.sparse_to_dense validate_indices=True
    m = cm.input_to_n_hot([[0, 2, 5, 7, 9, 11]], 128, 6)
    assert m.shape == (1, 128) and m.sum() == 6 and np.nonzero(m[0])[0].tolist() == [0, 2, 5, 7, 9, 11]
    with pytest.raises(cm.InvalidArgumentError):
        cm.input_to_n_hot([[2, 0, 5, 7, 9, 11]], 128, 6)       # out of order
    with pytest.raises(cm.InvalidArgumentError):
        cm.input_to_n_hot([[1, 1, 5, 7, 9, 11]], 128, 6)       # repeated
    with pytest.raises(cm.InvalidArgumentError):
        cm.input_to_n_hot([[1, 2, 5, 7, 9, 128]], 128, 6)      # out of bounds
    with pytest.raises(cm.InvalidArgumentError):
        cm.input_to_one_hot([3, -1], 4)                        # plain one-hot has no padding id
    with pytest.raises(cm.InvalidArgumentError):
        cm.input_to_one_hot_plus([4], 4)


def test_kat4_gather_identity_between_the_two_formulations():
    """network(one_hot(x)) == W[x]+b and network(concat(n_hot(r), one_hot_plus(c))) ==
    sum_j W[r_j] + (c>=0 ? W[128+c] : 0) + b: the dense graph and the gather graph agree."""
    params = cm.init_params(seed=3, dtype=np.float64)
    rng = np.random.default_rng(0)
    for n in params:
        if n.endswith('bias'):
            params[n] = rng.standard_normal(params[n].shape)
    batch = cm.synthetic_batch(64, seed=9)
    fs = cm.features(batch, dtype=np.float64)
    dense = cm.forward(fs, params)
    gather = cm.forward_gather(batch, params)
    assert np.abs(dense - gather).max() <= 1e-12 * np.abs(dense).max()
    # and the one-liner itself
    W, b = params['dense/kernel'], params['dense/bias']
    x = batch['user']
    assert np.allclose(cm.input_to_one_hot(x, 100, np.float64) @ W + b, W[x] + b, rtol=0, atol=1e-15)


def test_kat5_loss_at_zero_logits():
    # r_model.py:116-119: ln(128) for a valid label, 0 for label -1, mean over the FULL batch
    labels = np.array([7, -1, 0, 127], dtype=np.int32)
    y = cm.input_to_one_hot_plus(labels, 128, np.float64)
    z = np.zeros((4, 128))
    assert abs(math.log(128) - 4.852030263919617) < 1e-15
    assert abs(cm.loss(z, y) - 3 * math.log(128) / 4) < 1e-12
    # TensorFlow's fused kernel: gradient is softmax - labels, also for the all-zero label row
    g = cm.loss_grad_logits(z, y)
    assert np.allclose(g[1], (1.0 / 128) / 4) and abs(g[0, 7] - (1.0 / 128 - 1.0) / 4) < 1e-15


def test_kat6_layer_shapes_and_names():
    # r_model.py:31-40: 11 layers, 5,910 parameters, TF names dense .. dense_10
    shapes = cm.layer_shapes()
    assert shapes == [(100, 8), (20, 4), (256, 8), (8, 3), (3, 3), (256, 8), (8, 3), (3, 3), (18, 8), (8, 4),
                      (4, 128)]
    params = cm.init_params(seed=0)
    assert sum(p.size for p in params.values()) == 5910
    assert list(params)[:4] == ['dense/kernel', 'dense/bias', 'dense_1/kernel', 'dense_1/bias']
    assert list(params)[-1] == 'dense_10/bias'
    first = [0, 1, 2, 5, 8]
    for k in range(11):
        w = params[cm.var_name(k, 'kernel')]
        if k in first:
            assert np.abs(w).max() <= 2.0 and w.min() < 0          # truncated normal
        else:
            assert 0.0 <= w.min() and w.max() < 1.0                 # uniform [0,1)
        assert not params[cm.var_name(k, 'bias')].any()


def test_top_k_order_and_ties():
    vals, idx = cm.top_k(np.array([[0.1, 0.5, 0.5, 0.2], [1.0, 1.0, 1.0, 1.0]]), 3)
    assert idx.tolist() == [[1, 2, 3], [0, 1, 2]]
    idx2, vals2 = cm.output(np.zeros((2, 128)), 6)
    assert idx2.tolist() == [[0, 1, 2, 3, 4, 5]] * 2 and np.allclose(vals2, 1.0 / 128)
    i1, v1 = cm.top_one_output(np.zeros((1, 128)))
    assert i1.shape == () and v1.shape == ()                        # squeeze -> scalars at batch 1


def test_adam_tf_first_steps_by_hand():
    """t=1: m = 0.1 g, v = 0.001 g^2, lr_t = lr*sqrt(0.001)/0.1 -> step = lr*g/(|g| + eps*sqrt(1000))...
    written out with TensorFlow's epsilon placement (outside the bias correction)."""
    p = {'w': np.array([1.0, -2.0, 0.5])}
    g = {'w': np.array([0.5, -0.25, 0.0])}
    opt = cm.AdamTF(p, lr=1e-4)
    opt.apply(p, g)
    # the kernel forms alpha and (1-beta) in fp32: within 1e-5 of the exact values, not equal
    alpha, omb1, omb2 = oe.tf_adam_scalars(1e-4, 0.9, 0.999, 1)
    assert abs(alpha / (1e-4 * math.sqrt(1 - 0.999) / (1 - 0.9)) - 1) < 1e-5
    assert omb1 == float(np.float32(1) - np.float32(0.9)) and abs(omb2 / 0.001 - 1) < 2e-5
    want = np.array([1.0, -2.0, 0.5]) - alpha * (omb1 * g['w']) / (np.sqrt(omb2 * g['w'] ** 2) + 1e-8)
    assert np.allclose(p['w'], want, rtol=0, atol=1e-15)
    assert p['w'][2] == 0.5                       # zero gradient, zero state: no move at t=1
    opt.apply(p, {'w': np.zeros(3)})              # dense Adam: zero gradient still moves via m
    assert p['w'][0] < want[0] and p['w'][2] == 0.5


def test_backward_matches_finite_differences():
    params = cm.init_params(seed=1, dtype=np.float64)
    batch = cm.synthetic_batch(16, seed=2)
    y = cm.input_to_one_hot_plus(batch['label'], 128, np.float64)
    fs = cm.features(batch, dtype=np.float64)

    def f():
        return cm.loss(cm.forward(fs, params), y)

    logits, acts, widths = cm.forward(fs, params, keep=True)
    grads = cm.backward(params, acts, widths, cm.loss_grad_logits(logits, y))
    rng = np.random.default_rng(0)
    for name in ['dense/kernel', 'dense_2/kernel', 'dense_4/bias', 'dense_8/kernel', 'dense_10/kernel']:
        p = params[name]
        for _ in range(4):
            i = tuple(rng.integers(0, s) for s in p.shape)
            if name == 'dense_2/kernel':
                i = (int(batch['recommend_colours_1'][0, 0]), i[1])   # a touched row
            old = p[i]
            p[i] = old + 1e-6
            up = f()
            p[i] = old - 1e-6
            dn = f()
            p[i] = old
            # all-zero label rows are where TF's gradient (softmax - y) departs from d(loss):
            # finite differences see 0 for them, so compare on the labelled rows' share
            fd = (up - dn) / 2e-6
            lab = batch['label'] >= 0
            logits2, acts2, widths2 = cm.forward(fs, params, keep=True)
            d = cm.loss_grad_logits(logits2, y)
            d[~lab] = 0.0
            g_true = cm.backward(params, acts2, widths2, d)[name][i]
            assert abs(fd - g_true) <= 1e-6 * max(1.0, abs(g_true))
    # untouched rows of a first-layer kernel get an exact zero
    untouched = sorted(set(range(100)) - set(batch['user'].tolist()))
    assert not grads['dense/kernel'][untouched].any()


def test_record_grammar_and_sharding():
    # r_model.py:143-202
    rec = cm.parse_record("12 3 1,5,9,20,77,100 9 0,2,4,6,8,10 -1 42\n")
    assert rec[0] == 12 and rec[1] == 3 and rec[2].tolist() == [1, 5, 9, 20, 77, 100]
    assert rec[3] == 9 and rec[4].tolist() == [0, 2, 4, 6, 8, 10] and rec[5] == -1 and rec[6] == 42
    assert all(r.dtype == np.int32 for r in rec)
    rec = cm.parse_record("  1,2,3,4,5,6    ", with_label=True)    # empty fields take their defaults
    assert rec[0] == 0 and rec[1] == 0 and rec[2].tolist() == [1, 2, 3, 4, 5, 6] and rec[3] == 0
    assert rec[4].tolist() == [0, 0, 0, 0, 0, 0] and rec[5] == 0 and rec[6] == 0
    with pytest.raises(cm.InvalidArgumentError):
        cm.parse_record("1 2 3", with_label=False)
    with pytest.raises(cm.InvalidArgumentError):
        cm.parse_record("1 2 1,2,3,4,5,6 x 1,2,3,4,5,6 0", with_label=False)
    assert cm.shard_files(['a', 'b', 'c', 'd', 'e'], 2, 0) == ['a', 'c', 'e']
    assert cm.shard_files(['a', 'b', 'c', 'd', 'e'], 2, 1) == ['b', 'd']
    assert cm.shard_files('a') == ['a']
    batch = cm.synthetic_batch(7, seed=1)
    lines = [cm.format_record(batch, i) for i in range(7)]
    got = list(cm.batches_from_lines(lines, 3))
    assert [b['user'].shape[0] for b in got] == [3, 3, 1]             # final partial batch kept
    assert np.array_equal(np.concatenate([b['recommend_colours_2'] for b in got]), batch['recommend_colours_2'])
    assert np.array_equal(np.concatenate([b['label'] for b in got]), batch['label'])


def test_synthetic_batch_satisfies_the_reference_contract():
    b = cm.synthetic_batch(500)
    cm.features(b)                                  # strictly increasing n-hot rows etc: must not raise
    for t in (1, 2):
        rec, click = b['recommend_colours_%d' % t], b['click_colour_%d' % t]
        assert ((click == -1) | (rec == click[:, None]).any(axis=1)).all()
    assert (b['label'] == -1).any() and (b['label'] >= 0).any()


def test_sparse_prepare_and_dense_grad_oracles_agree_with_one_hot_matmul():
    rng = np.random.default_rng(4)
    rows, dim, B, L = [5, 40], 3, 50, 4
    toffs = np.array([0, 5, 45])
    ids = np.stack([rng.integers(-1, r, size=(B, L)) for r in rows], axis=1)
    grad = rng.standard_normal((B, 2 * dim))
    s, perm, uniq, seg = oe.prepare(toffs, 2, ids, bag_len=L)
    assert (np.diff(s) >= 0).all() and np.array_equal(uniq, np.unique(s)) and seg[-1] == len(s)
    assert all((np.diff(perm[seg[i]:seg[i + 1]]) > 0).all() for i in range(len(uniq)))   # stable
    dW = oe.dense_grad(45, dim, oe.row_grads(toffs, 2, ids, grad, dim, bag_len=L))
    for t in range(2):
        X = np.zeros((B, rows[t]))
        for b in range(B):
            for v in ids[b, t]:
                if v >= 0:
                    X[b, v] += 1
        assert np.allclose(dW[toffs[t]:toffs[t + 1]], X.T @ grad[:, t * dim:(t + 1) * dim])


def test_fm_identity_against_explicit_pairwise_sum():
    rows = [7, 5, 3, 11]
    st = ofm.State(rows, 4, 2, seed=0, dtype=np.float64)
    ids, dense, labels = ofm.synthetic_batch(9, rows, 2, seed=1)
    z, S, r = ofm.forward(st.V, st.w1, st.toffs, ids, dense, st.wd, 0.25)
    for b in range(9):
        v = st.V[r[b]]
        pair = sum(v[i] @ v[j] for i in range(4) for j in range(i + 1, 4))
        want = pair + st.w1[r[b]].sum() + dense[b].astype(np.float64) @ st.wd + 0.25
        assert abs(z[b] - want) < 1e-12
    # gradient of the mean BCE wrt one embedding row by finite differences
    y = labels.astype(np.float64)

    def loss():
        zz, _, _ = ofm.forward(st.V, st.w1, st.toffs, ids, dense, st.wd, 0.25)
        return ofm.bce_with_logits(zz, y)[0].mean()

    per, sig = ofm.bce_with_logits(z, y)
    uniq, dV, dw1, dwd, dbias = ofm.grads(st.V, st.toffs, ids, dense, (sig - y) / 9, S)
    k = 2
    row = int(uniq[k])
    old = st.V[row, 1]
    st.V[row, 1] = old + 1e-6
    up = loss()
    st.V[row, 1] = old - 1e-6
    dn = loss()
    st.V[row, 1] = old
    assert abs((up - dn) / 2e-6 - dV[k, 1]) < 1e-8
