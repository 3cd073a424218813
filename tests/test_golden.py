"""CPU: the oracle reproduces the committed golden vectors (guards oracle drift)."""
import os

import numpy as np

from oracle import colour_model as cm
from oracle import fm as ofm

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_colour():
    z = np.load(os.path.join(GOLD, 'colour_model_b32.npz'))
    init = {k[5:]: z[k] for k in z.files if k.startswith('init/')}
    init = {n: init[n] for n in cm.init_params(0)}            # creation order
    batch = {k[6:]: z[k] for k in z.files if k.startswith('batch/')}
    return z, init, batch


def test_colour_model_golden():
    z, init, batch = load_colour()
    params = {n: p.astype(np.float64) for n, p in init.items()}
    opt = cm.AdamTF(params, lr=1e-4)
    for step in range(2):
        acc, _, top_idx, loss, logits = cm.train_step(params, opt, batch)
        assert np.allclose(logits, z['step%d/logits' % step], rtol=1e-12, atol=0)
        assert abs(loss - z['step%d/loss' % step]) < 1e-12 and acc == z['step%d/acc' % step]
        assert np.array_equal(top_idx, z['step%d/top_idx' % step])
    for n in params:
        assert np.allclose(params[n], z['final/' + n], rtol=1e-12, atol=0)


def test_fm_golden():
    z = np.load(os.path.join(GOLD, 'fm_b64.npz'))
    st = ofm.State(z['rows'].tolist(), 16, 13, seed=7, dtype=np.float32, opt='adagrad_rowwise')
    assert np.array_equal(st.V, z['V']) and np.array_equal(st.w1, z['w1'])
    logit, S, _ = ofm.forward(st.V, st.w1, st.toffs, z['ids'], z['dense'], st.wd, st.bias)
    assert np.allclose(logit, z['logit'], rtol=1e-12) and np.allclose(S, z['S'], rtol=1e-12)
    loss, _ = ofm.train_step(st, z['ids'], z['dense'], z['labels'], lr=0.01)
    assert abs(loss - z['loss']) < 1e-12 and np.array_equal(st.V, z['V_after'])
