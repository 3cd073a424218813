"""Generates tests/golden/*.npz from the numpy oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The vectors are inputs and expected outputs only.
They come from the oracle, not from a TensorFlow run of the reference (TensorFlow is not
available offline and the reference ships no vectors of its own), so they guard against
drift of the oracle and give the GPU tests a fixed target; they do not pin the oracle to the
reference - the hand-derived known answers in tests/test_oracle_kat.py do that."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import colour_model as cm  # noqa: E402
from oracle import fm as ofm  # noqa: E402


def colour():
    params = cm.init_params(seed=20211207, dtype=np.float64)
    rng = np.random.default_rng(1)
    for n in params:
        if n.endswith('bias'):
            params[n] = rng.standard_normal(params[n].shape) * 0.1
    init = {n: p.astype(np.float32) for n, p in params.items()}
    params = {n: p.astype(np.float64) for n, p in init.items()}
    batch = cm.synthetic_batch(32, seed=20211207)
    opt = cm.AdamTF(params, lr=1e-4)
    out = {'init/' + n: p for n, p in init.items()}
    out.update({'batch/' + k: v for k, v in batch.items()})
    for step in range(2):
        acc, _, top_idx, loss, logits = cm.train_step(params, opt, batch)
        out['step%d/logits' % step] = logits
        out['step%d/loss' % step] = np.float64(loss)
        out['step%d/acc' % step] = np.float64(acc)
        out['step%d/top_idx' % step] = top_idx
    out.update({'final/' + n: p for n, p in params.items()})
    np.savez_compressed(os.path.join(HERE, 'colour_model_b32.npz'), **out)


def fm():
    rows = [50, 7, 300, 3, 1000, 24]
    st = ofm.State(rows, 16, 13, seed=7, dtype=np.float32, opt='adagrad_rowwise')
    ids, dense, labels = ofm.synthetic_batch(64, rows, 13, seed=8, dist='zipf')
    out = dict(rows=np.array(rows), V=st.V.copy(), w1=st.w1.copy(), wd=st.wd.copy(), ids=ids, dense=dense,
               labels=labels)
    z, S, _ = ofm.forward(st.V, st.w1, st.toffs, ids, dense, st.wd, st.bias)
    out['logit'], out['S'] = z, S
    loss, _ = ofm.train_step(st, ids, dense, labels, lr=0.01)
    out['loss'] = np.float64(loss)
    out['V_after'], out['w1_after'], out['wd_after'], out['bias_after'] = st.V, st.w1, st.wd, np.float64(st.bias)
    np.savez_compressed(os.path.join(HERE, 'fm_b64.npz'), **out)


if __name__ == '__main__':
    colour()
    fm()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith('.npz')))
