"""DIN local activation unit + a linear output layer, restated in numpy.

TEST INFRASTRUCTURE ONLY - nothing under recommendation_amd/ may import this.

No DIN exists in the reference (its nearest analogue is the 6-id bag with uniform weights,
recommendation/code/r_model.py:68-80).  This restates the unit of the DIN paper on one item table:
  a_j    = MLP([q, k_j, q - k_j, q * k_j])      4D -> h1 -> h2 -> 1, ReLU hidden layers
  pooled = sum_{j < len} a_j k_j                 (no softmax)
  logit  = Wh . [pooled | q] + bh,  BCE loss
``storage='bf16'`` rounds the tensors the device path stores in bfloat16 where it rounds them
(see oracle/dcn.py for why gradients are compared that way).  ``factored=True`` evaluates the first
layer as  q (Wa+Wc)^T + k (Wb-Wc)^T + (q*k) Wd^T  (W1 = [Wa | Wb | Wc | Wd]): the same function and
the same gradients in exact arithmetic, but a different set of stored tensors ([k | q*k] rows,
the combined weights), which is what the device's factored path rounds.  Parity against the reference itself
is unpinned: there is nothing to compare with.
"""
import numpy as np

from . import embedding as emb
from . import fm as ofm
from .dcn import _keep, bf16_round


class State(object):
    def __init__(self, rows, dim=64, h1=80, h2=40, seed=0, dtype=np.float32):
        rng = np.random.default_rng(seed)
        self.rows, self.dim, self.h1, self.h2 = rows, dim, h1, h2
        self.E = (rng.standard_normal((rows, dim)) * 0.3).astype(dtype)
        self.W1 = (rng.standard_normal((h1, 4 * dim)) * np.sqrt(2.0 / (4 * dim))).astype(dtype)
        self.b1 = (rng.standard_normal(h1) * 0.05).astype(dtype)
        self.W2 = (rng.standard_normal((h2, h1)) * np.sqrt(2.0 / h1)).astype(dtype)
        self.b2 = (rng.standard_normal(h2) * 0.05).astype(dtype)
        self.w3 = (rng.standard_normal(h2) / np.sqrt(h2)).astype(dtype)
        self.b3 = np.zeros(1, dtype)
        self.Wh = (rng.standard_normal(2 * dim) / np.sqrt(2 * dim)).astype(dtype)
        self.bh = np.zeros(1, dtype)
        self.t = 0
        self.acc = np.zeros(rows, dtype)
        self.adam = {}

    def dense_params(self):
        return {'lau/W1': self.W1, 'lau/b1': self.b1, 'lau/W2': self.W2, 'lau/b2': self.b2, 'lau/w3': self.w3,
                'lau/b3': self.b3, 'out/W': self.Wh, 'out/b': self.bh}


def synthetic_batch(B, L, rows, seed=0):
    """history ids [B, L] int64, candidate ids [B], lengths ~ U[1, L] int32, labels."""
    rng = np.random.default_rng(seed)
    hist = rng.integers(0, rows, size=(B, L)).astype(np.int64)
    qid = rng.integers(0, rows, size=B).astype(np.int64)
    lens = rng.integers(1, L + 1, size=B).astype(np.int32)
    labels = (rng.random(B) < 0.3).astype(np.float32)
    return hist, qid, lens, labels


def _w1_blocks(st, dtype):
    D = st.dim
    W = st.W1.astype(dtype)
    return W[:, :D], W[:, D:2 * D], W[:, 2 * D:3 * D], W[:, 3 * D:]


def forward(st, hist, qid, lens, dtype=np.float64, storage=None, keep=False, factored=False):
    r = bf16_round if storage == 'bf16' else _keep
    hist, qid = np.asarray(hist, np.int64), np.asarray(qid, np.int64)
    B, L = hist.shape
    D = st.dim
    mask = (np.arange(L)[None, :] < np.asarray(lens)[:, None]) & (hist >= 0)
    q = st.E[qid].astype(dtype)
    k = st.E[np.where(mask, hist, 0)].astype(dtype) * mask[:, :, None]
    qb = np.broadcast_to(q[:, None, :], k.shape)
    if factored:
        Wa, Wb, Wc, Wd = _w1_blocks(st, dtype)
        A = r(np.concatenate([k, qb * k], axis=2) * mask[:, :, None]).reshape(B * L, 2 * D)
        Wf = np.concatenate([r(Wb - Wc), r(Wd)], axis=1)              # the bf16 copy the contraction reads
        rowbias = q @ (Wa + Wc).T + st.b1.astype(dtype)                # fp32 on the device, per sample
        H1 = r(np.maximum(A @ Wf.T + np.repeat(rowbias, L, axis=0), 0))
    else:
        A = r(np.concatenate([qb, k, qb - k, qb * k], axis=2) * mask[:, :, None]).reshape(B * L, 4 * D)
        H1 = r(np.maximum(A @ st.W1.T.astype(dtype) + st.b1.astype(dtype), 0))
    H2 = r(np.maximum(H1 @ st.W2.T.astype(dtype) + st.b2.astype(dtype), 0))
    a = (H2 @ st.w3.astype(dtype) + dtype(st.b3[0])).reshape(B, L)
    kr = (A[:, :D] if factored else A[:, D:2 * D]).reshape(B, L, D)   # the k block as stored
    pooled = ((a * mask)[:, :, None] * kr).sum(1)
    feat = np.concatenate([pooled, q], axis=1)
    z = feat @ st.Wh.astype(dtype) + dtype(st.bh[0])
    if keep:
        return z, dict(mask=mask, q=q, A=A, H1=H1, H2=H2, a=a, kr=kr, pooled=pooled, feat=feat)
    return z


def grads(st, hist, qid, lens, labels, dtype=np.float64, storage=None, factored=False):
    """Mean BCE loss, gradients of the dense parameters by name, and of the item table as
    (unique rows, summed gradient rows)."""
    r = bf16_round if storage == 'bf16' else _keep
    z, k = forward(st, hist, qid, lens, dtype, storage, keep=True, factored=factored)
    hist, qid = np.asarray(hist, np.int64), np.asarray(qid, np.int64)
    B, L = hist.shape
    D = st.dim
    y = np.asarray(labels, dtype)
    per, sig = ofm.bce_with_logits(z, y)
    g = (sig - y) / B
    out = {'out/W': k['feat'].T @ g, 'out/b': np.array([g.sum()])}
    dfeat = np.outer(g, st.Wh.astype(dtype))
    dpooled, dq = dfeat[:, :D], dfeat[:, D:].copy()
    mask, A, H1, H2, a, kr = k['mask'], k['A'], k['H1'], k['H2'], k['a'], k['kr']
    da = ((dpooled[:, None, :] * kr).sum(2) * mask).reshape(B * L)
    out['lau/w3'] = H2.T @ da
    out['lau/b3'] = np.array([da.sum()])
    dH2 = r(np.outer(da, st.w3.astype(dtype)) * (H2 > 0))
    out['lau/W2'] = dH2.T @ H1
    out['lau/b2'] = dH2.sum(0)
    dZ1 = r((dH2 @ st.W2.astype(dtype)) * (H1 > 0))
    out['lau/b1'] = dZ1.sum(0)
    m3 = mask[:, :, None]
    if factored:
        Wa, Wb, Wc, Wd = _w1_blocks(st, dtype)
        S = dZ1.reshape(B, L, -1).sum(1)                               # per-sample sums
        gWa, gWf = S.T @ k['q'], dZ1.T @ A                             # gWf = [dZ^T K | dZ^T (Q*K)]
        gWb, gWd = gWf[:, :D], gWf[:, D:]
        out['lau/W1'] = np.concatenate([gWa, gWb, gWa - gWb, gWd], axis=1)
        dA = r(dZ1 @ np.concatenate([r(Wb - Wc), r(Wd)], axis=1)).reshape(B, L, 2 * D)
        gk, gp = dA[:, :, :D], dA[:, :, D:]
        dk = ((a * mask)[:, :, None] * dpooled[:, None, :] + gk + k['q'][:, None, :] * gp) * m3
        dq += (kr * gp * m3).sum(1) + S @ (Wa + Wc)
    else:
        out['lau/W1'] = dZ1.T @ A
        dA = r(dZ1 @ st.W1.astype(dtype)).reshape(B, L, 4 * D)
        gq, gk, gd, gp = (dA[:, :, i * D:(i + 1) * D] for i in range(4))
        qr = A[:, :D].reshape(B, L, D)
        dk = ((a * mask)[:, :, None] * dpooled[:, None, :] + gk - gd + qr * gp) * m3
        dq += ((gq + gd + kr * gp) * m3).sum(1)
    rows = np.concatenate([np.where(mask, hist, -1), qid[:, None]], axis=1).reshape(-1)
    gr = np.concatenate([dk, dq[:, None, :]], axis=1).reshape(-1, D)
    live = rows >= 0
    rows, gr = rows[live], gr[live]
    order = np.argsort(rows, kind='stable')
    uniq, start = np.unique(rows[order], return_index=True)
    end = np.append(start[1:], rows.size)
    dE = np.stack([gr[order[lo:hi]].sum(0) for lo, hi in zip(start, end)])
    return per.mean(), out, uniq, dE, dict(z=z, g=g, **k)


def train_step(st, hist, qid, lens, labels, lr=1e-3, emb_lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, emb_eps=1e-8,
               dtype=np.float64, storage=None, factored=False):
    loss, gr, uniq, dE, _ = grads(st, hist, qid, lens, labels, dtype, storage, factored)
    st.t += 1
    alpha, omb1, omb2 = emb.tf_adam_scalars(lr, beta1, beta2, st.t)
    for name, p in st.dense_params().items():
        m, v = st.adam.get(name, (np.zeros_like(p), np.zeros_like(p)))
        w, mm, vv = emb.adam_tf_update(p.astype(dtype), m.astype(dtype), v.astype(dtype), gr[name], alpha, omb1, omb2,
                                       eps)
        p[...] = w.astype(p.dtype)
        st.adam[name] = (mm.astype(p.dtype), vv.astype(p.dtype))
    st.acc[uniq] += (dE * dE).mean(axis=1).astype(st.acc.dtype)
    st.E[uniq] -= (emb_lr * dE / (np.sqrt(st.acc[uniq].astype(dtype)) + emb_eps)[:, None]).astype(st.E.dtype)
    return loss
