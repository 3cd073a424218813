"""DCN-v2 (full-rank cross layers + ReLU MLP + 1-wide output, BCE loss) restated in numpy.

TEST INFRASTRUCTURE ONLY - nothing under recommendation_amd/ may import this.

There is no DCN in the reference (recommendation/code/r_model.py:24-58 is a linear network);
this is the published DCN-v2 recurrence  x_{l+1} = x0 * (W_l x_l + b_l) + x_l  on the
Criteo-shaped input  x0 = [26 embeddings of dim 16 | 13 dense features]  that the benchmark
definition names.  Parity of the bf16 device path against this fp32/fp64 restatement is held to
1e-2; parity against the reference itself is unpinned (nothing to compare with).
"""
import numpy as np

from . import embedding as emb
from . import fm as ofm


def bf16_round(a):
    """Round to the nearest bfloat16 (ties to even), returned in the input's float type."""
    a = np.asarray(a)
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(a.dtype if a.dtype.kind == 'f' else np.float32)


def _keep(a):
    return a


class State(object):
    def __init__(self, rows_per_table, dim, n_dense, n_cross=3, mlp=(1024, 1024, 512), seed=0, dtype=np.float32):
        rng = np.random.default_rng(seed)
        self.toffs = np.concatenate([[0], np.cumsum(rows_per_table)]).astype(np.int64)
        self.T, self.dim, self.n_dense = len(rows_per_table), dim, n_dense
        total = int(self.toffs[-1])
        d = self.T * dim + n_dense
        self.d = d
        self.V = (rng.standard_normal((total, dim)) * 0.05).astype(dtype)
        self.Wc = [(rng.standard_normal((d, d)) / np.sqrt(d) * 0.5).astype(dtype) for _ in range(n_cross)]
        self.bc = [(rng.standard_normal(d) * 0.05).astype(dtype) for _ in range(n_cross)]
        widths = [d] + list(mlp)
        self.Wm = [(rng.standard_normal((widths[i + 1], widths[i])) * np.sqrt(2.0 / widths[i])).astype(dtype)
                   for i in range(len(mlp))]
        self.bm = [(rng.standard_normal(widths[i + 1]) * 0.05).astype(dtype) for i in range(len(mlp))]
        self.w_out = (rng.standard_normal(widths[-1]) / np.sqrt(widths[-1])).astype(dtype)
        self.b_out = np.zeros(1, dtype)
        self.t = 0
        self.acc = np.zeros(total, dtype)                       # row-wise Adagrad accumulator of V
        self.adam = {}                                          # name -> (m, v)

    def dense_params(self):
        out = {}
        for l, (W, b) in enumerate(zip(self.Wc, self.bc)):
            out['cross%d/W' % l], out['cross%d/b' % l] = W, b
        for i, (W, b) in enumerate(zip(self.Wm, self.bm)):
            out['mlp%d/W' % i], out['mlp%d/b' % i] = W, b
        out['out/w'], out['out/b'] = self.w_out, self.b_out
        return out


def forward(st, ids, dense, dtype=np.float64, keep=False, storage=None):
    """``storage='bf16'`` rounds every tensor the device path STORES (input row, u_l, x_l, h_i) to
    bfloat16 where the device rounds it, with exact arithmetic in between: what the bf16 path
    computes up to accumulation order, so that ReLU gates agree with the device's."""
    r = bf16_round if storage == 'bf16' else _keep
    ids = np.asarray(ids, np.int64)
    B = ids.shape[0]
    rows = ids + st.toffs[:-1][None, :]
    x0 = r(np.concatenate([st.V[rows].reshape(B, -1).astype(dtype), np.asarray(dense, dtype)], axis=1))
    xs, us = [x0], []
    for W, b in zip(st.Wc, st.bc):
        u = xs[-1] @ W.T.astype(dtype) + b.astype(dtype)
        us.append(r(u))                      # stored for the backward
        xs.append(r(x0 * u + xs[-1]))        # formed from the unrounded u
    hs = [xs[-1]]
    for W, b in zip(st.Wm, st.bm):
        hs.append(r(np.maximum(hs[-1] @ W.T.astype(dtype) + b.astype(dtype), 0)))
    z = hs[-1] @ st.w_out.astype(dtype) + dtype(st.b_out[0])
    if keep:
        return z, dict(rows=rows, xs=xs, us=us, hs=hs)
    return z


def grads(st, ids, dense, labels, dtype=np.float64, storage=None):
    """Mean BCE loss and the gradient of every parameter (dense ones by name, V as {row: grad}).
    ``storage='bf16'``: as in forward(), and the stored backward tensors (dz_i, dx_l, du_l) are
    rounded where the device rounds them."""
    r = bf16_round if storage == 'bf16' else _keep
    z, k = forward(st, ids, dense, dtype, keep=True, storage=storage)
    B = z.shape[0]
    y = np.asarray(labels, dtype)
    per, sig = ofm.bce_with_logits(z, y)
    g = (sig - y) / B
    out = {}
    hs, xs, us = k['hs'], k['xs'], k['us']
    out['out/w'] = hs[-1].T @ g
    out['out/b'] = np.array([g.sum()])
    dh = np.outer(g, st.w_out.astype(dtype))
    for i in reversed(range(len(st.Wm))):
        dz = r(dh * (hs[i + 1] > 0))
        out['mlp%d/W' % i] = dz.T @ hs[i]
        out['mlp%d/b' % i] = dz.sum(0)
        dh = dz @ st.Wm[i].astype(dtype)
    v = dh                                                      # d loss / d x_L, before it is stored
    x0 = xs[0]
    dx0 = np.zeros_like(x0)
    for l in reversed(range(len(st.Wc))):
        du = r(v * x0)
        dx0 += r(v) * us[l]                                     # d x0 is formed from the STORED dx_{l+1} and u_l
        out['cross%d/W' % l] = du.T @ xs[l]
        out['cross%d/b' % l] = du.sum(0)
        v = du @ st.Wc[l].astype(dtype) + r(v)                  # the residual is read back from its stored form
    dx0 += r(v)
    demb = dx0[:, :st.T * st.dim].reshape(B, st.T, st.dim)
    rows = k['rows']
    flat = rows.reshape(-1)
    order = np.argsort(flat, kind='stable')
    uniq, start = np.unique(flat[order], return_index=True)
    end = np.append(start[1:], flat.size)
    dV = np.stack([demb.reshape(-1, st.dim)[order[lo:hi]].sum(0) for lo, hi in zip(start, end)])
    return per.mean(), out, uniq, dV, dict(z=z, g=g, dx0=dx0, **k)


def train_step(st, ids, dense, labels, lr=1e-3, emb_lr=1e-2, beta1=0.9, beta2=0.999, eps=1e-8, emb_eps=1e-8,
               dtype=np.float64, storage=None):
    """TensorFlow-form Adam on every dense parameter, row-wise Adagrad on the touched rows of V."""
    loss, gr, uniq, dV, _ = grads(st, ids, dense, labels, dtype, storage)
    st.t += 1
    alpha, omb1, omb2 = emb.tf_adam_scalars(lr, beta1, beta2, st.t)
    for name, p in st.dense_params().items():
        m, v = st.adam.get(name, (np.zeros_like(p), np.zeros_like(p)))
        w, mm, vv = emb.adam_tf_update(p.astype(dtype), m.astype(dtype), v.astype(dtype), gr[name], alpha, omb1, omb2,
                                       eps)
        p[...] = w.astype(p.dtype)
        st.adam[name] = (mm.astype(p.dtype), vv.astype(p.dtype))
    st.acc[uniq] += (dV * dV).mean(axis=1).astype(st.acc.dtype)
    st.V[uniq] -= (emb_lr * dV / (np.sqrt(st.acc[uniq].astype(dtype)) + emb_eps)[:, None]).astype(st.V.dtype)
    return loss
