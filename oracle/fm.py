"""Numpy restatement of the factorisation machine (TEST INFRASTRUCTURE ONLY).

PARITY UNPINNED versus the reference: the reference contains no FM (its only
model is the linear colour network, r_model.py:24-140).  This follows the
published second-order FM (Rendle 2010) with the O(T*D) identity
    sum_{f<f'} <v_f, v_f'> = 0.5 * sum_k [ (sum_f v_fk)^2 - sum_f v_fk^2 ]
plus first-order weights, a linear term over the dense features and a bias;
loss = mean binary cross-entropy with logits.  It is vectorised numpy in the
gather formulation (a B x 10M one-hot is infeasible), used both as the parity
checker and as the timed CPU baseline.
"""
import numpy as np

from . import embedding as emb


def forward(V, w1, table_offsets, ids, dense, wd, bias, dtype=np.float64):
    """Returns (logit[B], S[B,D], rows[B,T])."""
    toffs = np.asarray(table_offsets, dtype=np.int64)
    rows = np.asarray(ids, dtype=np.int64) + toffs[:-1][None, :]
    v = np.asarray(V)[rows].astype(dtype)          # [B,T,D] gather
    S = v.sum(axis=1)
    pair = 0.5 * ((S * S).sum(axis=1) - (v * v).sum(axis=(1, 2)))
    lin = np.asarray(w1)[rows].astype(dtype).sum(axis=1)
    if dense is not None:
        lin = lin + np.asarray(dense, dtype=dtype) @ np.asarray(wd, dtype=dtype)
    return pair + lin + dtype(bias), S, rows


def bce_with_logits(z, y):
    per = np.maximum(z, 0) - z * y + np.log1p(np.exp(-np.abs(z)))
    sig = np.where(z >= 0, 1.0 / (1.0 + np.exp(-np.abs(z))), np.exp(-np.abs(z)) / (1.0 + np.exp(-np.abs(z))))
    return per, sig


def grads(V, table_offsets, ids, dense, g, S, dtype=np.float64):
    """Row gradients via the same sorted order as the device path.
    dV[r] = sum_i g[b_i]*(S[b_i] - V[r]);  dw1[r] = sum_i g[b_i]."""
    B, T = np.asarray(ids).shape
    sorted_rows, perm, unique_rows, seg = emb.prepare(table_offsets, T, ids, bag_len=1)
    b_of = perm // T
    gs = np.asarray(g, dtype=dtype)[b_of]
    contrib = gs[:, None] * np.asarray(S, dtype=dtype)[b_of]
    starts = seg[:-1]
    gsum = np.add.reduceat(gs, starts) if len(starts) else np.zeros(0, dtype)
    acc = np.add.reduceat(contrib, starts, axis=0) if len(starts) else np.zeros((0, S.shape[1]), dtype)
    dV = acc - np.asarray(V)[unique_rows].astype(dtype) * gsum[:, None]
    dwd = np.asarray(dense, dtype=dtype).T @ np.asarray(g, dtype=dtype) if dense is not None else np.zeros(0, dtype)
    dbias = np.asarray(g, dtype=dtype).sum()
    return unique_rows, dV, gsum, dwd, dbias


class State(object):
    """Parameters + optimizer state of one FM."""

    def __init__(self, rows_per_table, dim, n_dense, seed=0, dtype=np.float32, opt='adam_tf'):
        rng = np.random.default_rng(seed)
        self.toffs = np.concatenate([[0], np.cumsum(rows_per_table)]).astype(np.int64)
        total = int(self.toffs[-1])
        self.V = (rng.standard_normal((total, dim)) * 0.05).astype(dtype)
        self.w1 = (rng.standard_normal(total) * 0.05).astype(dtype)
        self.wd = (rng.standard_normal(n_dense) * 0.05).astype(dtype)
        self.bias = dtype(0.0)
        self.opt = opt
        self.t = 0
        if opt == 'adam_tf':
            self.mV, self.vV = np.zeros_like(self.V), np.zeros_like(self.V)
            self.mw, self.vw = np.zeros_like(self.w1), np.zeros_like(self.w1)
        elif opt == 'adagrad_rowwise':
            self.aV, self.aw = np.zeros(total, dtype), np.zeros(total, dtype)
        self.md, self.vd = np.zeros(n_dense + 1, dtype), np.zeros(n_dense + 1, dtype)


def train_step(st, ids, dense, labels, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float64):
    """One FM training step; sparse (touched rows only) update of V and w1, TensorFlow-form
    Adam on the dense side.  Returns (mean loss, logits)."""
    B = np.asarray(ids).shape[0]
    z, S, _ = forward(st.V, st.w1, st.toffs, ids, dense, st.wd, st.bias, dtype)
    per, sig = bce_with_logits(z, np.asarray(labels, dtype=dtype))
    g = (sig - labels) / B
    uniq, dV, dw1, dwd, dbias = grads(st.V, st.toffs, ids, dense, g, S, dtype)
    st.t += 1
    lr_t, omb1, omb2 = emb.tf_adam_scalars(lr, beta1, beta2, st.t)
    wdt = st.V.dtype
    if st.opt == 'sgd':
        st.V[uniq] -= (lr * dV).astype(wdt)
        st.w1[uniq] -= (lr * dw1).astype(wdt)
    elif st.opt == 'adagrad_rowwise':
        st.aV[uniq] += (dV * dV).mean(axis=1).astype(wdt)
        st.V[uniq] -= (lr * dV / (np.sqrt(st.aV[uniq].astype(dtype)) + eps)[:, None]).astype(wdt)
        st.aw[uniq] += (dw1 * dw1).astype(wdt)
        st.w1[uniq] -= (lr * dw1 / (np.sqrt(st.aw[uniq].astype(dtype)) + eps)).astype(wdt)
    else:
        for w, m, v, gr in ((st.V, st.mV, st.vV, dV), (st.w1, st.mw, st.vw, dw1)):
            nw, mm, vv = emb.adam_tf_update(w[uniq].astype(dtype), m[uniq].astype(dtype), v[uniq].astype(dtype),
                                            gr, lr_t, omb1, omb2, eps)
            m[uniq], v[uniq], w[uniq] = mm.astype(wdt), vv.astype(wdt), nw.astype(wdt)
    gd = np.concatenate([dwd, [dbias]])
    dp = np.concatenate([st.wd, [st.bias]]).astype(dtype)
    dp, md, vd = emb.adam_tf_update(dp, st.md.astype(dtype), st.vd.astype(dtype), gd, lr_t, omb1, omb2, eps)
    st.md, st.vd = md.astype(wdt), vd.astype(wdt)
    st.wd = dp[:-1].astype(wdt)
    st.bias = wdt.type(dp[-1])
    return per.mean(), z


CRITEO_KAGGLE_CARDINALITIES = [1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593,
                               3194, 27, 14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105,
                               142572]


def synthetic_batch(B, rows_per_table, n_dense=13, seed=0, dist='uniform', alpha=1.05, idx_dtype=np.int64):
    """Criteo-shaped synthetic batch: one id per field (uniform or Zipf(alpha) ranks),
    N(0,1) dense features, Bernoulli(0.25) labels."""
    rng = np.random.default_rng(seed)
    T = len(rows_per_table)
    ids = np.empty((B, T), dtype=idx_dtype)
    for t, n in enumerate(rows_per_table):
        if dist == 'uniform':
            ids[:, t] = rng.integers(0, n, size=B)
        else:
            # inverse-CDF sampling of a bounded Zipf via the continuous approximation
            u = rng.random(B)
            if abs(alpha - 1.0) < 1e-9:
                r = np.exp(u * np.log(n + 1.0))
            else:
                a = 1.0 - alpha
                r = ((u * ((n + 1.0) ** a - 1.0)) + 1.0) ** (1.0 / a)
            ids[:, t] = np.clip(np.floor(r).astype(np.int64) - 1, 0, n - 1)
    dense = rng.standard_normal((B, n_dense)).astype(np.float32)
    labels = (rng.random(B) < 0.25).astype(np.float32)
    return ids, dense, labels


def train_step_mt(st, ids, dense, labels, pool, n_workers, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
                  dtype=np.float64):
    """train_step spread over the threads of ``pool`` (a concurrent.futures.ThreadPoolExecutor with
    ``n_workers`` threads; numpy releases the interpreter lock inside its array loops): the forward
    over contiguous shards of the samples, the sparse backward + update over the TABLES (the rows of
    two tables are disjoint, and sorting a table's id column with a stable sort is the global stable
    sort by (row, position) restricted to that table, so every run is summed in the order train_step
    sums it).  Same results as train_step; it exists so that the CPU baseline can use every core."""
    ids = np.asarray(ids)
    B, T = ids.shape
    toffs = np.asarray(st.toffs, dtype=np.int64)
    bounds = np.linspace(0, B, n_workers + 1).astype(np.int64)

    def fwd(i):
        lo, hi = int(bounds[i]), int(bounds[i + 1])
        if lo == hi:
            return np.zeros(0, dtype), np.zeros((0, st.V.shape[1]), dtype)
        z, S, _ = forward(st.V, st.w1, toffs, ids[lo:hi], None if dense is None else dense[lo:hi], st.wd, st.bias, dtype)
        return z, S

    parts = list(pool.map(fwd, range(n_workers)))
    z = np.concatenate([p[0] for p in parts])
    S = np.concatenate([p[1] for p in parts])
    per, sig = bce_with_logits(z, np.asarray(labels, dtype=dtype))
    g = ((sig - labels) / B).astype(dtype)
    st.t += 1
    lr_t, omb1, omb2 = emb.tf_adam_scalars(lr, beta1, beta2, st.t)
    wdt = st.V.dtype

    def bwd(t):
        col = ids[:, t].astype(np.int64)
        order = np.argsort(col, kind='stable')
        uniq_ids, start = np.unique(col[order], return_index=True)
        uniq = toffs[t] + uniq_ids
        gs = g[order]
        gsum = np.add.reduceat(gs, start)
        acc = np.add.reduceat(gs[:, None] * S[order], start, axis=0)
        dV = acc - st.V[uniq].astype(dtype) * gsum[:, None]
        dw1 = gsum
        if st.opt == 'sgd':
            st.V[uniq] -= (lr * dV).astype(wdt)
            st.w1[uniq] -= (lr * dw1).astype(wdt)
        elif st.opt == 'adagrad_rowwise':
            st.aV[uniq] += (dV * dV).mean(axis=1).astype(wdt)
            st.V[uniq] -= (lr * dV / (np.sqrt(st.aV[uniq].astype(dtype)) + eps)[:, None]).astype(wdt)
            st.aw[uniq] += (dw1 * dw1).astype(wdt)
            st.w1[uniq] -= (lr * dw1 / (np.sqrt(st.aw[uniq].astype(dtype)) + eps)).astype(wdt)
        else:
            for w, m, v, gr in ((st.V, st.mV, st.vV, dV), (st.w1, st.mw, st.vw, dw1)):
                nw, mm, vv = emb.adam_tf_update(w[uniq].astype(dtype), m[uniq].astype(dtype), v[uniq].astype(dtype),
                                                gr, lr_t, omb1, omb2, eps)
                m[uniq], v[uniq], w[uniq] = mm.astype(wdt), vv.astype(wdt), nw.astype(wdt)

    list(pool.map(bwd, range(T)))
    dwd = np.asarray(dense, dtype=dtype).T @ g if dense is not None else np.zeros(0, dtype)
    gd = np.concatenate([dwd, [g.sum()]])
    dp = np.concatenate([st.wd, [st.bias]]).astype(dtype)
    dp, md, vd = emb.adam_tf_update(dp, st.md.astype(dtype), st.vd.astype(dtype), gd, lr_t, omb1, omb2, eps)
    st.md, st.vd = md.astype(wdt), vd.astype(wdt)
    st.wd = dp[:-1].astype(wdt)
    st.bias = wdt.type(dp[-1])
    return per.mean(), z
