"""Numpy restatement of bag pooling and of the sort-by-row sparse backward
(TEST INFRASTRUCTURE ONLY).

Bag pooling is the gather identity of the reference's first layers:
``n_hot(x) @ W == sum_j W[x_j]`` (r_model.py:68-80 feeding :53-54) and a
negative id contributes nothing (r_model.py:82-96).  The sparse backward has
no reference counterpart (TensorFlow forms the dense ``X^T dY``,
model_trainer.py:157-158); its oracle here is numpy's stable argsort /
unique, which the device artefacts must match bit for bit, and
``dense_grad`` must equal ``X^T dY`` computed from materialised one-hots.
"""
import numpy as np


def bag_slices(indices, offsets, bag_len, n_bags):
    flat = np.asarray(indices).reshape(-1)
    for i in range(n_bags):
        if offsets is not None:
            yield i, flat[int(offsets[i]):int(offsets[i + 1])]
        else:
            yield i, flat[i * bag_len:(i + 1) * bag_len]


def embag_fwd(weights, table_offsets, T, indices, offsets=None, bag_len=None, mode='sum', bias=None,
              dtype=np.float64):
    """out[b, t*dim:(t+1)*dim] = pool over valid ids of bag b*T+t; sequential summation."""
    weights = np.asarray(weights)
    dim = weights.shape[1]
    flat = np.asarray(indices).reshape(-1)
    if offsets is None:
        if bag_len is None:
            bag_len = np.asarray(indices).shape[-1] if np.asarray(indices).ndim > 1 else 1
        n_bags = flat.size // max(bag_len, 1)
    else:
        n_bags = len(offsets) - 1
    out = np.zeros((n_bags // T, T * dim), dtype=dtype)
    for i, ids in bag_slices(indices, offsets, bag_len, n_bags):
        t = i % T
        base = int(table_offsets[t])
        rows = int(table_offsets[t + 1]) - base
        acc = np.zeros(dim, dtype=dtype)
        cnt = 0
        for v in ids:
            v = int(v)
            if 0 <= v < rows:
                acc += weights[base + v].astype(dtype)
                cnt += 1
        if mode == 'mean' and cnt > 1:
            acc /= cnt
        if bias is not None:
            acc += np.asarray(bias, dtype=dtype)[t * dim:(t + 1) * dim]
        out[i // T, t * dim:(t + 1) * dim] = acc
    return out


def entry_keys(table_offsets, T, indices, offsets=None, bag_len=None):
    """Global row of every entry (sentinel = total_rows for padding / invalid), entry->bag map."""
    flat = np.asarray(indices).reshape(-1).astype(np.int64)
    n = flat.size
    if offsets is None:
        if bag_len is None:
            bag_len = np.asarray(indices).shape[-1] if np.asarray(indices).ndim > 1 else 1
        bag = np.arange(n) // max(bag_len, 1)
    else:
        offsets = np.asarray(offsets)
        bag = np.searchsorted(offsets, np.arange(n), side='right') - 1
    t = bag % T
    toffs = np.asarray(table_offsets, dtype=np.int64)
    rows = toffs[t + 1] - toffs[t]
    total = int(toffs[-1])
    valid = (flat >= 0) & (flat < rows)
    keys = np.where(valid, toffs[t] + flat, total)
    return keys, bag, total


def prepare(table_offsets, T, indices, offsets=None, bag_len=None):
    """(sorted_rows, perm, unique_rows, seg_offsets) restricted to valid entries."""
    keys, bag, total = entry_keys(table_offsets, T, indices, offsets, bag_len)
    perm = np.argsort(keys, kind='stable')
    sorted_rows = keys[perm]
    n_valid = int((sorted_rows != total).sum())
    sorted_rows, perm = sorted_rows[:n_valid], perm[:n_valid]
    unique_rows, first = np.unique(sorted_rows, return_index=True)
    seg_offsets = np.concatenate([first, [n_valid]]).astype(np.int32)
    return sorted_rows, perm.astype(np.int32), unique_rows, seg_offsets


def row_grads(table_offsets, T, indices, grad_out, dim, offsets=None, bag_len=None, mode='sum',
              dtype=np.float64):
    """dict row -> summed gradient (sequential in sorted order)."""
    keys, bag, total = entry_keys(table_offsets, T, indices, offsets, bag_len)
    grad_out = np.asarray(grad_out, dtype=dtype)
    counts = np.bincount(bag[keys != total], minlength=bag.max() + 1 if bag.size else 0)
    out = {}
    order = np.argsort(keys, kind='stable')
    for p in order:
        k = int(keys[p])
        if k == total:
            break
        b, t = int(bag[p]) // T, int(bag[p]) % T
        g = grad_out[b, t * dim:(t + 1) * dim]
        if mode == 'mean' and counts[bag[p]] > 1:
            g = g / counts[bag[p]]
        out[k] = out.get(k, np.zeros(dim, dtype=dtype)) + g
    return out


def dense_grad(total_rows, dim, grads, dtype=np.float64):
    dW = np.zeros((total_rows, dim), dtype=dtype)
    for r, g in grads.items():
        dW[r] = g
    return dW


def apply_sgd(w, grads, lr):
    for r, g in grads.items():
        w[r] -= lr * g


def apply_adagrad_rowwise(w, state, grads, lr, eps):
    for r, g in grads.items():
        state[r] += np.mean(g * g)
        w[r] -= lr * g / (np.sqrt(state[r]) + eps)


def tf_adam_scalars(lr, beta1, beta2, t):
    """The scalars of TensorFlow's ApplyAdam kernel as its fp32 arithmetic yields them:
    beta1_power / beta2_power are fp32 variables multiplied by beta once per step,
    alpha = lr*sqrt(1-beta2_power)/(1-beta1_power) and (1-beta) are formed in fp32.
    ((1 - 0.999f) differs from 0.001 by 1.3e-5 relative: not negligible at the 1e-5 bar.)
    Returns (alpha, 1-beta1, 1-beta2) as python floats holding fp32 values."""
    f = np.float32
    b1p, b2p = f(1.0), f(1.0)
    for _ in range(int(t)):
        b1p, b2p = f(b1p * f(beta1)), f(b2p * f(beta2))
    alpha = f(f(lr) * np.sqrt(f(1.0) - b2p) / (f(1.0) - b1p))
    return float(alpha), float(f(1.0) - f(beta1)), float(f(1.0) - f(beta2))


def adam_lr_t(lr, beta1, beta2, t):
    return tf_adam_scalars(lr, beta1, beta2, t)[0]


def adam_tf_update(w, m, v, g, alpha, omb1, omb2, eps):
    """ApplyAdam's three statements: m += (g-m)(1-b1); v += (g^2-v)(1-b2);
    w -= m*alpha/(sqrt(v)+eps).  Returns the new (w, m, v)."""
    m = m + (g - m) * omb1
    v = v + (g * g - v) * omb2
    return w - (m * alpha) / (np.sqrt(v) + eps), m, v


def apply_adam_tf(w, m, v, grads, lr_t, beta1, beta2, eps, dense=False):
    """TensorFlow-form Adam; ``dense`` also decays/moves rows absent from ``grads``."""
    rows = range(w.shape[0]) if dense else grads.keys()
    zero = np.zeros(w.shape[1:], dtype=w.dtype)
    omb1 = float(np.float32(1.0) - np.float32(beta1))
    omb2 = float(np.float32(1.0) - np.float32(beta2))
    for r in rows:
        w[r], m[r], v[r] = adam_tf_update(w[r], m[r], v[r], grads.get(r, zero), lr_t, omb1, omb2, eps)


def step_mt(W, state, m, v, table_offsets, T, ids, grad_out, opt, pool, workers, lr=0.01, eps=1e-8, lr_t=1e-3,
            beta1=0.9, beta2=0.999, out=None):
    """One embedding-bag step on the host at full speed: the same arithmetic as embag_fwd + row_grads +
    apply_* (sum pooling, fixed-width bags, no padding ids), vectorised per table and spread over ``workers``
    threads of ``pool`` (a table is independent of every other; numpy's gathers, sorts and reductions release
    the interpreter lock).  Per table: forward = gather + sum over the bag; backward = stable argsort of the
    table's ids, segment sums of the bags' gradient rows by np.add.reduceat, the optimizer on the distinct
    rows.  This is what bench.py times as the CPU baseline of the embedding workloads; tests hold it to the
    per-row functions above.  Returns the pooled output [B, T*D]."""
    W = np.asarray(W)
    ids = np.asarray(ids)
    B = ids.shape[0]
    L = ids.shape[2] if ids.ndim == 3 else 1
    ids = ids.reshape(B, T, L)
    D = W.shape[1]
    if out is None:
        out = np.empty((B, T * D), dtype=W.dtype)
    omb1 = np.float32(1.0) - np.float32(beta1)
    omb2 = np.float32(1.0) - np.float32(beta2)

    def table(t):
        base = int(table_offsets[t])
        idt = ids[:, t, :]
        rows = W[base + idt.reshape(-1)].reshape(B, L, D)
        out[:, t * D:(t + 1) * D] = rows.sum(axis=1) if L > 1 else rows[:, 0, :]
        keys = idt.reshape(-1)
        order = np.argsort(keys, kind='stable')
        sk = keys[order]
        starts = np.flatnonzero(np.concatenate([[True], sk[1:] != sk[:-1]]))
        g = grad_out[:, t * D:(t + 1) * D]
        gsum = np.add.reduceat(g[order // L], starts, axis=0)
        r = base + sk[starts]
        if opt == 'sgd':
            W[r] -= W.dtype.type(lr) * gsum
        elif opt == 'adagrad_rowwise':
            st = state[r] + (gsum * gsum).mean(axis=1)
            state[r] = st
            W[r] -= (W.dtype.type(lr) / (np.sqrt(st) + W.dtype.type(eps)))[:, None] * gsum
        else:
            mi = m[r] + (gsum - m[r]) * omb1
            vi = v[r] + (gsum * gsum - v[r]) * omb2
            m[r], v[r] = mi, vi
            W[r] -= mi * W.dtype.type(lr_t) / (np.sqrt(vi) + W.dtype.type(eps))

    list(pool.map(table, range(T))) if workers > 1 else [table(t) for t in range(T)]
    return out
