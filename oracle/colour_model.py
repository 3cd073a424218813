"""Numpy restatement of the reference colour model (TEST INFRASTRUCTURE ONLY).

Follows ``color_project/dependencies/python/recommendation/code/r_model.py``
op for op, *including* the materialisation of the dense one-hot / n-hot
matrices and the dense matmul that the reference performs, so that the
embedding-gather formulation used by the HIP path is checked against the
reference's own O(B*V*D) algorithm and not against itself.  A second,
independent gather formulation (``forward_gather``) is kept beside it; tests
assert both agree.

TensorFlow semantics restated here (TF 1.15 documented behaviour; TensorFlow
itself is not available, so these are pinned by hand-derived known answers):

* ``tf.sparse_to_dense(..., validate_indices=True)``: indices must be in
  bounds, lexicographically sorted and not repeated, else
  ``InvalidArgumentError``.
* ``tf.nn.top_k``: descending by value, ties resolved to the lower index.
* ``tf.nn.softmax_cross_entropy_with_logits``: loss_i = -sum_c y_ic *
  log_softmax(z)_ic and the fused kernel's gradient is ``softmax(z) - y``
  whatever the row sum of y is (an all-zero label row still back-propagates
  ``softmax(z)``).
* ``tf.train.AdamOptimizer`` (the ApplyAdam kernel): alpha = lr*sqrt(1-b2^t)/(1-b1^t);
  m += (g-m)(1-b1); v += (g^2-v)(1-b2); theta -= m*alpha/(sqrt(v)+eps); applied densely to
  every variable, the scalars formed in fp32 as the kernel forms them.
"""
from collections import OrderedDict

import numpy as np


class InvalidArgumentError(ValueError):
    """Stands in for ``tf.errors.InvalidArgumentError``."""


class OutOfRangeError(StopIteration):
    """Stands in for ``tf.errors.OutOfRangeError`` (end of a dataset)."""


DEFAULT_UNITS = [[8], [4], [8, 3, 3], [8, 3, 3], 8, 4]  # r_model.py:31


# ---------------------------------------------------------------------------
# encoders (r_model.py:60-96)
# ---------------------------------------------------------------------------
def sparse_to_dense(indices, shape, value=1.0, default=0.0, dtype=np.float32):
    """``tf.sparse_to_dense`` with ``validate_indices=True`` (r_model.py:65,76,90)."""
    indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
    out = np.full(shape, default, dtype=dtype)
    prev = None
    for k in range(indices.shape[0]):
        r, c = int(indices[k, 0]), int(indices[k, 1])
        if not (0 <= r < shape[0] and 0 <= c < shape[1]):
            raise InvalidArgumentError(
                "indices[%d] = [%d,%d] is out of bounds: need 0 <= index < [%d,%d]"
                % (k, r, c, shape[0], shape[1]))
        if prev is not None:
            if (r, c) == prev:
                raise InvalidArgumentError("indices[%d] = [%d,%d] is repeated" % (k, r, c))
            if (r, c) < prev:
                raise InvalidArgumentError("indices[%d] = [%d,%d] is out of order" % (k, r, c))
        prev = (r, c)
        out[r, c] = value
    return out


def input_to_one_hot(x, size, dtype=np.float32):
    """r_model.py:60-66."""
    x = np.asarray(x, dtype=np.int32).reshape(-1)
    batch_size = x.size
    concat = np.stack([np.arange(batch_size, dtype=np.int64), x.astype(np.int64)], axis=1)
    return sparse_to_dense(concat, (batch_size, size), dtype=dtype)


def input_to_n_hot(x, size, recommend_num, dtype=np.float32):
    """r_model.py:68-80.  Bag width is the model attribute, not the tensor shape."""
    x = np.asarray(x, dtype=np.int32)
    batch_size = x.shape[0]
    if x.ndim != 2 or x.shape[1] != recommend_num:
        raise InvalidArgumentError("n-hot input must be [batch, %d], got %r" % (recommend_num, x.shape))
    rows = np.broadcast_to(np.arange(batch_size, dtype=np.int64)[:, None], (batch_size, recommend_num))
    concat = np.stack([rows, x.astype(np.int64)], axis=-1).reshape(-1, 2)
    return sparse_to_dense(concat, (batch_size, size), dtype=dtype)


def input_to_one_hot_plus(x, size, dtype=np.float32):
    """r_model.py:82-96: negative id -> all-zero row (partition / stitch)."""
    x = np.asarray(x, dtype=np.int32).reshape(-1)
    mask = x >= 0
    pos = x[mask]
    one_hot = input_to_one_hot(pos, size, dtype=dtype) if pos.size else np.zeros((0, size), dtype)
    res = np.zeros((x.size, size), dtype=dtype)
    res[np.nonzero(mask)[0]] = one_hot
    return res


def features(batch, colour_count=128, recommend_num=6, user_count=100, country_count=20,
             dtype=np.float32):
    """r_model.py:42-50."""
    return OrderedDict([
        ('user', input_to_one_hot(batch['user'], user_count, dtype)),
        ('country', input_to_one_hot(batch['country'], country_count, dtype)),
        ('recommend_colours_1', input_to_n_hot(batch['recommend_colours_1'], colour_count, recommend_num, dtype)),
        ('click_colour_1', input_to_one_hot_plus(batch['click_colour_1'], colour_count, dtype)),
        ('recommend_colours_2', input_to_n_hot(batch['recommend_colours_2'], colour_count, recommend_num, dtype)),
        ('click_colour_2', input_to_one_hot_plus(batch['click_colour_2'], colour_count, dtype)),
    ])


# ---------------------------------------------------------------------------
# parameters (r_model.py:31-40, 52-58)
# ---------------------------------------------------------------------------
def layer_shapes(colour_count=128, user_count=100, country_count=20, units=None):
    """Kernel shapes in TF creation order ``dense`` .. ``dense_10`` (r_model.py:32-39)."""
    units = DEFAULT_UNITS if units is None else units
    shapes = []

    def tower(n_in, widths):
        for w in widths:
            shapes.append((n_in, w))
            n_in = w
        return n_in

    u = tower(user_count, units[0])
    c = tower(country_count, units[1])
    r1 = tower(2 * colour_count, units[2])
    r2 = tower(2 * colour_count, units[3])
    tower(u + c + r1 + r2, [units[4], units[5], colour_count])
    return shapes


def tower_sizes(units=None):
    units = DEFAULT_UNITS if units is None else units
    return [len(units[0]), len(units[1]), len(units[2]), len(units[3]), 3]


def var_name(k, what):
    return ('dense' if k == 0 else 'dense_%d' % k) + '/' + what


def init_params(seed=0, colour_count=128, user_count=100, country_count=20, units=None,
                dtype=np.float32):
    """First layer of each tower: truncated normal(0,1); others: uniform[0,1); bias 0
    (r_model.py:53-57).  The random stream is numpy's, not TensorFlow's."""
    rng = np.random.default_rng(seed)
    shapes = layer_shapes(colour_count, user_count, country_count, units)
    firsts = set(np.cumsum([0] + tower_sizes(units))[:-1].tolist())
    params = OrderedDict()
    for k, (n_in, n_out) in enumerate(shapes):
        if k in firsts:
            w = rng.standard_normal((n_in, n_out))
            bad = np.abs(w) > 2.0
            while bad.any():
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2.0
        else:
            w = rng.random((n_in, n_out))
        params[var_name(k, 'kernel')] = w.astype(dtype)
        params[var_name(k, 'bias')] = np.zeros((n_out,), dtype=dtype)
    return params


# ---------------------------------------------------------------------------
# forward (r_model.py:31-40, 52-58)
# ---------------------------------------------------------------------------
def forward(fs, params, units=None, keep=False):
    """Dense formulation, exactly the reference's graph.  ``keep`` returns the
    per-layer inputs needed by ``backward``."""
    sizes = tower_sizes(units)
    k = 0
    acts = []  # (layer index, input matrix)

    def network(x, n_layers):
        nonlocal k
        for _ in range(n_layers):
            acts.append((k, x))
            x = x @ params[var_name(k, 'kernel')] + params[var_name(k, 'bias')]
            k += 1
        return x

    user_layer = network(fs['user'], sizes[0])
    country_layer = network(fs['country'], sizes[1])
    r_1 = network(np.concatenate([fs['recommend_colours_1'], fs['click_colour_1']], axis=1), sizes[2])
    r_2 = network(np.concatenate([fs['recommend_colours_2'], fs['click_colour_2']], axis=1), sizes[3])
    concat = np.concatenate([user_layer, country_layer, r_1, r_2], axis=1)
    logits = network(concat, sizes[4])
    if keep:
        widths = [user_layer.shape[1], country_layer.shape[1], r_1.shape[1], r_2.shape[1]]
        return logits, acts, widths
    return logits


def forward_gather(batch, params, colour_count=128, units=None):
    """Independent formulation: first layers as embedding row gathers + sum pooling,
    never building a one-hot.  one_hot(x) @ W + b == W[x] + b."""
    sizes = tower_sizes(units)
    starts = np.cumsum([0] + sizes)
    dtype = params[var_name(0, 'kernel')].dtype

    def rest(x, k0, k1):
        for k in range(k0, k1):
            x = x @ params[var_name(k, 'kernel')] + params[var_name(k, 'bias')]
        return x

    def first(k, rows_list):
        w = params[var_name(k, 'kernel')]
        acc = np.zeros((len(rows_list), w.shape[1]), dtype=dtype)
        for i, rows in enumerate(rows_list):
            for r in rows:
                acc[i] += w[r]
        return acc + params[var_name(k, 'bias')]

    user = np.asarray(batch['user']).reshape(-1)
    country = np.asarray(batch['country']).reshape(-1)
    u = rest(first(starts[0], [[int(v)] for v in user]), starts[0] + 1, starts[1])
    c = rest(first(starts[1], [[int(v)] for v in country]), starts[1] + 1, starts[2])
    towers = []
    for t, (rk, ck) in enumerate([('recommend_colours_1', 'click_colour_1'),
                                  ('recommend_colours_2', 'click_colour_2')]):
        rec = np.asarray(batch[rk])
        clk = np.asarray(batch[ck]).reshape(-1)
        rows = []
        for i in range(rec.shape[0]):
            ids = [int(v) for v in rec[i]]
            if clk[i] >= 0:
                ids.append(colour_count + int(clk[i]))
            rows.append(ids)
        k0 = starts[2 + t]
        towers.append(rest(first(k0, rows), k0 + 1, starts[3 + t]))
    concat = np.concatenate([u, c] + towers, axis=1)
    return rest(concat, starts[4], starts[5])


# ---------------------------------------------------------------------------
# heads (r_model.py:98-119, 132-140)
# ---------------------------------------------------------------------------
def softmax(z):
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return e / e.sum(axis=1, keepdims=True)


def log_softmax(z):
    z = z - z.max(axis=1, keepdims=True)
    return z - np.log(np.exp(z).sum(axis=1, keepdims=True))


def top_k(values, k):
    """``tf.nn.top_k``: descending, ties -> lower index first. Returns (values, indices)."""
    values = np.asarray(values)
    order = np.argsort(-values, axis=1, kind='stable')[:, :k]
    return np.take_along_axis(values, order, axis=1), order.astype(np.int32)


def output(logits, recommend_num=6):
    """r_model.py:98-101: returns (indices, values) - swapped w.r.t. top_k."""
    vals, idx = top_k(softmax(logits), recommend_num)
    return idx, vals


def top_one_output(logits):
    """r_model.py:103-106: squeeze -> scalars when the batch is 1."""
    vals, idx = top_k(softmax(logits), 1)
    return np.squeeze(idx), np.squeeze(vals)


def loss(logits, labels_dense):
    """r_model.py:116-119: mean over the full batch of -sum(y * log_softmax(z))."""
    per_row = -(labels_dense * log_softmax(logits)).sum(axis=1)
    return per_row.mean(dtype=logits.dtype)


def loss_grad_logits(logits, labels_dense):
    """Gradient of ``loss`` as TF's fused xent kernel produces it: (softmax - y) / B."""
    return (softmax(logits) - labels_dense) / logits.dtype.type(logits.shape[0])


def accuracy(labels, predictions, element_count):
    """r_model.py:132-140 (tile / reshape / transpose / equal / matmul with ones)."""
    labels = np.asarray(labels, dtype=np.int32).reshape(-1)
    predictions = np.asarray(predictions, dtype=np.int32)
    batch_size = labels.size
    a = np.tile(labels, element_count).reshape(element_count, batch_size).T
    a = (a == predictions).astype(np.float32)
    per_row = a @ np.ones((element_count, 1), dtype=np.float32)
    return np.float32(per_row.sum() / np.float32(batch_size)), per_row


# ---------------------------------------------------------------------------
# backward + optimizer (model_trainer.py:138-158; local_train_utils.py:33-63)
# ---------------------------------------------------------------------------
def backward(params, acts, widths, dlogits, units=None):
    """Dense gradients of every kernel/bias: dW = X^T dY, db = sum dY, dX = dY W^T.
    The gradient of a first-layer kernel is dense [V, D] with exact zeros on rows
    no sample touched - what TF's MatMul gradient yields for a one-hot X."""
    sizes = tower_sizes(units)
    starts = np.cumsum([0] + sizes)
    inputs = dict(acts)
    grads = OrderedDict()

    def back(d, k_hi, k_lo):
        for k in range(k_hi - 1, k_lo - 1, -1):
            x = inputs[k]
            grads[var_name(k, 'kernel')] = x.T @ d
            grads[var_name(k, 'bias')] = d.sum(axis=0)
            d = d @ params[var_name(k, 'kernel')].T
        return d

    dconcat = back(dlogits, starts[5], starts[4])
    col = 0
    for t in range(4):
        back(dconcat[:, col:col + widths[t]], starts[t + 1], starts[t])
        col += widths[t]
    return OrderedDict((n, grads[n]) for n in params)


class AdamTF(object):
    """``tf.train.AdamOptimizer`` (TF1 form), dense on every variable."""

    def __init__(self, params, lr=1e-4, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.m = OrderedDict((n, np.zeros_like(p)) for n, p in params.items())
        self.v = OrderedDict((n, np.zeros_like(p)) for n, p in params.items())
        self.t = 0

    def apply(self, params, grads):
        """TensorFlow's ApplyAdam kernel: m += (g-m)(1-b1); v += (g^2-v)(1-b2);
        var -= m*alpha/(sqrt(v)+eps), with alpha and (1-beta) as its fp32 arithmetic forms them."""
        from .embedding import adam_tf_update, tf_adam_scalars
        self.t += 1
        alpha, omb1, omb2 = tf_adam_scalars(self.lr, self.beta1, self.beta2, self.t)
        for n, p in params.items():
            g = grads[n].astype(p.dtype)
            dt = p.dtype.type
            new_p, self.m[n], self.v[n] = adam_tf_update(p, self.m[n], self.v[n], g, dt(alpha), dt(omb1),
                                                         dt(omb2), dt(self.eps))
            p[...] = new_p


def train_step(params, opt, batch, colour_count=128, recommend_num=6, user_count=100,
               country_count=20, units=None):
    """One pass of the hot loop (local_train_utils.py:42-63): returns what the
    reference fetches - (acc, labels, top_indices, loss) - and updates ``params``."""
    dtype = params[var_name(0, 'kernel')].dtype
    label_dense = input_to_one_hot_plus(batch['label'], colour_count, dtype)
    fs = features(batch, colour_count, recommend_num, user_count, country_count, dtype)
    logits, acts, widths = forward(fs, params, units, keep=True)
    top_idx, _ = output(logits, recommend_num)
    acc, _ = accuracy(batch['label'], top_idx, recommend_num)
    ls = loss(logits, label_dense)
    grads = backward(params, acts, widths, loss_grad_logits(logits, label_dense), units)
    opt.apply(params, grads)
    return acc, np.asarray(batch['label']), top_idx, ls, logits


# ---------------------------------------------------------------------------
# record grammar (r_model.py:143-202)
# ---------------------------------------------------------------------------
_BAG_DEFAULT = '0,0,0,0,0,0'


def _to_int32(text):
    v = int(text)
    if not -2 ** 31 <= v < 2 ** 31:
        raise InvalidArgumentError("Field is not a valid int32: %r" % text)
    return v


def parse_record(line, with_label=True):
    """``tf.decode_csv(value, record_defaults, field_delim=' ')`` + ``multiply_split``
    (r_model.py:146-153 / 162-170).  Fields split on ONE space; an empty field
    takes its default (0 or '0,0,0,0,0,0')."""
    n_fields = 7 if with_label else 6
    fields = line.rstrip('\r\n').split(' ')
    if len(fields) != n_fields:
        raise InvalidArgumentError("Expect %d fields but have %d in record" % (n_fields, len(fields)))
    bag_cols = (2, 4)
    out = []
    for k, f in enumerate(fields):
        if k in bag_cols:
            text = f if f != '' else _BAG_DEFAULT
            out.append(np.array([_to_int32(t) for t in text.split(',') if t != ''], dtype=np.int32))
        else:
            try:
                out.append(np.int32(_to_int32(f) if f != '' else 0))
            except ValueError:
                raise InvalidArgumentError("Field %d in record is not a valid int32: %r" % (k, f))
    return tuple(out)


_KEYS = ('user', 'country', 'recommend_colours_1', 'click_colour_1',
         'recommend_colours_2', 'click_colour_2', 'label')


def batches_from_lines(lines, batch_size, with_label=True):
    """ds.map(parse_csv).batch(batch_size): final partial batch is kept."""
    rows = []
    for ln in lines:
        if ln.strip('\r\n') == '':
            continue
        rows.append(parse_record(ln, with_label))
        if len(rows) == batch_size:
            yield _stack(rows, with_label)
            rows = []
    if rows:
        yield _stack(rows, with_label)


def _stack(rows, with_label):
    keys = _KEYS if with_label else _KEYS[:-1]
    return OrderedDict((k, np.stack([r[i] for r in rows]).astype(np.int32)) for i, k in enumerate(keys))


def shard_files(files, parallelism=None, index=None):
    """r_model.py:178-181."""
    files = list(files) if isinstance(files, (list, tuple)) else [files]
    if parallelism is not None and index is not None:
        return files[index::parallelism]
    return files


# ---------------------------------------------------------------------------
# synthetic input distribution (data.py:38-64, 81-105, 136-160)
# ---------------------------------------------------------------------------
def synthetic_batch(batch_size, seed=20211207, colour_count=128, recommend_num=6, user_count=100,
                    country_count=20, with_label=True):
    """uid ~ U[0,user_count); country = fixed random map[uid]; rec = sorted
    ``recommend_num``-subset of range(colour_count); click uniform over rec + {-1};
    label in [0,colour_count) or -1.  The reference seeds from the clock; this uses a
    fixed seed."""
    rng = np.random.default_rng(seed)
    country_of = rng.integers(0, country_count, size=user_count)
    user = rng.integers(0, user_count, size=batch_size).astype(np.int32)
    batch = OrderedDict(user=user, country=country_of[user].astype(np.int32))
    for t in (1, 2):
        rec = np.stack([np.sort(rng.choice(colour_count, size=recommend_num, replace=False))
                        for _ in range(batch_size)]).astype(np.int32)
        pick = rng.integers(0, recommend_num + 1, size=batch_size)
        click = np.where(pick < recommend_num,
                         np.take_along_axis(rec, np.minimum(pick, recommend_num - 1)[:, None], 1)[:, 0],
                         -1).astype(np.int32)
        batch['recommend_colours_%d' % t] = rec
        batch['click_colour_%d' % t] = click
    if with_label:
        lab = rng.integers(0, colour_count, size=batch_size)
        batch['label'] = np.where(rng.random(batch_size) < 0.3, -1, lab).astype(np.int32)
    return batch


def format_record(batch, i, with_label=True):
    """One text record in the grammar ``parse_record`` reads (data.py:99-105, 139-160)."""
    parts = [str(int(batch['user'][i])), str(int(batch['country'][i])),
             ','.join(str(int(v)) for v in batch['recommend_colours_1'][i]),
             str(int(batch['click_colour_1'][i])),
             ','.join(str(int(v)) for v in batch['recommend_colours_2'][i]),
             str(int(batch['click_colour_2'][i]))]
    if with_label:
        parts.append(str(int(batch['label'][i])))
    return ' '.join(parts)
