"""Drop-in for the reference's model file: same class, method names, argument order and
results as ``recommendation/code/r_model.py`` - ``RecommendationModel`` (:24-140) and
``Sample`` (:143-202) - with the arithmetic running in HIP kernels.

What changes underneath:

* ``input_to_one_hot`` / ``input_to_n_hot`` / ``input_to_one_hot_plus`` (:60-96) return a
  ``SparseFeature``: the ids plus the width of the one-hot matrix they stand for.  The
  dense matrix is only built (bit-exactly, on the device) when a caller asks for it with
  ``to_dense()`` - the loss does, for its label tensor.
* ``network`` (:52-58): when its input is a ``SparseFeature`` (or a concat of them, :34-37)
  the first ``tf.layers.dense`` becomes one embedding gather + sum pooling + bias launch,
  because one_hot(x) @ W == W[x] and n_hot(x) @ W == sum_j W[x_j]; a padding id (< 0)
  contributes a zero row exactly like the all-zero one-hot row.  Later layers are small
  fp32 linear kernels.
* gradients: the kernel of a first layer gets the same dense [V, D] gradient TensorFlow's
  MatMul gradient yields (exact zeros on untouched rows), built by the deterministic
  sort-by-row segmented sum, so ``AdamOptimizer`` reproduces tf.train.AdamOptimizer.
* eager instead of graph mode: methods take and return torch device tensors; invalid ids
  raise ``InvalidArgumentError`` from ``features`` instead of from ``session.run``.

Variables are created in the reference's order and carry TensorFlow's automatic names
``dense/kernel``, ``dense/bias``, ``dense_1/kernel`` ... ``dense_10/bias``.
"""
import logging
from collections import OrderedDict

import numpy as np
import torch

from . import ops
from .ops import InvalidArgumentError

logger = logging.getLogger(__name__)


class OutOfRangeError(StopIteration):
    """End of a dataset (``tf.errors.OutOfRangeError``)."""


class SparseFeature(object):
    """A one-hot / n-hot matrix [batch, size] kept as its ids [batch, width] (id < 0 = no entry)."""

    def __init__(self, ids, size):
        self.ids = ids if ids.dim() == 2 else ids.view(-1, 1)
        self.size = int(size)

    @property
    def shape(self):
        return (self.ids.shape[0], self.size)

    def to_dense(self):
        return ops.onehot_dense(self.ids, self.size)

    @staticmethod
    def concat(parts):
        """tf.concat(..., axis=1) of one-hot matrices == ids shifted by the widths before them."""
        if (1 < len(parts) <= ops.MI_MAX_ID_COLUMNS and len(set(p.ids.dtype for p in parts)) == 1 and
                all(p.ids.is_cuda for p in parts)):    # one launch on the device; host tensors: plain index arithmetic
            return SparseFeature(ops.concat_id_columns([(p.ids, p.size) for p in parts]), sum(p.size for p in parts))
        ids, base = [], 0
        for p in parts:
            ids.append(torch.where(p.ids >= 0, p.ids + base, p.ids) if base else p.ids)
            base += p.size
        return SparseFeature(torch.cat(ids, dim=1).contiguous(), base)


def concat(values, axis=1):
    """``tf.concat`` for the two kinds of value the model passes around."""
    if all(isinstance(v, SparseFeature) for v in values):
        assert axis == 1
        return SparseFeature.concat(values)
    values = [v.to_dense() if isinstance(v, SparseFeature) else v for v in values]
    return torch.cat(values, dim=axis)


# ----------------------------------------------------------------------------- autograd glue
class _GradSink(object):
    """Where a variable's gradient goes when it can simply be stored: its slice of the model's flat gradient
    buffer, written by the backward kernel itself.  The op's backward then hands autograd no gradient for the
    variable, which saves the temporary and the add autograd would launch per variable.  Storing equals
    accumulating only while the slice is still zero and nothing else will add to it, so a sink allows it when
    the variable entered exactly one op since the last finished backward (`uses`) and nothing was written to
    the slice since it was zeroed (`dirty`); every other case takes autograd's accumulating path."""
    __slots__ = ('view', 'uses', 'dirty')

    def __init__(self, view):
        self.view, self.uses, self.dirty = view, 0, False

    def take(self):
        """The buffer to store into, or None when the gradient has to be accumulated by autograd."""
        direct = self.uses == 1 and not self.dirty
        self.dirty = True
        return self.view if direct else None


def _take(sink):
    return sink.take() if sink is not None else None


class _EmbagDense(torch.autograd.Function):
    """first tf.layers.dense on a one-hot input: forward = gather+pool+bias, backward = the
    dense kernel gradient X^T dY (sorted segmented sum) and the bias gradient."""

    @staticmethod
    def forward(ctx, kernel, bias, ids, toffs, ksink=None, bsink=None):
        ctx.ids, ctx.toffs = ids, toffs
        ctx.shape = kernel.shape
        ctx.sinks = (ksink, bsink)
        return ops.embag_fwd(kernel, toffs, 1, ids, bag_len=ids.shape[1], bias=bias)

    @staticmethod
    def backward(ctx, dy):
        if dy.stride(-1) != 1:
            dy = dy.contiguous()          # a column slice of a wider gradient is read in place (row stride = its ld)
        rows, dim = ctx.shape
        # without run arrays: the sort is one launch (the table's entries sorted in the LDS of a CU or two) and the
        # segmented sum finds the runs itself - two launches where the compacted-run form takes ten
        plan = ops.SortedPlan(ctx.toffs, 1, rows, ctx.ids, bag_len=ctx.ids.shape[1], runs=False)
        gk, gb = _take(ctx.sinks[0]), _take(ctx.sinks[1])
        dW = ops.bwd_dense_grad(plan, dy, dim, rows, grad_ld=dy.stride(0), out=gk)
        db = ops.colsum(dy, out=gb)
        return (None if gk is not None else dW), (None if gb is not None else db), None, None, None, None


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kernel, bias, ksink=None, bsink=None):
        x = x.contiguous()
        ctx.save_for_backward(x, kernel)
        ctx.sinks = (ksink, bsink)
        return ops.linear_fwd(x, kernel, bias)

    @staticmethod
    def backward(ctx, dy):
        x, kernel = ctx.saved_tensors
        gk, gb = _take(ctx.sinks[0]), _take(ctx.sinks[1])
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        dx, dW, db = ops.linear_bwd(x, kernel, dy, need_dx=ctx.needs_input_grad[0], dW_out=gk, db_out=gb)
        return dx, (None if gk is not None else dW), (None if gb is not None else db), None, None


class _MeanSoftmaxXent(torch.autograd.Function):
    """reduce_mean(softmax_cross_entropy_with_logits): gradient is (softmax - labels)/B for
    every row, also rows whose label vector is all zero (TensorFlow's fused kernel)."""

    @staticmethod
    def forward(ctx, logits, labels):
        B = logits.shape[0]
        r = ops.softmax_heads(logits.contiguous(), labels=labels, grad_scale=1.0 / B, want_loss=True,
                              want_dlogits=True)
        ctx.save_for_backward(r['dlogits'])
        return ops.reduce_sum(r['loss_rows'], 1.0 / B).view(())

    @staticmethod
    def backward(ctx, g):
        (dlogits,) = ctx.saved_tensors
        return dlogits * g, None


# ----------------------------------------------------------------------------- the model
class RecommendationModel(object):
    def __init__(self, colour_count, recommend_num, user_count, country_count, device=None, seed=None,
                 validate=True):
        self.colour_count = colour_count
        self.recommend_num = recommend_num
        self.user_count = user_count
        self.country_count = country_count
        self.device = torch.device(device if device is not None else 'cuda:0')
        self.validate = validate
        self.check_every = 1
        self._feature_calls = 0
        self._in_graph = False           # set while a step is captured / replayed as a HIP graph: no host reads
        self._gen = torch.Generator(device='cpu')
        self._gen.manual_seed(0 if seed is None else int(seed))
        self._arena = torch.zeros(0, dtype=torch.float32, device=self.device)
        self._grads = None               # flat gradient buffer, laid out like the arena
        self._leaves = {}                # name -> autograd leaf sharing the arena's memory (its .grad a view of _grads)
        self._sinks = {}                 # name -> _GradSink over the same view of _grads
        self._leaves_of = None           # data_ptr of the arena the leaves were made for
        self._want_grad = False
        self._slots = OrderedDict()      # name -> (offset, shape)
        self._cursor = 0                 # index of the next tf.layers.dense call
        self._toffs = {}
        self._err = None
        self._pending_checks = None      # id columns collected by features() for one fused check

    # -- variables --------------------------------------------------------------------------
    @staticmethod
    def _layer_name(k):
        return 'dense' if k == 0 else 'dense_%d' % k

    def _new_variable(self, name, value):
        off = self._arena.numel()
        pad = (-value.numel()) % 4          # keep every variable 16-byte aligned for float4 access
        grown = torch.zeros(off + value.numel() + pad, dtype=torch.float32, device=self.device)
        grown[:off] = self._arena.detach()
        grown[off:off + value.numel()] = value.reshape(-1).to(self.device)
        self._arena = grown
        self._slots[name] = (off, tuple(value.shape))

    def _view(self, name):
        off, shape = self._slots[name]
        n = int(np.prod(shape))
        return self._arena[off:off + n].view(shape)

    def _leaf(self, name):
        """The variable as an autograd leaf: a detached view of the arena (so the optimizer's in-place update of
        the arena is the update of the leaf) whose .grad is a view of one flat gradient buffer - the backward
        kernels store each variable's gradient straight into its slice (see _GradSink; autograd adds it there in
        the cases that need accumulation) instead of building a zero-filled full-size gradient of a shared
        parameter tensor per variable and adding those up."""
        if self._leaves_of != self._arena.data_ptr() or self._grads is None or self._grads.numel() != self._arena.numel():
            self._grads = torch.zeros_like(self._arena)
            self._leaves = {}
            self._sinks = {}
            self._leaves_of = self._arena.data_ptr()
        leaf = self._leaves.get(name)
        if leaf is None:
            off, shape = self._slots[name]
            n = int(np.prod(shape))
            leaf = self._arena[off:off + n].view(shape).detach().requires_grad_(True)
            leaf.grad = self._grads[off:off + n].view(shape)
            self._leaves[name] = leaf
            self._sinks[name] = _GradSink(leaf.grad)
        return leaf

    def zero_grad(self):
        if self._grads is not None:
            self._grads.zero_()
        for sink in self._sinks.values():
            sink.dirty = False

    def backward_done(self):
        """Tell the model that the graphs built by its forward calls so far have been backpropagated (the train
        step does, after loss.backward()): the next forward starts counting the uses of each variable afresh.
        Without it gradients stay correct - every op just falls back to autograd's accumulating adds."""
        for sink in self._sinks.values():
            sink.uses = 0

    def grad_arena(self):
        """The flat gradient buffer (same layout as parameters()); filled by backward."""
        return self._grads

    def _dense_vars(self, in_dim, units, first):
        """Variables of the next ``tf.layers.dense`` call (r_model.py:53-57): truncated normal
        (|x| <= 2, redrawn) for the first layer of a network, uniform [0,1) for the others,
        zero bias."""
        base = self._layer_name(self._cursor)
        self._cursor += 1
        if base + '/kernel' not in self._slots:
            if first:
                k = torch.empty((in_dim, units), dtype=torch.float32)
                torch.nn.init.trunc_normal_(k, mean=0.0, std=1.0, a=-2.0, b=2.0, generator=self._gen)
            else:
                k = torch.rand((in_dim, units), dtype=torch.float32, generator=self._gen)
            self._new_variable(base + '/kernel', k)
            self._new_variable(base + '/bias', torch.zeros((units,), dtype=torch.float32))
        sinks = (None, None)
        if self._want_grad and torch.is_grad_enabled():
            kernel, bias = self._leaf(base + '/kernel'), self._leaf(base + '/bias')
            sinks = (self._sinks[base + '/kernel'], self._sinks[base + '/bias'])
            for sink in sinks:
                sink.uses += 1
        else:
            kernel, bias = self._view(base + '/kernel'), self._view(base + '/bias')
        if tuple(kernel.shape) != (in_dim, units):
            raise ValueError("layer %s was built for %s, now asked for %s" % (base, tuple(kernel.shape),
                                                                               (in_dim, units)))
        return kernel, bias, sinks

    @property
    def variables(self):
        """name -> tensor view, in creation order (TensorFlow's variable names)."""
        return OrderedDict((n, self._view(n).detach()) for n in self._slots)

    def parameters(self):
        """The single flat fp32 buffer holding every variable (the autograd leaves are views of it: _leaf)."""
        return self._arena

    def load_variables(self, values):
        """Set variables from a name -> array mapping (e.g. the oracle's or a checkpoint's)."""
        with torch.no_grad():
            for name, v in values.items():
                v = torch.as_tensor(np.asarray(v), dtype=torch.float32)
                if name not in self._slots:
                    self._new_variable(name, v)
                else:
                    self._view(name).copy_(v.to(self.device))

    def _table_offsets(self, size):
        if size not in self._toffs:
            self._toffs[size] = ops.table_offsets([size], self.device)
        return self._toffs[size]

    def _ids(self, x):
        if not torch.is_tensor(x):
            x = torch.as_tensor(np.asarray(x))
        if x.dtype not in (torch.int32, torch.int64):
            x = x.to(torch.int32)
        return x.to(self.device).contiguous()

    def _check(self, ids, size, flags):
        if not self.validate:
            return
        if self._err is None:
            if self._in_graph:
                raise RuntimeError("the id error word must exist before a step is captured")
            self._err = ops.new_error_word(self.device)
        if self._pending_checks is not None:        # inside features(): all its columns are checked by one launch
            self._pending_checks.append((ids, size, flags))
            return
        ops.validate_ids(self._table_offsets(size), 1, ids, flags, self._err, bag_len=ids.shape[-1] if ids.dim() > 1 else 1)

    def _flush_checks(self):
        checks, self._pending_checks = self._pending_checks, None
        if not checks:
            return
        if len(set(c[0].dtype for c in checks)) == 1 and len(checks) <= ops.MI_MAX_ID_COLUMNS:
            ops.validate_id_columns(checks, self._err)
        else:
            for ids, size, flags in checks:
                ops.validate_ids(self._table_offsets(size), 1, ids, flags, self._err,
                                 bag_len=ids.shape[-1] if ids.dim() > 1 else 1)

    def check_errors(self):
        """Raise InvalidArgumentError if any id seen since the last check was invalid.  Inside a captured step
        nothing is read back (the kernels still OR their findings into the error word); the owner of the graph
        calls this between replays."""
        if self._err is not None and not self._in_graph:
            ops.check_error_word(self._err)

    # -- reference API ------------------------------------------------------------------------
    def forward(self, features, units=[[8], [4], [8, 3, 3], [8, 3, 3], 8, 4]):
        n_layers = len(units[0]) + len(units[1]) + len(units[2]) + len(units[3]) + 3
        if self._layer_name(n_layers - 1) + '/bias' not in self._slots:
            # first call: create every variable (in the reference's order) before any view of the
            # parameter buffer enters the autograd graph - creating one reallocates the buffer
            with torch.no_grad():
                self._wire(features, units)
        self._want_grad = torch.is_grad_enabled()
        return self._wire(features, units)

    def _wire(self, features, units):
        """The graph of r_model.py:31-40."""
        self._cursor = 0
        user_layer = self.network(features['user'], units[0])
        country_layer = self.network(features['country'], units[1])
        click_1_feature = concat([features['recommend_colours_1'], features['click_colour_1']], axis=1)
        r_1 = self.network(click_1_feature, units[2])
        click_2_feature = concat([features['recommend_colours_2'], features['click_colour_2']], axis=1)
        r_2 = self.network(click_2_feature, units[3])
        concat_layer = concat([user_layer, country_layer, r_1, r_2], axis=1)
        last_layer = self.network(concat_layer, [units[4], units[5], self.colour_count])
        return last_layer

    def features(self, features):
        self._pending_checks = [] if self.validate else None
        try:
            user_fs = self.input_to_one_hot(features['user'], self.user_count)
            country_fs = self.input_to_one_hot(features['country'], self.country_count)
            r_c_1 = self.input_to_n_hot(features['recommend_colours_1'], self.colour_count)
            c_c_1 = self.input_to_one_hot_plus(features['click_colour_1'], self.colour_count)
            r_c_2 = self.input_to_n_hot(features['recommend_colours_2'], self.colour_count)
            c_c_2 = self.input_to_one_hot_plus(features['click_colour_2'], self.colour_count)
        except Exception:
            self._pending_checks = None
            raise
        self._flush_checks()
        # The error word is read (a host-device synchronisation) every `check_every` calls: 1 reports a
        # bad id in the call that fed it, as session.run does; a training loop that can live with hearing
        # of it a few steps later (the bad ids are never dereferenced meanwhile) sets it higher and calls
        # check_errors() once at the end.
        self._feature_calls += 1
        if self.check_every and self._feature_calls % self.check_every == 0:
            self.check_errors()
        return {'user': user_fs, 'country': country_fs, 'recommend_colours_1': r_c_1, 'click_colour_1': c_c_1,
                'recommend_colours_2': r_c_2, 'click_colour_2': c_c_2}

    def network(self, inputs, units):
        requires_grad = self._want_grad and torch.is_grad_enabled()
        if isinstance(inputs, SparseFeature):
            kernel, bias, sinks = self._dense_vars(inputs.size, units[0], first=True)
            if requires_grad:
                layer = _EmbagDense.apply(kernel, bias, inputs.ids, self._table_offsets(inputs.size), *sinks)
            else:
                layer = ops.embag_fwd(kernel, self._table_offsets(inputs.size), 1, inputs.ids,
                                      bag_len=inputs.ids.shape[1], bias=bias)
        else:
            kernel, bias, sinks = self._dense_vars(inputs.shape[1], units[0], first=True)
            layer = (_Linear.apply(inputs, kernel, bias, *sinks) if requires_grad
                     else ops.linear_fwd(inputs.contiguous(), kernel, bias))
        for i in units[1:]:
            kernel, bias, sinks = self._dense_vars(layer.shape[1], i, first=False)
            layer = (_Linear.apply(layer, kernel, bias, *sinks) if requires_grad
                     else ops.linear_fwd(layer.contiguous(), kernel, bias))
        return layer

    def input_to_one_hot(self, input, size):
        ids = self._ids(input).view(-1, 1)
        self._check(ids, size, ops.MI_CHECK_RANGE | ops.MI_CHECK_NONNEG)
        return SparseFeature(ids, size)

    def input_to_n_hot(self, input, size):
        ids = self._ids(input)
        if ids.dim() != 2 or ids.shape[1] != self.recommend_num:
            raise InvalidArgumentError("n-hot input must be [batch, %d], got %s" % (self.recommend_num,
                                                                                    tuple(ids.shape)))
        self._check(ids, size, ops.MI_CHECK_RANGE | ops.MI_CHECK_NONNEG | ops.MI_CHECK_INCREASING)
        return SparseFeature(ids, size)

    def input_to_one_hot_plus(self, input, size):
        ids = self._ids(input).view(-1, 1)
        self._check(ids, size, ops.MI_CHECK_RANGE)
        return SparseFeature(ids, size)

    def output(self, input):
        r = ops.softmax_heads(input.detach().contiguous(), k=self.recommend_num, want_top=True)
        return r['top_idx'], r['top_val']

    def top_one_output(self, input):
        r = ops.softmax_heads(input.detach().contiguous(), k=1, want_top=True)
        return torch.squeeze(r['top_idx']), torch.squeeze(r['top_val'])

    def softmax(self, input):
        return ops.softmax_heads(input.detach().contiguous(), want_probs=True)['probs']

    def inference(self, features):
        fs = self.features(features)
        with torch.no_grad():
            l = self.forward(fs)
        return self.output(l)[0]

    def loss(self, logits, labels):
        if isinstance(labels, SparseFeature):
            self.check_errors()
            labels = labels.to_dense()
        if logits.requires_grad:
            return _MeanSoftmaxXent.apply(logits, labels)
        r = ops.softmax_heads(logits.contiguous(), labels=labels, want_loss=True)
        return ops.reduce_sum(r['loss_rows'], 1.0 / logits.shape[0]).view(())

    def sparse_feature(self, input, size, dimension):
        """Dead code in the reference (r_model.py:121-130, no caller; it hands a SparseTensor to
        tf.feature_column.embedding_column, which TensorFlow rejects)."""
        raise NotImplementedError("sparse_feature has no caller in the reference and cannot run there either")

    def accuracy(self, labels, predictions, element_count):
        labels = self._ids(labels).view(-1)
        predictions = self._ids(predictions)
        if predictions.shape != (labels.numel(), element_count):
            raise InvalidArgumentError("predictions must be [%d, %d]" % (labels.numel(), element_count))
        hits = ops.hit_rows(labels, predictions)
        return ops.reduce_sum(hits, 1.0 / max(labels.numel(), 1)).view(())


# ----------------------------------------------------------------------------- records
_BAG_DEFAULT = '0,0,0,0,0,0'
COLUMNS = ('user', 'country', 'recommend_colours_1', 'click_colour_1', 'recommend_colours_2', 'click_colour_2',
           'label')


def _int32(text):
    v = int(text)
    if not -2 ** 31 <= v < 2 ** 31:
        raise InvalidArgumentError("Field is not a valid int32: %r" % text)
    return v


def parse_record(line, n_fields):
    """One text record -> python ints / lists, following tf.decode_csv(field_delim=' ') with the
    reference's record_defaults and its ',' split of the two bag fields (r_model.py:146-153)."""
    fields = line.rstrip('\r\n').split(' ')
    if len(fields) != n_fields:
        raise InvalidArgumentError("Expect %d fields but have %d in record" % (n_fields, len(fields)))
    out = []
    for k, f in enumerate(fields):
        try:
            if k in (2, 4):
                out.append([_int32(t) for t in (f if f != '' else _BAG_DEFAULT).split(',') if t != ''])
            else:
                out.append(_int32(f) if f != '' else 0)
        except ValueError:
            raise InvalidArgumentError("Field %d in record is not a valid int32: %r" % (k, f))
    return out


def parse_records(text, n_fields, bag_len=6, max_rows=None):
    """Many text records at once through the library's host-side tokenizer (mi_parse_records): the
    same grammar as parse_record, every list field exactly ``bag_len`` long.  ``text``: bytes or str.
    Returns (columns, consumed): the n_fields int32 numpy columns in record order (list fields as
    [n, bag_len]) and the number of bytes taken."""
    import ctypes
    from . import _lib
    if isinstance(text, str):
        text = text.encode()
    if max_rows is None:
        max_rows = text.count(b'\n') + 1
    scalars = np.empty((max_rows, n_fields - 2), dtype=np.int32)
    bags = np.empty((max_rows, 2, bag_len), dtype=np.int32)
    n, used = ctypes.c_int64(0), ctypes.c_int64(0)
    try:
        _lib.call('mi_parse_records', text, len(text), n_fields, bag_len, max_rows, scalars.ctypes.data,
                  bags.ctypes.data, ctypes.byref(n), ctypes.byref(used))
    except _lib.MiError as e:
        raise InvalidArgumentError(str(e))
    n = n.value
    cols, si = [], 0
    for k in range(n_fields):
        if k in (2, 4):
            cols.append(np.ascontiguousarray(bags[:n, 0 if k == 2 else 1]))
        else:
            cols.append(np.ascontiguousarray(scalars[:n, si]))
            si += 1
    return tuple(cols), used.value


class Dataset(object):
    """The slice of tf.data the reference uses: lines -> map(parse_csv) -> repeat -> batch.
    A batch of lines is tokenised by one call of the library's host-side tokenizer; set
    ``fast=False`` to parse record by record in Python (same results, used as the cross-check)."""

    def __init__(self, line_source, n_fields, batch_size, repeat, fast=True):
        self._source, self._n_fields, self._batch, self._repeat = line_source, n_fields, batch_size, repeat
        self._fast = fast

    def _lines(self):
        epoch = 0
        while self._repeat < 0 or epoch < self._repeat:
            n = 0
            for ln in self._source():
                if ln.strip('\r\n') == '':
                    continue
                n += 1
                yield ln
            if n == 0:
                return
            epoch += 1

    def __iter__(self):
        if self._fast:
            for cols in self._iter_fast():
                yield cols
            return
        rows = []
        for ln in self._lines():
            rows.append(parse_record(ln, self._n_fields))
            if len(rows) == self._batch:
                yield self._stack(rows)
                rows = []
        if rows:                       # tf.data batch() keeps the final partial batch
            yield self._stack(rows)

    def _iter_fast(self):
        lines, bag_len = [], None
        def flush():
            text = '\n'.join(l.rstrip('\r\n') for l in lines)
            cols, _ = parse_records(text, self._n_fields, bag_len, len(lines))
            return tuple(torch.from_numpy(c) for c in cols)
        for ln in self._lines():
            if bag_len is None:        # the batch's list width (tf.sparse.to_dense needs one width per batch)
                f = ln.rstrip('\r\n').split(' ')
                bag_len = max(1, len([t for t in f[2].split(',') if t != ''])) if len(f) > 2 and f[2] != '' else 6
            lines.append(ln)
            if len(lines) == self._batch:
                yield flush()
                lines = []
        if lines:
            yield flush()

    def _stack(self, rows):
        cols = []
        for k in range(self._n_fields):
            cols.append(torch.from_numpy(np.asarray([r[k] for r in rows], dtype=np.int32)))
        return tuple(cols)

    def make_one_shot_iterator(self):
        return _Iterator(iter(self))


class _Iterator(object):
    def __init__(self, it):
        self._it = it

    def get_next(self):
        try:
            return next(self._it)
        except StopIteration:
            raise OutOfRangeError("End of sequence")

    def __iter__(self):
        return self._it


def _file_lines(files):
    def source():
        for path in files:
            with open(path, 'r') as f:
                for ln in f:
                    yield ln
    return source


class Sample(object):
    @staticmethod
    def read_csv(file_path, batch_size):
        """6 columns, one epoch (r_model.py:145-158)."""
        return Dataset(_file_lines([file_path]), 6, batch_size, repeat=1)

    @staticmethod
    def read_label_data(file_path, batch_size, parallelism=None, index=None):
        """7 columns, repeated for ever, files sharded ``files[index::parallelism]``
        (r_model.py:161-187)."""
        files = file_path if isinstance(file_path, list) else [file_path]
        logger.info("{} all files: {}".format(index, files))
        if parallelism is not None and index is not None:
            partitioned_files = files[index::parallelism]
        else:
            partitioned_files = files
        logger.info("index: {} read files: {}".format(index, partitioned_files))
        return Dataset(_file_lines(partitioned_files), 7, batch_size, repeat=-1)

    @staticmethod
    def read_flink_dataset(dataset, batch_size):
        """7 columns from any iterable of record strings (r_model.py:190-202; the reference's
        source is a Flink stream)."""
        return Dataset(lambda: iter(dataset), 7, batch_size, repeat=1)
