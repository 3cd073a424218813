"""Torch-tensor wrappers over the C ABI (libmi355rec.so).

PyTorch is used for device memory and streams only: every arithmetic step
below is one call into a hand-written HIP kernel on torch's current stream.
"""
import ctypes

import torch

from . import _lib
from ._lib import (MI_CHECK_INCREASING, MI_CHECK_NONNEG, MI_CHECK_RANGE, MI_DT_BF16, MI_DT_F32,
                   MI_IDERR_CAPACITY, MI_IDERR_ORDER, MI_IDERR_RANGE, MI_OPT_ADAGRAD_ROWWISE, MI_OPT_ADAM_TF, MI_OPT_SGD,
                   MI_POOL_MEAN, MI_POOL_SUM, BwdPlan, MiError)


class InvalidArgumentError(ValueError):
    """Raised where TensorFlow would raise ``tf.errors.InvalidArgumentError`` (an id out
    of range, or the ids of an n-hot row repeated / out of order)."""


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_RAW_DEVICE = getattr(torch._C, '_cuda_getDevice', None)
if _RAW_DEVICE is None:
    _RAW_STREAM = None


def _stream():
    # torch.cuda.current_stream() costs ~8 us a call (device-index resolution in Python), more than enqueueing
    # the kernel it is asked for; the raw accessors are the ones torch's own compiled-graph runtime uses
    if _RAW_STREAM is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise MiError("the HIP path needs device tensors; got a %s tensor (no CPU fallback)" % t.device)


def _idx_bits(idx):
    if idx.dtype == torch.int32:
        return 32
    if idx.dtype == torch.int64:
        return 64
    raise TypeError("ids must be int32 or int64, got %s" % idx.dtype)


def new_error_word(device):
    return torch.zeros(1, dtype=torch.int32, device=device)


def check_error_word(err, what="ids"):
    """Synchronising read of the device error word; raises like TensorFlow would."""
    code = int(err.item())
    if code:
        err.zero_()
        msgs = []
        if code & MI_IDERR_RANGE:
            msgs.append("index out of bounds")
        if code & MI_IDERR_ORDER:
            msgs.append("indices of a row are out of order or repeated")
        if code & MI_IDERR_CAPACITY:
            msgs.append("a rank pair needs more rows than the exchange block holds (raise capacity_factor)")
        if code & ~(MI_IDERR_RANGE | MI_IDERR_ORDER | MI_IDERR_CAPACITY):
            msgs.append("internal device error 0x%x" % code)
        raise InvalidArgumentError("%s: %s" % (what, '; '.join(msgs)))


class TFAdamSchedule(object):
    """The scalar side of tf.train.AdamOptimizer as TensorFlow's fp32 arithmetic forms it:
    beta1_power / beta2_power are fp32 values multiplied by beta once per step and
    alpha_t = lr*sqrt(1 - beta2_power)/(1 - beta1_power) is evaluated in fp32."""

    def __init__(self, lr, beta1=0.9, beta2=0.999):
        import numpy as np
        self._f = np.float32
        self.lr, self.beta1, self.beta2 = lr, beta1, beta2
        self.b1p, self.b2p = self._f(1.0), self._f(1.0)
        self.t = 0

    def step(self):
        """Advance to the next step and return its alpha_t (a python float holding an fp32 value)."""
        import numpy as np
        f = self._f
        self.t += 1
        self.b1p, self.b2p = f(self.b1p * f(self.beta1)), f(self.b2p * f(self.beta2))
        return float(f(f(self.lr) * np.sqrt(f(1.0) - self.b2p) / (f(1.0) - self.b1p)))


def table_offsets(rows_per_table, device):
    """int64 [T+1] global row offsets for tables stored back to back."""
    offs = [0]
    for r in rows_per_table:
        offs.append(offs[-1] + int(r))
    return torch.tensor(offs, dtype=torch.int64, device=device)


# ----------------------------------------------------------------------------- forward
def embag_fwd(weights, toffs, T, indices, offsets=None, bag_len=None, pool_mode=MI_POOL_SUM, check_flags=0,
              bias=None, err=None, out=None, out_dtype=torch.float32, out_ld=None, n_bags=None):
    """Pooled embedding rows for all bags of all T tables (one launch)."""
    _need_cuda(weights, toffs, indices, offsets, bias, err, out)
    dim = weights.shape[1]
    if offsets is None:
        if bag_len is None:
            bag_len = indices.shape[-1] if indices.dim() > 1 else 1
        if n_bags is None:
            n_bags = indices.numel() // max(bag_len, 1)
    else:
        n_bags = offsets.numel() - 1
        bag_len = 0
    B = n_bags // T
    if out_ld is None:
        out_ld = T * dim
    if out is None:
        out = torch.empty((B, out_ld), dtype=out_dtype, device=weights.device)
    dt = MI_DT_BF16 if out.dtype == torch.bfloat16 else MI_DT_F32
    indices = indices.contiguous()
    _lib.call('mi_embag_fwd', _p(weights), _p(toffs), T, dim, _p(indices), _idx_bits(indices), _p(offsets),
              bag_len, n_bags, pool_mode, check_flags, _p(bias), _p(out), dt, out_ld, _p(err), _stream())
    return out


def onehot_dense(indices, size, check_flags=0, err=None):
    _need_cuda(indices, err)
    idx = indices.contiguous()
    if idx.dim() == 1:
        idx = idx.view(-1, 1)
    n_bags, bag_len = idx.shape
    out = torch.empty((n_bags, size), dtype=torch.float32, device=idx.device)
    _lib.call('mi_onehot_dense', _p(idx), _idx_bits(idx), bag_len, n_bags, size, check_flags, _p(out), _p(err),
              _stream())
    return out


def validate_ids(toffs, T, indices, check_flags, err, offsets=None, bag_len=None):
    _need_cuda(toffs, indices, err, offsets)
    idx = indices.contiguous()
    if offsets is None:
        if bag_len is None:
            bag_len = idx.shape[-1] if idx.dim() > 1 else 1
        n_bags = idx.numel() // max(bag_len, 1)
    else:
        n_bags, bag_len = offsets.numel() - 1, 0
    _lib.call('mi_validate_ids', _p(toffs), T, _p(idx), _idx_bits(idx), _p(offsets), bag_len, n_bags, check_flags,
              _p(err), _stream())


MI_MAX_ID_COLUMNS = 8


def _id_columns(columns):
    """[(ids [B, L], table size, check flags)] -> (array of struct mi_id_column, idx_bits, tensors kept alive)."""
    if not 1 <= len(columns) <= MI_MAX_ID_COLUMNS:
        raise MiError("1..%d id columns, got %d" % (MI_MAX_ID_COLUMNS, len(columns)))
    keep = [ids.contiguous() for ids, _, _ in columns]
    _need_cuda(*keep)
    bits = _idx_bits(keep[0])
    if any(_idx_bits(t) != bits for t in keep):
        raise MiError("the id columns of one call must share an integer type")
    arr = (_lib.IdColumn * len(columns))()
    for c, t, (_, size, flags) in zip(arr, keep, columns):
        bag_len = t.shape[-1] if t.dim() > 1 else 1
        c.ids, c.n_bags, c.rows, c.bag_len, c.check_flags = t.data_ptr(), t.numel() // max(bag_len, 1), size, bag_len, flags
    return arr, bits, keep


def validate_id_columns(columns, err):
    """validate_ids for several columns [(ids, table size, check flags)] in one launch."""
    arr, bits, keep = _id_columns(columns)
    _need_cuda(err)
    _lib.call('mi_validate_id_columns', arr, len(columns), bits, _p(err), _stream())


def concat_id_columns(columns):
    """Ids of tf.concat(axis=1) of one-hot / n-hot matrices given as [(ids [B, L_i], table size_i)]: column i's ids
    shifted by the sizes before it (ids < 0 kept), one launch."""
    arr, bits, keep = _id_columns([(ids, size, 0) for ids, size in columns])
    B = keep[0].shape[0]
    width = sum(int(c.bag_len) for c in arr)
    out = torch.empty((B, width), dtype=keep[0].dtype, device=keep[0].device)
    _lib.call('mi_concat_id_columns', arr, len(columns), bits, B, _p(out), _stream())
    return out


# ----------------------------------------------------------------------------- sparse backward
class SortedPlan(object):
    """Device artefacts of ``mi_embag_bwd_prepare`` plus the struct the apply calls take."""

    def __init__(self, toffs, T, total_rows, indices, offsets=None, bag_len=None, pool_mode=MI_POOL_SUM,
                 workspace=None, runs=True):
        """``runs=False`` skips run detection (unique_rows / seg_offsets / counts stay unset):
        the fused apply kernels then find the runs in sorted_rows themselves."""
        _need_cuda(toffs, indices, offsets)
        self.runs = runs
        dev = indices.device
        self.indices = indices.contiguous()
        self.offsets = offsets
        self.toffs = toffs
        self.T = T
        self.total_rows = int(total_rows)
        n = self.indices.numel()
        if offsets is None:
            if bag_len is None:
                bag_len = self.indices.shape[-1] if self.indices.dim() > 1 else 1
            n_bags = n // max(bag_len, 1)
        else:
            n_bags, bag_len = offsets.numel() - 1, 0
        self.n_entries, self.n_bags, self.bag_len = n, n_bags, bag_len
        self.sorted_rows = torch.empty(max(n, 1), dtype=torch.int32, device=dev)  # uint32 bit patterns
        self.perm = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        self.unique_rows = torch.empty(max(n, 1), dtype=torch.int32, device=dev) if runs else None
        self.seg_offsets = torch.empty(n + 1, dtype=torch.int32, device=dev) if runs else None
        self.counts = torch.zeros(2, dtype=torch.int32, device=dev) if runs else None
        self.long_list = torch.zeros(n + 2, dtype=torch.int32, device=dev)   # first two ints must start at zero
        ws_bytes = _lib.load().mi_embag_bwd_workspace_bytes(n)
        if workspace is None or workspace.numel() < ws_bytes:
            workspace = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
        self.workspace = workspace
        self.pool_mode = pool_mode
        # per-bag facts the apply kernels would otherwise recompute per entry: the mean-pooling scale
        # and, for CSR bags, the bag of each position
        self.bag_scale = (torch.empty(max(n_bags, 1), dtype=torch.float32, device=dev)
                          if pool_mode == MI_POOL_MEAN else None)
        self.entry_bag = torch.empty(max(n, 1), dtype=torch.int32, device=dev) if offsets is not None else None
        self._bind()
        self.prepare()

    def prepare(self, indices=None):
        """(Re)run the sort. ``indices`` must have the shape/dtype this plan was built for;
        no device memory is allocated, so the call can sit inside a captured graph."""
        if indices is not None:
            indices = indices.contiguous()
            if indices.numel() != self.n_entries or indices.dtype != self.indices.dtype:
                raise MiError("SortedPlan.prepare: indices shape/dtype differ from the plan's")
            self.indices = indices
            self._bind()
        _lib.call('mi_embag_bwd_prepare', _p(self.toffs), self.T, self.total_rows, _p(self.indices),
                  _idx_bits(self.indices), _p(self.offsets), self.bag_len, self.n_bags, self.n_entries,
                  _p(self.sorted_rows), _p(self.perm), _p(self.unique_rows), _p(self.seg_offsets), _p(self.counts),
                  _p(self.workspace), self.workspace.numel(), _stream())
        if self.bag_scale is not None or self.entry_bag is not None:
            _lib.call('mi_embag_bag_info', _p(self.toffs), self.T, _p(self.indices), _idx_bits(self.indices),
                      _p(self.offsets), self.bag_len, self.n_bags, _p(self.bag_scale), _p(self.entry_bag), _stream())
        return self

    def prepare_prefix(self, n):
        """Sort only the first ``n`` entries of the plan's index buffer (fixed bag length, n a
        multiple of it): for callers whose entry count changes from step to step below a known
        capacity.  Allocates nothing."""
        cap = self.indices.numel()
        if self.offsets is not None or n < 1 or n > cap or n % max(self.bag_len, 1):
            raise MiError("SortedPlan.prepare_prefix: n=%d outside 1..%d or not whole bags" % (n, cap))
        self.n_entries, self.n_bags = n, n // max(self.bag_len, 1)
        self._bind()
        return self.prepare()

    def _bind(self):
        n, n_bags, bag_len, T, offsets, toffs = (self.n_entries, self.n_bags, self.bag_len, self.T, self.offsets,
                                                 self.toffs)
        pool_mode = self.pool_mode
        self.c = BwdPlan(
            sorted_rows=self.sorted_rows.data_ptr(), perm=self.perm.data_ptr(),
            unique_rows=self.unique_rows.data_ptr() if self.runs else None,
            seg_offsets=self.seg_offsets.data_ptr() if self.runs else None,
            counts=self.counts.data_ptr() if self.runs else None, n_entries=n, n_bags=n_bags,
            total_rows=self.total_rows,
            offsets=offsets.data_ptr() if offsets is not None else None, bag_len=bag_len, T=T,
            pool_mode=pool_mode, long_list=self.long_list.data_ptr(), indices=self.indices.data_ptr(),
            idx_bits=_idx_bits(self.indices), table_offsets=toffs.data_ptr(),
            bag_scale=self.bag_scale.data_ptr() if self.bag_scale is not None else None,
            entry_bag=self.entry_bag.data_ptr() if self.entry_bag is not None else None)

    def ref(self):
        return ctypes.byref(self.c)

    def host(self):
        """(sorted_rows, perm, unique_rows, seg_offsets) trimmed to their valid lengths, on the host."""
        n_unique, n_valid = (int(v) for v in self.counts.tolist())
        u32 = lambda t, k: (t[:k].cpu().numpy().astype('int64') & 0xFFFFFFFF)
        return (u32(self.sorted_rows, n_valid), self.perm[:n_valid].cpu().numpy(), u32(self.unique_rows, n_unique),
                self.seg_offsets[:n_unique + 1].cpu().numpy())


def sort_pairs_u32(keys, vals, key_bits=32):
    """In-place stable radix sort of int32-typed tensors holding uint32 bit patterns."""
    _need_cuda(keys, vals)
    n = keys.numel()
    ws_bytes = _lib.load().mi_embag_bwd_workspace_bytes(n)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=keys.device)
    _lib.call('mi_sort_pairs_u32', _p(keys), _p(vals), n, key_bits, _p(ws), ws.numel(), _stream())
    return keys, vals


def bwd_dense_grad(plan, grad_out, dim, total_rows, grad_ld=None, out=None):
    if grad_ld is None or grad_out.stride(-1) != 1:
        grad_out = grad_out.contiguous()
        grad_ld = None
    dW = out if out is not None else torch.empty((total_rows, dim), dtype=torch.float32, device=grad_out.device)
    assert dW.is_contiguous() and tuple(dW.shape) == (total_rows, dim)
    _lib.call('mi_embag_bwd_dense_grad', plan.ref(), _p(grad_out), grad_ld or grad_out.shape[-1], dim, total_rows,
              _p(dW), _stream())
    return dW


def bwd_apply_sgd(plan, grad_out, weights, lr, grad_ld=None):
    _lib.call('mi_embag_bwd_apply_sgd', plan.ref(), _p(grad_out), grad_ld or grad_out.shape[-1], weights.shape[1],
              _p(weights), lr, _stream())


def bwd_apply_adagrad_rowwise(plan, grad_out, weights, state, lr, eps, grad_ld=None):
    _lib.call('mi_embag_bwd_apply_adagrad_rowwise', plan.ref(), _p(grad_out), grad_ld or grad_out.shape[-1],
              weights.shape[1], _p(weights), _p(state), lr, eps, _stream())


def bwd_apply_adam_tf(plan, grad_out, weights, m, v, lr_t, beta1, beta2, eps, grad_ld=None):
    _lib.call('mi_embag_bwd_apply_adam_tf', plan.ref(), _p(grad_out), grad_ld or grad_out.shape[-1],
              weights.shape[1], _p(weights), _p(m), _p(v), lr_t, beta1, beta2, eps, _stream())


def bwd_apply_adam_dense_tf(plan, grad_out, weights, m, v, lr_t, beta1, beta2, eps, grad_ld=None):
    _lib.call('mi_embag_bwd_apply_adam_dense_tf', plan.ref(), _p(grad_out), grad_ld or grad_out.shape[-1],
              weights.shape[1], weights.shape[0], _p(weights), _p(m), _p(v), lr_t, beta1, beta2, eps, _stream())


# ----------------------------------------------------------------------------- small dense pieces
def linear_fwd(x, W, b=None, out=None):
    _need_cuda(x, W, b, out)
    assert x.stride(-1) == 1 and W.is_contiguous()
    B, K = x.shape
    N = W.shape[1]
    if out is None:
        out = torch.empty((B, N), dtype=torch.float32, device=x.device)
    _lib.call('mi_linear_fwd', _p(x), x.stride(0), _p(W), _p(b), _p(out), out.stride(0), B, K, N, _stream())
    return out


def linear_bwd(x, W, dy, need_dx=True, dW_out=None, db_out=None):
    """dx, dW, db of y = x W + b; dW_out / db_out, when given, receive the two parameter gradients (stored, not
    added)."""
    _need_cuda(x, W, dy, dW_out, db_out)
    assert x.stride(-1) == 1 and dy.stride(-1) == 1 and W.is_contiguous()
    B, K = x.shape
    N = W.shape[1]
    dev = x.device
    dx = torch.empty((B, K), dtype=torch.float32, device=dev) if need_dx else None
    dW = dW_out if dW_out is not None else torch.empty((K, N), dtype=torch.float32, device=dev)
    db = db_out if db_out is not None else torch.empty((N,), dtype=torch.float32, device=dev)
    assert dW.is_contiguous() and tuple(dW.shape) == (K, N) and db.is_contiguous() and db.numel() == N
    ws_bytes = _lib.load().mi_linear_bwd_workspace_bytes(B, K, N)
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device=dev)
    _lib.call('mi_linear_bwd', _p(x), x.stride(0), _p(W), _p(dy), dy.stride(0), _p(dx), K, _p(dW), _p(db), B, K, N,
              _p(ws), ws.numel(), _stream())
    return dx, dW, db


def softmax_heads(logits, k=0, labels=None, label_ids=None, grad_scale=1.0, want_probs=False, want_top=False,
                  want_loss=False, want_dlogits=False, want_hits=False):
    _need_cuda(logits, labels, label_ids)
    assert logits.stride(-1) == 1
    B, N = logits.shape
    dev = logits.device
    f32 = dict(dtype=torch.float32, device=dev)
    probs = torch.empty((B, N), **f32) if want_probs else None
    top_idx = torch.empty((B, k), dtype=torch.int32, device=dev) if want_top else None
    top_val = torch.empty((B, k), **f32) if want_top else None
    loss_rows = torch.empty((B,), **f32) if want_loss else None
    dlogits = torch.empty((B, N), **f32) if want_dlogits else None
    hit_rows = torch.empty((B,), **f32) if want_hits else None
    if labels is not None:
        labels = labels.contiguous()
    _lib.call('mi_softmax_heads', _p(logits), logits.stride(0), B, N, k, _p(labels), _p(label_ids), grad_scale,
              _p(probs), _p(top_idx), _p(top_val), _p(loss_rows), _p(dlogits), _p(hit_rows), _stream())
    return dict(probs=probs, top_idx=top_idx, top_val=top_val, loss_rows=loss_rows, dlogits=dlogits,
                hit_rows=hit_rows)


def hit_rows(labels, preds):
    _need_cuda(labels, preds)
    labels = labels.contiguous().to(torch.int32)
    preds = preds.contiguous().to(torch.int32)
    B, k = preds.shape
    out = torch.empty((B,), dtype=torch.float32, device=labels.device)
    _lib.call('mi_hit_rows', _p(labels), _p(preds), B, k, _p(out), _stream())
    return out


def colsum(x, out=None):
    _need_cuda(x, out)
    assert x.stride(-1) == 1
    B, N = x.shape
    if out is None:
        out = torch.empty((N,), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.numel() == N
    _lib.call('mi_colsum', _p(x), x.stride(0), B, N, _p(out), _stream())
    return out


def reduce_sum(x, scale=1.0, out=None):
    _need_cuda(x)
    x = x.contiguous()
    if out is None:
        out = torch.empty((1,), dtype=torch.float32, device=x.device)
    _lib.call('mi_reduce_sum', _p(x), x.numel(), scale, _p(out), _stream())
    return out


def adam_tf_dense(p, m, v, g, lr_t, beta1, beta2, eps, lr_dev=None):
    _need_cuda(p, m, v, g)
    assert p.is_contiguous() and m.is_contiguous() and v.is_contiguous() and g.is_contiguous()
    _lib.call('mi_adam_tf_dense', _p(p), _p(m), _p(v), _p(g), p.numel(), lr_t, _p(lr_dev), beta1, beta2, eps,
              _stream())


# ----------------------------------------------------------------------------- FM
class FmRecords(object):
    """Row records of an FM (struct mi_fm_table): V[dim], w1 and the optimizer state of both,
    side by side in one 128-byte-aligned record per global row."""

    def __init__(self, total_rows, dim, opt, device):
        self.dim, self.opt = int(dim), opt
        nxt = self.dim
        self.off_w1 = nxt
        self.off_vs0 = self.off_vs1 = self.off_ws0 = self.off_ws1 = 0
        if opt == MI_OPT_ADAGRAD_ROWWISE:
            self.off_vs0, self.off_ws0 = nxt + 1, nxt + 2
            nxt += 3
        elif opt == MI_OPT_ADAM_TF:
            self.off_ws0, self.off_ws1 = nxt + 1, nxt + 2
            nxt = -(-(nxt + 3) // 32) * 32          # m and v start on a fresh 128-byte line
            self.off_vs0, self.off_vs1 = nxt, nxt + self.dim
            nxt += 2 * self.dim
        else:
            nxt += 1
        self.stride = -(-nxt // 32) * 32            # whole 128-byte lines per record
        self.rec = torch.zeros((int(total_rows), self.stride), dtype=torch.float32, device=device)
        self.c = _lib.FmTable(rec=self.rec.data_ptr(), stride=self.stride, dim=self.dim, off_w1=self.off_w1,
                              off_vs0=self.off_vs0, off_vs1=self.off_vs1, off_ws0=self.off_ws0,
                              off_ws1=self.off_ws1)

    def ref(self):
        return ctypes.byref(self.c)

    @property
    def V(self):
        return self.rec[:, :self.dim]

    @property
    def w1(self):
        return self.rec[:, self.off_w1]


def fm_fwd(table, toffs, ids, dense=None, wd=None, bias=None, labels=None, gscale=1.0, err=None, out=None):
    """Returns dict(logit, S, loss_rows, g). ``out`` may hold preallocated buffers of those names."""
    _need_cuda(table.rec, toffs, ids, dense, wd, bias, labels, err)
    B, T = ids.shape
    dim = table.dim
    dev = table.rec.device
    out = out or {}
    f32 = dict(dtype=torch.float32, device=dev)
    logit = out.get('logit') if out.get('logit') is not None else torch.empty((B,), **f32)
    S = out.get('S') if out.get('S') is not None else torch.empty((B, dim), **f32)
    loss_rows = g = None
    if labels is not None:
        loss_rows = out.get('loss_rows') if out.get('loss_rows') is not None else torch.empty((B,), **f32)
        g = out.get('g') if out.get('g') is not None else torch.empty((B,), **f32)
    n_dense = dense.shape[1] if dense is not None else 0
    _lib.call('mi_fm_fwd', table.ref(), _p(toffs), T, _p(ids), _idx_bits(ids), B, _p(dense), n_dense, _p(wd), _p(bias),
              _p(labels), gscale, _p(logit), _p(S), _p(loss_rows), _p(g), _p(err), _stream())
    return dict(logit=logit, S=S, loss_rows=loss_rows, g=g)


def fm_bwd_apply(plan, table, g, S, opt, lr, beta1=0.9, beta2=0.999, eps=1e-8, lr_dev=None, entry_sample=None):
    _lib.call('mi_fm_bwd_apply', plan.ref(), table.ref(), _p(g), _p(S), _p(entry_sample), opt, lr, _p(lr_dev), beta1,
              beta2, eps, _stream())


def fm_fwd_sort_max_batch():
    return int(_lib.load().mi_fm_fwd_sort_max_batch())


def fm_lds_sort_single_max():
    """Largest batch whose per-table sort takes ids in either layout (beyond it: field-major ids only)."""
    return int(_lib.load().mi_fm_lds_sort_single_max())


def fm_fwd_sort(table, toffs, ids, plan, dense=None, wd=None, bias=None, labels=None, gscale=1.0, err=None, out=None,
                sched=None, lr=0.0, beta1=0.9, beta2=0.999, ids_tm=None):
    """fm_fwd + the sort of ``plan`` (built without run arrays for ids of this shape) + one optimizer
    schedule step (``sched``), one launch.  Returns what fm_fwd returns; ``plan`` is ready for the apply.
    ``plan=None``: forward + schedule only (the caller sorted this batch earlier).
    ``ids_tm``: the same ids table-major, [T][B] (one array per field); read instead of ``ids`` when given."""
    _need_cuda(table.rec, toffs, ids, dense, wd, bias, labels, err, sched)
    B, T = ids.shape
    if plan is not None and (plan.runs or plan.n_entries != B * T or plan.bag_len != 1 or plan.T != T):
        raise MiError("fm_fwd_sort: the plan must be built with runs=False for ids of this shape")
    buf = out or {}
    f32 = dict(dtype=torch.float32, device=table.rec.device)
    logit = buf['logit'] if buf.get('logit') is not None else torch.empty((B,), **f32)
    S = buf['S'] if buf.get('S') is not None else torch.empty((B, table.dim), **f32)
    loss_rows = g = None
    if labels is not None:
        loss_rows = buf['loss_rows'] if buf.get('loss_rows') is not None else torch.empty((B,), **f32)
        g = buf['g'] if buf.get('g') is not None else torch.empty((B,), **f32)
    n_dense = dense.shape[1] if dense is not None else 0
    ids = ids.contiguous()
    if ids_tm is not None:
        _need_cuda(ids_tm)
        if tuple(ids_tm.shape) != (T, B) or ids_tm.dtype != ids.dtype or not ids_tm.is_contiguous():
            raise MiError("fm_fwd_sort: ids_tm must be the contiguous [T][B] transpose of ids")
    src = ids if ids_tm is None else ids_tm
    _lib.call('mi_fm_fwd_sort', table.ref(), _p(toffs), T, _p(src), _idx_bits(ids), B, _p(dense), n_dense, _p(wd),
              _p(bias), _p(labels), gscale, _p(logit), _p(S), _p(loss_rows), _p(g), _p(err),
              plan.total_rows if plan is not None else 0, _p(plan.sorted_rows) if plan is not None else None,
              _p(plan.perm) if plan is not None else None, _p(sched), lr, beta1, beta2, 0 if ids_tm is None else 1,
              _stream())
    return dict(logit=logit, S=S, loss_rows=loss_rows, g=g)


def fm_sort_ids(toffs, ids, plan, ids_tm=None):
    """The per-table LDS sort of ``ids`` [B][T] (read from ``ids_tm`` [T][B] when given) into ``plan``."""
    B, T = ids.shape
    _need_cuda(toffs, ids, ids_tm)
    if plan.runs or plan.n_entries != B * T or plan.bag_len != 1 or plan.T != T:
        raise MiError("fm_sort_ids: the plan must be built with runs=False for ids of this shape")
    if ids_tm is not None and (tuple(ids_tm.shape) != (T, B) or ids_tm.dtype != ids.dtype or not ids_tm.is_contiguous()):
        raise MiError("fm_sort_ids: ids_tm must be the contiguous [T][B] transpose of ids")
    src = ids.contiguous() if ids_tm is None else ids_tm
    _lib.call('mi_fm_sort_ids', _p(toffs), T, _p(src), _idx_bits(ids), B, 0 if ids_tm is None else 1, plan.total_rows,
              _p(plan.sorted_rows), _p(plan.perm), _stream())


def fm_bwd_apply_dense(plan, table, g, S, opt, lr, beta1, beta2, eps, dense, loss_rows, dp, dm, dv, loss_out,
                       lr_dev=None, dense_lr_t=0.0, dense_lr_dev=None, stage=None):
    """fm_bwd_apply (step size lr / lr_dev) + fm_dense_step (dense_lr_t / dense_lr_dev), one launch.
    ``stage`` = (device block, pinned host block of the same size): the launch also copies the host block - the
    NEXT batch - into the device block (mi_fm_bwd_apply_dense_stage)."""
    n_dense = dense.shape[1] if dense is not None else 0
    if stage is not None:
        dst, src = stage
        _need_cuda(dst)
        if src.is_cuda or not src.is_pinned() or not src.is_contiguous() or not dst.is_contiguous() or \
                src.numel() * src.element_size() != dst.numel() * dst.element_size():
            raise MiError("fm_bwd_apply_dense: stage takes (device block, pinned host block) of one size")
        _lib.call('mi_fm_bwd_apply_dense_stage', plan.ref(), table.ref(), _p(g), _p(S), opt, lr, _p(lr_dev), beta1, beta2,
                  eps, _p(dense), n_dense, _p(loss_rows), _p(dp), _p(dm), _p(dv), dense_lr_t, _p(dense_lr_dev),
                  _p(loss_out), dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size(), _stream())
        return
    _lib.call('mi_fm_bwd_apply_dense', plan.ref(), table.ref(), _p(g), _p(S), opt, lr, _p(lr_dev), beta1, beta2, eps,
              _p(dense), n_dense, _p(loss_rows), _p(dp), _p(dm), _p(dv), dense_lr_t, _p(dense_lr_dev), _p(loss_out),
              _stream())


def fm_bwd_dense(g, dense, out=None):
    n_dense = dense.shape[1] if dense is not None else 0
    if out is None:
        out = torch.empty((n_dense + 1,), dtype=torch.float32, device=g.device)
    _lib.call('mi_fm_bwd_dense', _p(g), _p(dense), n_dense, g.numel(), _p(out), _stream())
    return out


def fm_dense_step(g, dense, loss_rows, dp, dm, dv, lr_t, beta1, beta2, eps, loss_out, lr_dev=None):
    """Dense-side gradients + TensorFlow-form Adam on dp + mean loss, one launch."""
    n_dense = dense.shape[1] if dense is not None else 0
    _lib.call('mi_fm_dense_step', _p(g), _p(dense), n_dense, g.numel(), _p(loss_rows), _p(dp), _p(dm), _p(dv), lr_t,
              _p(lr_dev), beta1, beta2, eps, _p(loss_out), _stream())


def adam_schedule_step(sched, lr, beta1, beta2):
    """Advance the device-resident {alpha_t, beta1_power, beta2_power} by one optimizer step."""
    _lib.call('mi_adam_schedule_step', _p(sched), lr, beta1, beta2, _stream())


# ----------------------------------------------------------------------------- sharded tables
def shard_block_ints(B, T):
    """Ints per rank pair in the row exchange: chunk counts [ceil(B/64)][T], then slots [T][B]."""
    return -(-B // 64) * T + B * T


def shard_row_floats(dim, n_dense=0):
    """Floats per sample in the gathered rows [S | g | loss | pad], a multiple of 4 (the dense side
    travels as per-workgroup partial sums behind the sample rows; with n_dense the row also has
    room for the dense features)."""
    return -(-(dim + 2 + n_dense) // 4) * 4


def shard_route(toffs, ids, n_ranks, send, err=None):
    B, T = ids.shape
    if send.numel() < n_ranks * shard_block_ints(B, T):
        raise MiError("shard_route: send buffer too small")
    _lib.call('mi_shard_route', _p(toffs), T, _p(ids), _idx_bits(ids), B, n_ranks, _p(send), _p(err), _stream())


def shard_owner_fwd_sort(table, local_rows, toffs, rank, n_ranks, recv, B, partial, sorted_rows, perm, n_total,
                         sched, lr, beta1, beta2, err=None, row_stride=None):
    """One launch: partial forward of every (source rank, sample) + sort of the received slots
    (sorted_rows / perm [N*B*T] and n_total [1]; all None for the forward alone) + Adam schedule.
    The payload of entry (s, b) is s*row_stride + b (row_stride defaults to B)."""
    T = toffs.numel() - 1
    if recv.numel() < n_ranks * shard_block_ints(B, T) or partial.numel() < n_ranks * B * (table.dim + 2):
        raise MiError("shard_owner_fwd_sort: buffer too small")
    if sorted_rows is not None and min(sorted_rows.numel(), perm.numel()) < n_ranks * B * T:
        raise MiError("shard_owner_fwd_sort: sort output too small")
    _lib.call('mi_shard_owner_fwd_sort', table.ref(), local_rows, _p(toffs), T, rank, n_ranks, _p(recv), B,
              B if row_stride is None else row_stride, _p(partial), _p(sorted_rows), _p(perm), _p(n_total), _p(sched),
              lr, beta1, beta2, _p(err), _stream())


def shard_dense_part_floats(B, dim, n_dense):
    """Floats the combine launch leaves behind the B sample rows (one dense-side partial per workgroup)."""
    return int(_lib.load().mi_shard_dense_part_floats(B, dim, n_dense))


def fm_combine(partials, n_ranks, B, dim, dense, wd, bias, labels, gscale, X, logit=None, part_out=None):
    n_dense = dense.shape[1] if dense is not None else 0
    if X.shape[0] < B or partials.numel() < n_ranks * B * (dim + 2):
        raise MiError("fm_combine: buffer too small")
    if part_out is not None and part_out.numel() < shard_dense_part_floats(B, dim, n_dense):
        raise MiError("fm_combine: part_out too small")
    _lib.call('mi_fm_combine', _p(partials), n_ranks, B, dim, _p(dense), n_dense, _p(wd), _p(bias), _p(labels),
              gscale, _p(logit), _p(X), X.shape[1], _p(part_out), _stream())


def shard_bwd_scratch_floats(n_slots):
    return int(_lib.load().mi_shard_bwd_scratch_floats(n_slots))


def shard_owner_bwd_dense(table, local_rows, sorted_rows, perm, n_total, n_slots, X_all, n_ranks, B, row_stride, opt,
                          lr, beta1, beta2, eps, n_dense, dp, dm, dv, dense_lr_t, scratch, loss_out, lr_dev=None,
                          dense_lr_dev=None):
    """One launch: sparse backward + fused optimizer over the owner's sorted entries, and the dense
    side (sum of all ranks' combine partials, Adam on dp, mean loss)."""
    if min(sorted_rows.numel(), perm.numel()) < n_slots or scratch.numel() < shard_bwd_scratch_floats(n_slots) \
            or X_all.shape[0] < n_ranks * row_stride:
        raise MiError("shard_owner_bwd_dense: buffer too small")
    _lib.call('mi_shard_owner_bwd_dense', table.ref(), local_rows, _p(sorted_rows), _p(perm), _p(n_total), n_slots,
              _p(X_all), X_all.shape[1], n_ranks, B, row_stride, opt, lr, _p(lr_dev), beta1, beta2, eps, n_dense,
              _p(dp), _p(dm), _p(dv), dense_lr_t, _p(dense_lr_dev), _p(scratch), _p(loss_out), _stream())


def fm_dense_step_rows(X, dim, n_dense, dp, dm, dv, lr_t, beta1, beta2, eps, loss_out, lr_dev=None):
    _lib.call('mi_fm_dense_step_rows', _p(X), X.shape[1], dim, n_dense, X.shape[0], _p(dp), _p(dm), _p(dv), lr_t,
              _p(lr_dev), beta1, beta2, eps, _p(loss_out), _stream())


# ----------------------------------------------------------------------------- sharded embedding bags
def shard_bag_slots_workspace_bytes(n_bags, n_parts):
    return int(_lib.load().mi_shard_bag_slots_workspace_bytes(n_bags, n_parts))


def shard_bag_slots(toffs, T, ids, bag_len, n_bags, n_parts, owner, n_ranks, cap, local_ids, row_of_bag, bag_of_row,
                    cnt, workspace, err=None):
    """Which bags rank ``owner`` contributes to (owner view, ``ids`` = all ranks' ids) or, with
    owner=-1, which owners contribute to this rank's bags (requester view); see mi_shard_bag_slots."""
    _need_cuda(toffs, ids, local_ids, row_of_bag, bag_of_row, cnt, workspace, err)
    _lib.call('mi_shard_bag_slots', _p(toffs), T, _p(ids), _idx_bits(ids), bag_len, n_bags, n_parts, owner, n_ranks, cap,
              _p(local_ids), _p(row_of_bag), _p(bag_of_row), _p(cnt), _p(workspace), workspace.numel(), _p(err),
              _stream())


def shard_bag_bucket_workspace_bytes(n_bags, n_ranks):
    return int(_lib.load().mi_shard_bag_bucket_workspace_bytes(n_bags, n_ranks))


def shard_bag_bucket(toffs, T, ids, bag_len, n_bags, rank, n_ranks, cap, cap_ids, row_of_bag, cnt, send_ids, send_off,
                     workspace, err=None):
    """This rank's ids bucketed by owner, as the owners' local rows in CSR blocks; see mi_shard_bag_bucket."""
    _need_cuda(toffs, ids, row_of_bag, cnt, send_ids, send_off, workspace, err)
    _lib.call('mi_shard_bag_bucket', _p(toffs), T, _p(ids), _idx_bits(ids), bag_len, n_bags, rank, n_ranks, cap, cap_ids,
              _p(row_of_bag), _p(cnt), _p(send_ids), _p(send_off), _p(workspace), workspace.numel(), _p(err), _stream())


def shard_bag_gather(weights, ltoffs, T, local_ids, bag_len, n_bags, n_parts, cap, bag_of_row, cnt, out_rows):
    _lib.call('mi_shard_bag_gather', _p(weights), _p(ltoffs), T, weights.shape[1], _p(local_ids), bag_len, n_bags,
              n_parts, cap, _p(bag_of_row), _p(cnt), _p(out_rows), _stream())


def shard_bag_combine(rows, row_of_bag, n_parts, n_bags, T, dim, out):
    _lib.call('mi_shard_bag_combine', _p(rows), _p(row_of_bag), n_parts, n_bags, T, dim, _p(out), out.stride(0),
              _stream())


def shard_bag_pack(grad, T, dim, bag_of_row, cnt, n_parts, cap, rows_out):
    _lib.call('mi_shard_bag_pack', _p(grad), grad.stride(0), T, dim, _p(bag_of_row), _p(cnt), n_parts, cap,
              _p(rows_out), _stream())


def bwd_apply_rows(plan, grad_rows, row_of_bag, weights, opt, lr, state=None, eps=1e-8):
    """Sparse update with the bag gradients given as compact rows (grad_rows[row_of_bag[bag]])."""
    _lib.call('mi_embag_bwd_apply_rows', plan.ref(), _p(grad_rows), _p(row_of_bag), weights.shape[1], opt, _p(weights),
              _p(state), lr, eps, _stream())


# ----------------------------------------------------------------------------- bf16 MFMA contractions
MI_EPI_SLAB, MI_EPI_BIAS_ACT, MI_EPI_CROSS_FWD, MI_EPI_MASK, MI_EPI_CROSS_BWD = 0, 1, 2, 3, 4
MI_EPI_RELU = 1


def _ld(t):
    return t.stride(0) if t is not None else 0


def gemm_bf16(A, B, mode, M=None, N=None, K=None, a_rm=False, b_rm=False, k_splits=1, flags=0, out=None, bias=None,
              in0=None, in1=None, out1=None, bias_rows=0):
    """C[m][n] = sum_k A(m,k) B(n,k) on the matrix cores, then the epilogue ``mode`` (MI_EPI_*).
    A is [M][K] (or [K][M] with ``a_rm``), B is [N][K] (or [K][N] with ``b_rm``), both bf16 with
    contiguous rows; M, N, K default to the tensors' shapes (pass them to use a sub-block).  All
    epilogue tensors are 2-D with contiguous rows: bf16 except ``out`` under MI_EPI_SLAB
    ([k_splits, M, N] fp32) and ``bias`` [N] fp32."""
    _need_cuda(A, B, out, bias, in0, in1, out1)
    for t in (A, B, in0, in1, out1):
        if t is not None and (t.dtype != torch.bfloat16 or t.stride(-1) != 1):
            raise MiError("gemm_bf16: operands and bf16 epilogue tensors must be bf16 with contiguous rows")
    if M is None:
        M = A.shape[1] if a_rm else A.shape[0]
    if K is None:
        K = A.shape[0] if a_rm else A.shape[1]
    if N is None:
        N = B.shape[1] if b_rm else B.shape[0]
    e = _lib.GemmEpilogue(mode=mode, flags=flags, out=_p(out), ld_out=out.stride(-2) if out is not None else 0,
                          bias=_p(bias), in0=_p(in0), ld_in0=_ld(in0), in1=_p(in1), ld_in1=_ld(in1),
                          out1=_p(out1), ld_out1=_ld(out1), bias_ld=bias.stride(0) if bias_rows else 0,
                          bias_rows=bias_rows)
    _lib.call('mi_gemm_bf16', int(a_rm), int(b_rm), M, N, K, _p(A), A.stride(0), _p(B), B.stride(0), k_splits,
              ctypes.byref(e), _stream())
    return out


def pack_dense_bf16(dense, x, col0, d_pad):
    nd = dense.shape[1] if dense is not None else 0
    _lib.call('mi_pack_dense_bf16', _p(dense), nd, x.shape[0], _p(x), x.stride(0), col0, d_pad, _stream())


def colsum_bf16(X, slices, partial, N=None):
    _lib.call('mi_colsum_bf16', _p(X), X.stride(0), X.shape[0], N or X.shape[1], slices, _p(partial),
              partial.stride(0), _stream())


def slab_adam(slab, n_slabs, n, slab_stride, w, m, v, beta1, beta2, eps, lr_t=0.0, lr_dev=None, w_bf16=None,
              grad_out=None):
    _lib.call('mi_slab_adam', _p(slab), n_slabs, n, slab_stride, _p(w), _p(m), _p(v), lr_t, _p(lr_dev), beta1, beta2,
              eps, _p(w_bf16), _p(grad_out), _stream())


def head_fwd(h, w, bias, labels, gscale, logit, loss_rows, g):
    _lib.call('mi_head_fwd', _p(h), h.stride(0), h.shape[1], _p(w), _p(bias), _p(labels), gscale, h.shape[0],
              _p(logit), _p(loss_rows), _p(g), _stream())


def head_bwd(g, w, h, dh, slices, partial):
    _lib.call('mi_head_bwd', _p(g), _p(w), _p(h), h.stride(0), h.shape[1], h.shape[0], _p(dh), dh.stride(0), slices,
              _p(partial), partial.stride(0), _stream())


def cross_input_grad(dx, u, out, N=None):
    """out (fp32) = dx[0] + sum_l dx[l+1]*u[l]; dx: L+1 bf16 tensors [M, ld], u: L of them."""
    L = len(u)
    if len(dx) != L + 1:
        raise MiError("cross_input_grad: need one more dx than u")
    P = ctypes.c_void_p * 9
    pdx = P(*([t.data_ptr() for t in dx] + [None] * (8 - L)))
    pu = P(*([t.data_ptr() for t in u] + [None] * (9 - L)))
    _lib.call('mi_cross_input_grad', pdx, pu, L, dx[0].shape[0], N or dx[0].shape[1], dx[0].stride(0), _p(out),
              out.stride(0), _stream())


def cast_bf16(src, dst):
    _lib.call('mi_cast_bf16', _p(src), src.numel(), _p(dst), _stream())


def transpose_bf16(src, dst):
    """dst[c][r] = src[r][c] for 2-D bf16 tensors with contiguous rows."""
    R, C = src.shape
    if dst.shape[0] < C or dst.shape[1] < R or src.stride(1) != 1 or dst.stride(1) != 1:
        raise MiError("transpose_bf16: dst must be [C][R] with contiguous rows")
    _lib.call('mi_transpose_bf16', _p(src), src.stride(0), R, C, _p(dst), dst.stride(0), _stream())


# ----------------------------------------------------------------------------- DIN local activation unit
def din_lau_build(E, hist, qid, lengths, A, ids_all, q_out=None, err=None):
    """A [B*L][4*dim] = [q | k | q-k | q*k], or [B*L][2*dim] = [k | q*k] (chosen by A's width)."""
    B, L = hist.shape
    _lib.call('mi_din_lau_build', _p(E), E.shape[0], E.shape[1], _p(hist), _p(qid), _idx_bits(hist), _p(lengths), B, L,
              _p(A), _p(ids_all), _p(q_out), q_out.stride(0) if q_out is not None else 0, A.shape[1], _p(err), _stream())


def din_lau_fwd_fused_ok(dim, h1, h2, L, factored=True):
    return bool(_lib.load().mi_din_lau_fwd_fused_ok(dim, h1, h2, L, int(factored)))


def din_lau_fwd_fused(E, hist, qid, lengths, W1f, Wq, b1, W2, b2, w3, A, H1, H2, a, ids_all, feat, rowbias, err=None):
    """The factored activation unit's forward in one launch (a workgroup per sample): A, H1, H2 (bf16), a (fp32),
    ids_all and feat = [pooled | q], bit-identical to din_lau_build + din_qterm_fwd + two gemm_bf16 + head_fwd +
    din_lau_pool."""
    B, L = hist.shape
    _need_cuda(E, hist, qid, lengths, W1f, Wq, b1, W2, b2, w3, A, H1, H2, a, ids_all, feat)
    _lib.call('mi_din_lau_fwd_fused', _p(E), E.shape[0], _p(hist), _p(qid), _idx_bits(hist), _p(lengths), B, L, _p(W1f),
              _p(Wq), _p(b1), _p(W2), _p(b2), _p(w3), _p(A), _p(H1), _p(H2), _p(a), _p(ids_all), _p(feat),
              feat.stride(0), _p(rowbias), _p(err), _stream())


def din_lau_bwd_part_floats():
    return int(_lib.load().mi_din_lau_bwd_part_floats())


def din_lau_bwd_fused(A, H1, H2, a, ids_all, feat, dfeat, w3, W2T, W1fT, dZ2, dZ1, dE, S, part, part_sums):
    """The data path of the factored unit's backward in one launch per sample: dZ2, dZ1, dE (the candidate's row
    still without its S Wq term), S and the partial sums of the head's / second bias' gradients."""
    B, L1 = ids_all.shape
    _need_cuda(A, H1, H2, a, ids_all, feat, dfeat, w3, W2T, W1fT, dZ2, dZ1, dE, S, part, part_sums)
    _lib.call('mi_din_lau_bwd_fused', B, L1 - 1, _p(A), _p(H1), _p(H2), _p(a), _p(ids_all), _p(feat), feat.stride(0),
              _p(dfeat), dfeat.stride(0), _p(w3), _p(W2T), _p(W1fT), _p(dZ2), _p(dZ1), _p(dE), _p(S), _p(part),
              part_sums.shape[0], _p(part_sums), _stream())


def din_lau_pool(A, a, ids_all, dim, pooled):
    B, L1 = ids_all.shape
    _lib.call('mi_din_lau_pool', _p(A), A.shape[1], _p(a), _p(ids_all), B, L1 - 1, dim, _p(pooled), pooled.stride(0),
              _stream())


def din_lau_pool_bwd(A, dpooled, ids_all, dim, da):
    B, L1 = ids_all.shape
    _lib.call('mi_din_lau_pool_bwd', _p(A), A.shape[1], _p(dpooled), dpooled.stride(0), _p(ids_all), B, L1 - 1, dim,
              _p(da), _stream())


def din_lau_scatter_fact(A, dA, a, dpooled, dq_extra, q, ids_all, dim, dE):
    B, L1 = ids_all.shape
    _lib.call('mi_din_lau_scatter_fact', _p(A), _p(dA), _p(a), _p(dpooled), dpooled.stride(0), _p(dq_extra),
              dq_extra.stride(0) if dq_extra is not None else 0, _p(q), q.stride(0), _p(ids_all), B, L1 - 1, dim, _p(dE),
              _stream())


def din_qterm_part_floats(B, H, dim):
    return int(_lib.load().mi_din_qterm_part_floats(B, H, dim))


def din_qterm_fwd(q, Wq, b1, rowbias):
    """rowbias [B][H] = b1 + q Wq^T, the per-sample term of the factored first layer."""
    B, H = rowbias.shape
    _lib.call('mi_din_qterm_fwd', _p(q), q.stride(0), _p(Wq), _p(b1), B, H, Wq.shape[1], _p(rowbias), _stream())


def din_qterm_bwd(q, S, Wq, dq_in, dq, part):
    """dq = dq_in + S Wq; part = per-32-sample partials of q^T S and of the column sums of S."""
    B, H = S.shape
    if part.numel() < din_qterm_part_floats(B, H, Wq.shape[1]):
        raise MiError("din_qterm_bwd: part too small")
    _lib.call('mi_din_qterm_bwd', _p(q), q.stride(0), _p(S), _p(Wq), B, H, Wq.shape[1], _p(dq_in),
              dq_in.stride(0) if dq_in is not None else 0, _p(dq), dq.stride(0), _p(part), _stream())


def din_w1_step(slab, n_slabs, slab_stride, dWaT, n_a, a_stride, W, m, v, beta1, beta2, eps, Wf, WfT, Wq, WqT, lr_t=0.0,
                lr_dev=None, grad_out=None):
    H = W.shape[0]
    _lib.call('mi_din_w1_step', _p(slab), n_slabs, slab_stride, _p(dWaT), n_a, a_stride, H, W.shape[1] // 4, _p(W),
              _p(m), _p(v), lr_t, _p(lr_dev), beta1, beta2, eps, _p(Wf), _p(WfT), _p(Wq), _p(WqT), _p(grad_out),
              _stream())


def din_lau_scatter(A, dA, a, dpooled, dq_extra, ids_all, dim, dE):
    B, L1 = ids_all.shape
    _lib.call('mi_din_lau_scatter', _p(A), _p(dA), _p(a), _p(dpooled), dpooled.stride(0), _p(dq_extra),
              dq_extra.stride(0) if dq_extra is not None else 0, _p(ids_all), B, L1 - 1, dim, _p(dE), _stream())


def bce_logits(logits, labels, gscale, loss_rows, g):
    _lib.call('mi_bce_logits', _p(logits), _p(labels), logits.numel(), gscale, _p(loss_rows), _p(g), _stream())


def bwd_prepare_noalloc(toffs, T, total_rows, indices, sorted_rows, perm, unique_rows=None, seg_offsets=None,
                        counts=None, offsets=None, bag_len=1):
    """mi_embag_bwd_prepare on the library's built-in sort scratch: the call allocates nothing (the
    caller passes every output buffer).  At most MI_NOALLOC_MAX_ENTRIES entries; calls that overlap
    in time on one GPU share the scratch and race."""
    _need_cuda(toffs, indices, sorted_rows, perm, unique_rows, seg_offsets, counts, offsets)
    if not indices.is_contiguous():
        raise MiError("bwd_prepare_noalloc: indices must be contiguous (a copy would allocate)")
    n = indices.numel()
    n_bags = n // max(bag_len, 1) if offsets is None else offsets.numel() - 1
    _lib.call('mi_embag_bwd_prepare_noalloc', _p(toffs), T, total_rows, _p(indices), _idx_bits(indices), _p(offsets),
              bag_len if offsets is None else 0, n_bags, n, _p(sorted_rows), _p(perm), _p(unique_rows),
              _p(seg_offsets), _p(counts), _stream())


def stage_from_host(dst, src_pinned):
    """dst (device) = src_pinned (pinned host tensor of the same byte size), copied by a kernel on the current stream
    (mi_stage_from_host): orders with compute kernels on other streams without the host's help."""
    if not dst.is_cuda or src_pinned.is_cuda or not src_pinned.is_pinned():
        raise MiError("stage_from_host: dst must be a device tensor and src a pinned host tensor")
    nbytes = dst.numel() * dst.element_size()
    if nbytes != src_pinned.numel() * src_pinned.element_size() or not dst.is_contiguous() or not src_pinned.is_contiguous():
        raise MiError("stage_from_host: contiguous tensors of the same byte size")
    _lib.call('mi_stage_from_host', _p(dst), ctypes.c_void_p(src_pinned.data_ptr()), nbytes, _stream())


# ---- whole-network entry points (the cross network, the DIN activation unit) ------------------------------
MI_CROSS_PART_X, MI_CROSS_PART_U, MI_CROSS_PART_DX, MI_CROSS_PART_DU, MI_CROSS_PART_SLAB = range(5)
(MI_DIN_PART_A, MI_DIN_PART_DA, MI_DIN_PART_H1, MI_DIN_PART_DZ1, MI_DIN_PART_H2, MI_DIN_PART_DH2,
 MI_DIN_PART_WEIGHTS) = range(7)


def gemm_wgrad_splits(n_out, k_in, m):
    """K-slices of a weight-gradient contraction [n_out][k_in] over m rows (the library's own rule)."""
    return int(_lib.load().mi_gemm_wgrad_splits(n_out, k_in, m))


def adam_param(p, keep_grad=False):
    """mi_adam_param of an object with tensors w, m, v (+ h, hT for matrices, + grad)."""
    return _lib.AdamParam(w=_p(p.w), m=_p(p.m), v=_p(p.v), w_bf16=_p(getattr(p, 'h', None)),
                          w_bf16_t=_p(getattr(p, 'hT', None)), grad_out=_p(p.grad) if keep_grad else None)


def mul_bf16(a, b, out):
    _lib.call('mi_mul_bf16', _p(a), a.stride(0), _p(b), b.stride(0), a.shape[0], a.shape[1], _p(out), out.stride(0),
              _stream())


class CrossNetwork(object):
    """mi_cross_net + its workspace for one batch size; the workspace's tensors as views."""

    def __init__(self, Ws, bs, width, B, device, keep_grads=False):
        L = len(Ws)
        if not 1 <= L <= _lib.CROSS_MAX_LAYERS:
            raise MiError("CrossNetwork: 1..%d layers" % _lib.CROSS_MAX_LAYERS)
        self.L, self.width, self.B = L, int(width), int(B)
        self.c = _lib.CrossNet(n_layers=L, width=self.width)
        self._keep = (Ws, bs)
        for l in range(L):
            self.c.W[l] = adam_param(Ws[l], keep_grads)
            self.c.b[l] = adam_param(bs[l], keep_grads)
        n = int(_lib.load().mi_cross_workspace_bytes(self.B, self.width, L))
        self.ws = torch.empty((n,), dtype=torch.uint8, device=device)

    def part(self, what, index=0):
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.call('mi_cross_workspace_part', self.B, self.width, self.L, what, index, ctypes.byref(off), ctypes.byref(nb))
        raw = self.ws[off.value:off.value + nb.value]
        if what == MI_CROSS_PART_SLAB:
            return raw.view(torch.float32)
        return raw.view(torch.bfloat16).view(self.B, self.width)

    def forward(self, x0):
        last = ctypes.c_void_p()
        _lib.call('mi_cross_fwd', ctypes.byref(self.c), _p(x0), x0.stride(0), self.B, _p(self.ws), self.ws.numel(),
                  ctypes.byref(last), _stream())
        return self.part(MI_CROSS_PART_X, self.L)

    def backward(self, x0, dx0, beta1, beta2, eps, lr_t=0.0, lr_dev=None, du_last_ready=False, apply_adam=True):
        """``apply_adam=False``: gradients only, into the parameters' ``grad`` tensors (``keep_grads`` at construction)."""
        _lib.call('mi_cross_bwd', ctypes.byref(self.c), _p(x0), x0.stride(0), self.B, _p(self.ws), self.ws.numel(),
                  1 if du_last_ready else 0, 1 if apply_adam else 0, lr_t, _p(lr_dev), beta1, beta2, eps, _p(dx0),
                  dx0.stride(0), _stream())


class DinUnit(object):
    """mi_din_lau + its workspace for one batch size."""

    def __init__(self, model, B, device):
        m = model
        self.B = int(B)
        self.c = _lib.DinLau(dim=m.dim, h1=m.h1p, h2=m.h2p, L=m.L, factored=1 if m.factored else 0,
                             W1=adam_param(m.W1, m.keep_grads), b1=adam_param(m.b1, m.keep_grads),
                             W2=adam_param(m.W2, m.keep_grads), b2=adam_param(m.b2, m.keep_grads),
                             w3=adam_param(m.w3, m.keep_grads),
                             W1f=_p(m.W1f) if m.factored else None, W1fT=_p(m.W1fT) if m.factored else None,
                             Wq=_p(m.Wq) if m.factored else None, WqT=_p(m.WqT) if m.factored else None)
        self._keep = m
        n = int(_lib.load().mi_din_lau_workspace_bytes(self.B, ctypes.byref(self.c)))
        self.ws = torch.empty((n,), dtype=torch.uint8, device=device)
        self.M = self.B * m.L

    def part(self, what):
        off, nb = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.call('mi_din_lau_workspace_part', self.B, ctypes.byref(self.c), what, ctypes.byref(off), ctypes.byref(nb))
        raw = self.ws[off.value:off.value + nb.value]
        if what == MI_DIN_PART_WEIGHTS:
            return raw.view(torch.float32)
        return raw.view(torch.bfloat16).view(self.M, -1)

    def forward(self, E, hist, qid, lengths, feat, ids_all, err):
        _need_cuda(E, hist, qid, lengths, feat, ids_all)
        if qid.dtype != hist.dtype:
            raise MiError("DinUnit: hist and qid must have the same integer type")
        _lib.call('mi_din_lau_fwd', ctypes.byref(self.c), _p(E), E.shape[0], _p(hist.contiguous()), _p(qid.contiguous()),
                  _idx_bits(hist), _p(lengths), self.B, _p(self.ws), self.ws.numel(), _p(feat), feat.stride(0),
                  _p(ids_all), _p(err), _stream())

    def backward(self, feat, dfeat, ids_all, dE, beta1, beta2, eps, lr_t=0.0, lr_dev=None):
        _lib.call('mi_din_lau_bwd', ctypes.byref(self.c), self.B, _p(self.ws), self.ws.numel(), _p(feat), feat.stride(0),
                  _p(dfeat), dfeat.stride(0), _p(ids_all), lr_t, _p(lr_dev), beta1, beta2, eps, _p(dE), _stream())
