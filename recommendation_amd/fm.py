"""Factorisation machine on Criteo-shaped input: host side of the FM hot path.

The reference has no FM (recommendation/code/r_model.py:24-140 is a linear network); this
is the extension the benchmark definition names.  It reuses the colour model's building
blocks - row gather, sort-by-row sparse backward, TensorFlow-form Adam
(recommendation/code/model_trainer.py:157) - on 26 large tables.

One training step is five entry points of the C ABI and no torch arithmetic:
  mi_fm_fwd            gather + pairwise term in registers + BCE loss rows + dL/dlogit
  mi_embag_bwd_prepare stable sort of (row, position), run detection
  mi_fm_bwd_apply      per distinct row: sum_i g_i*(S_i - v) and the fused optimizer
  mi_fm_bwd_dense      gradients of the dense-feature weights and the bias
  mi_adam_tf_dense     their update; mi_reduce_sum gives the mean loss
All buffers are allocated once per batch shape, so a step allocates nothing.
"""
import torch

from . import ops

_OPT = {'sgd': ops.MI_OPT_SGD, 'adagrad_rowwise': ops.MI_OPT_ADAGRAD_ROWWISE, 'adam_tf': ops.MI_OPT_ADAM_TF}


class FactorizationMachine(object):
    def __init__(self, rows_per_table, dim, n_dense, device, opt='adam_tf', lr=1e-3, beta1=0.9, beta2=0.999,
                 eps=1e-8, seed=0, init_scale=0.05):
        if opt not in _OPT:
            raise ValueError("opt must be one of %s" % sorted(_OPT))
        self.device = torch.device(device)
        self.rows_per_table = [int(r) for r in rows_per_table]
        self.T, self.dim, self.n_dense = len(self.rows_per_table), int(dim), int(n_dense)
        self.total_rows = sum(self.rows_per_table)
        self.toffs = ops.table_offsets(self.rows_per_table, self.device)
        self.opt, self.lr, self.beta1, self.beta2, self.eps = opt, lr, beta1, beta2, eps
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.V = torch.randn((self.total_rows, self.dim), generator=gen, **f32).mul_(init_scale)
        self.w1 = torch.randn((self.total_rows,), generator=gen, **f32).mul_(init_scale)
        # dense side: wd[n_dense] followed by the bias, one flat buffer so one Adam launch covers both
        self.dp = torch.zeros((self.n_dense + 1,), **f32)
        self.dp[:self.n_dense] = torch.randn((self.n_dense,), generator=gen, **f32).mul_(init_scale)
        self.dm, self.dv = torch.zeros_like(self.dp), torch.zeros_like(self.dp)
        self.state = []
        if opt == 'adagrad_rowwise':
            self.state = [torch.zeros((self.total_rows,), **f32), torch.zeros((self.total_rows,), **f32)]
        elif opt == 'adam_tf':
            self.state = [torch.zeros_like(self.V), torch.zeros_like(self.V), torch.zeros_like(self.w1),
                          torch.zeros_like(self.w1)]
        self._schedule = ops.TFAdamSchedule(lr, beta1, beta2)
        self.err = ops.new_error_word(self.device)
        self._buf_B = None

    # -- views ------------------------------------------------------------------------------
    @property
    def wd(self):
        return self.dp[:self.n_dense]

    @property
    def bias(self):
        return self.dp[self.n_dense:]

    def load_state(self, st):
        """Copy parameters from an object with numpy V, w1, wd, bias (the test oracle's state)."""
        self.V.copy_(torch.from_numpy(st.V))
        self.w1.copy_(torch.from_numpy(st.w1))
        self.dp[:self.n_dense].copy_(torch.from_numpy(st.wd))
        self.dp[self.n_dense] = float(st.bias)

    def state_bytes(self):
        n = self.V.numel() + self.w1.numel()
        return 4 * (n + sum(s.numel() for s in self.state))

    # -- buffers ----------------------------------------------------------------------------
    def _buffers(self, ids):
        B = ids.shape[0]
        key = (B, ids.dtype)
        if self._buf_B != key:
            f32 = dict(dtype=torch.float32, device=self.device)
            self.buf = dict(logit=torch.empty((B,), **f32), S=torch.empty((B, self.dim), **f32),
                            loss_rows=torch.empty((B,), **f32), g=torch.empty((B,), **f32))
            self.dgrad = torch.empty((self.n_dense + 1,), **f32)
            self.loss = torch.empty((1,), **f32)
            self.plan = ops.SortedPlan(self.toffs, self.T, self.total_rows, ids, bag_len=1)
            self._buf_B = key
        return self.buf

    # -- steps ------------------------------------------------------------------------------
    def forward(self, ids, dense, labels=None):
        """logits [B]; with labels also fills loss rows and dL/dlogit of the mean BCE loss."""
        buf = self._buffers(ids)
        B = ids.shape[0]
        return ops.fm_fwd(self.V, self.w1, self.toffs, ids, dense if self.n_dense else None,
                          self.wd if self.n_dense else None, self.bias, labels, 1.0 / B, self.err, out=buf)

    def train_step(self, ids, dense, labels, check_ids=False):
        """One pass of the hot path; returns the mean loss as a 1-element device tensor."""
        B = ids.shape[0]
        r = self.forward(ids, dense, labels)
        self.plan.prepare(ids)
        lr_t = self._schedule.step()
        lr_sparse = lr_t if self.opt == 'adam_tf' else self.lr
        ops.fm_bwd_apply(self.plan, r['g'], r['S'], self.V, self.w1, _OPT[self.opt], self.state, lr_sparse,
                         self.beta1, self.beta2, self.eps)
        ops.fm_bwd_dense(r['g'], dense if self.n_dense else None, out=self.dgrad)
        ops.adam_tf_dense(self.dp, self.dm, self.dv, self.dgrad, lr_t, self.beta1, self.beta2, self.eps)
        ops.reduce_sum(r['loss_rows'], 1.0 / B, out=self.loss)
        if check_ids:
            ops.check_error_word(self.err, "FM ids")
        return self.loss
