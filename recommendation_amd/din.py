"""DIN local activation unit with a linear output layer: host side.

The reference has no DIN; its nearest analogue is the 6-id bag pooled with uniform weights
(recommendation/code/r_model.py:68-80).  Here every history item gets a learned weight:

  A      = [q | k_j | q-k_j | q*k_j]  bf16, one row per (sample, position)   mi_din_lau_build
  H1, H2 = relu(A W1^T + b1), relu(H1 W2^T + b2)                              mi_gemm_bf16
  a_j    = H2 . w3 + b3                                                       mi_head_fwd
  pooled = sum_{j < len} a_j k_j                                              mi_din_lau_pool
  logit  = Wh . [pooled | q] + bh, BCE loss                                   mi_linear_fwd, mi_bce_logits

The hidden widths 80 and 40 are padded to 128 and 64 (the contraction's 64-deep tile); padded
weights are zero and stay zero.  The backward mirrors it: mi_head_bwd, two data-gradient
contractions, three split-K weight gradients reduced in a fixed order and fed to TensorFlow-form
Adam (recommendation/code/model_trainer.py:157), mi_din_lau_scatter back to per-position gradients
of the embedding rows, then the same deterministic sort-by-row update as every other table here.
Positions beyond a sample's length are computed (their rows of A are zero) but masked out of the
pooled sum and of every gradient.

``factored`` (the default for a first hidden width up to 128): the first layer is linear in A, so
A W1^T = q (Wa+Wc)^T + k (Wb-Wc)^T + (q*k) Wd^T with W1 = [Wa | Wb | Wc | Wd].  Only [k | q*k] is
then stored per (sample, position) - half the bytes of the two largest tensors of the step - and
the q term is one bias row per sample (mi_din_qterm_fwd, mi_gemm_epilogue.bias_rows); its
gradients come from the per-sample sums of dZ (mi_din_qterm_bwd) and mi_din_w1_step assembles
d W1 = [S^T q | dZ^T K | S^T q - dZ^T K | dZ^T (Q*K)] for the Adam step on the master weights.
"""
import math

import torch

from . import ops
from .dcn import _Dense, _round_up


class DIN(object):
    def __init__(self, rows, device, dim=64, h1=80, h2=40, max_len=100, lr=1e-3, emb_lr=1e-2, beta1=0.9,
                 beta2=0.999, eps=1e-8, emb_eps=1e-8, seed=0, keep_grads=False, factored=None):
        if dim != 64:
            raise ValueError("DIN: only dim 64 is built")
        self.device = torch.device(device)
        self.rows, self.dim, self.h1, self.h2, self.L = int(rows), dim, int(h1), int(h2), int(max_len)
        self.h1p, self.h2p = _round_up(self.h1, 64), _round_up(self.h2, 64)
        self.lr, self.emb_lr, self.beta1, self.beta2, self.eps, self.emb_eps = lr, emb_lr, beta1, beta2, eps, emb_eps
        self.keep_grads = keep_grads
        # factored: the first layer as q (Wa+Wc)^T + k (Wb-Wc)^T + (q*k) Wd^T, rows [k | q*k] only (the
        # default wherever its kernels are built: first hidden width up to 128)
        self.factored = self.h1p <= 128 if factored is None else bool(factored)
        gen = torch.Generator(device='cpu')
        gen.manual_seed(seed)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.E = torch.randn((self.rows, dim), generator=gen).mul_(0.3).to(self.device)
        self.E_acc = torch.zeros((self.rows,), **f32)
        self.toffs = ops.table_offsets([self.rows], self.device)

        def padded(w, r, c):
            out = torch.zeros((r, c))
            out[:w.shape[0], :w.shape[1]] = w
            return out

        D4 = 4 * dim
        self.W1 = _Dense(padded(torch.randn((self.h1, D4), generator=gen) * math.sqrt(2.0 / D4), self.h1p, D4),
                         self.device, True)
        self.b1 = _Dense(padded(torch.randn((1, self.h1), generator=gen) * 0.05, 1, self.h1p).view(-1), self.device, False)
        self.W2 = _Dense(padded(torch.randn((self.h2, self.h1), generator=gen) * math.sqrt(2.0 / self.h1), self.h2p,
                                self.h1p), self.device, True)
        self.b2 = _Dense(padded(torch.randn((1, self.h2), generator=gen) * 0.05, 1, self.h2p).view(-1), self.device, False)
        w3 = torch.zeros(self.h2p + 4)                       # w3[h2p] followed by b3: one reduction covers both
        w3[:self.h2] = torch.randn((self.h2,), generator=gen) / math.sqrt(self.h2)
        self.w3 = _Dense(w3, self.device, False)
        # output layer on [pooled | q]: Wh[2*dim] and bh in one flat buffer (one Adam launch)
        head = torch.zeros(2 * dim + 1)
        head[:2 * dim] = torch.randn((2 * dim,), generator=gen) / math.sqrt(2 * dim)
        self.head = _Dense(head, self.device, False)
        self.sched = torch.tensor([0.0, 1.0, 1.0, 0.0], **f32)
        self.err = ops.new_error_word(self.device)
        self.steps = 0
        self._B = None
        self.overlap_sort = True   # the id sort beside the backward, on a side stream (train_step)
        self._side = None
        if self.factored:
            self._derive_w1()

    def _derive_w1(self):
        """The tensors the factored first layer reads, from the master W1 = [Wa | Wb | Wc | Wd]
        (after initialisation and load_state; a training step refreshes them in its weight-step kernel)."""
        D = self.dim
        Wa, Wb, Wc, Wd = (self.W1.w[:, i * D:(i + 1) * D] for i in range(4))
        W1f = torch.cat([Wb - Wc, Wd], dim=1).to(torch.bfloat16).contiguous()
        Wq = (Wa + Wc).contiguous()
        if getattr(self, 'W1f', None) is None:
            self.W1f, self.W1fT, self.Wq, self.WqT = W1f, W1f.t().contiguous(), Wq, Wq.t().contiguous()
        else:       # in place: the unit's parameter block (ops.DinUnit) holds these tensors' addresses
            self.W1f.copy_(W1f)
            self.W1fT.copy_(W1f.t())
            self.Wq.copy_(Wq)
            self.WqT.copy_(Wq.t())

    def dense_params(self):
        return {'lau/W1': self.W1, 'lau/b1': self.b1, 'lau/W2': self.W2, 'lau/b2': self.b2, 'lau/w3b3': self.w3,
                'out/Wb': self.head}

    def load_state(self, st):
        """Copy parameters from an object with the test oracle's fields (unpadded numpy arrays)."""
        dev = self.device
        self.E.copy_(torch.from_numpy(st.E))
        for p, w in ((self.W1, st.W1), (self.W2, st.W2)):
            p.w.zero_()
            p.w[:w.shape[0], :w.shape[1]] = torch.from_numpy(w).to(dev)
            ops.cast_bf16(p.w, p.h)
            ops.transpose_bf16(p.h, p.hT)
        for p, b in ((self.b1, st.b1), (self.b2, st.b2)):
            p.w.zero_()
            p.w[:b.shape[0]] = torch.from_numpy(b).to(dev)
        self.w3.w.zero_()
        self.w3.w[:self.h2] = torch.from_numpy(st.w3).to(dev)
        self.w3.w[self.h2p] = float(st.b3[0])
        self.head.w[:2 * self.dim] = torch.from_numpy(st.Wh).to(dev)
        self.head.w[2 * self.dim] = float(st.bh[0])
        if self.factored:
            self._derive_w1()

    # -- buffers ------------------------------------------------------------------------------
    def _splits(self, n_out, k_in, M):
        return ops.gemm_wgrad_splits(n_out, k_in, M)

    def _buffers(self, B, hist):
        if self._B == (B, hist.dtype):
            return
        L, D = self.L, self.dim
        M = B * L
        if M % 64:
            raise ops.MiError("DIN: batch * history length must be a multiple of 64 (the contraction tile depth)")
        dev = self.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        # the unit runs behind mi_din_lau_fwd / mi_din_lau_bwd; what it keeps between the two lives in that call
        # pair's workspace (the largest tensors are visible here as views)
        self.unit = ops.DinUnit(self, B, dev)
        self.A = self.unit.part(ops.MI_DIN_PART_A)
        self.dA = self.unit.part(ops.MI_DIN_PART_DA)
        self.H1, self.dZ1 = self.unit.part(ops.MI_DIN_PART_H1), self.unit.part(ops.MI_DIN_PART_DZ1)
        self.H2, self.dH2 = self.unit.part(ops.MI_DIN_PART_H2), self.unit.part(ops.MI_DIN_PART_DH2)
        self.a = self.unit.part(ops.MI_DIN_PART_WEIGHTS)
        self.ids_all = torch.empty((B, L + 1), dtype=torch.int32, device=dev)
        self.dE = torch.empty((B * (L + 1), D), **f32)
        self.feat = torch.empty((B, 2 * D), **f32)            # [pooled | q]
        self.logit, self.loss_rows, self.g = (torch.empty((B,), **f32) for _ in range(3))
        self.loss = torch.empty((1,), **f32)
        self.plan = ops.SortedPlan(self.toffs, 1, self.rows, self.ids_all.view(-1), bag_len=1, runs=False)
        self._B = (B, hist.dtype)

    def _side_stream(self):
        """The stream the id sort runs on beside the backward (None: sort in line, e.g. under graph capture)."""
        if not self.overlap_sort or torch.cuda.is_current_stream_capturing():
            return None
        if self._side is None:
            self._side = torch.cuda.Stream(self.device)
            self._ev_ids, self._ev_sorted = torch.cuda.Event(), torch.cuda.Event()
        return self._side

    # -- forward ------------------------------------------------------------------------------
    def forward(self, hist, qid, lengths, labels=None):
        """hist [B, L] ids, qid [B] ids, lengths [B] int32 -> logits [B]; pooled interest vectors in
        self.feat[:, :dim]."""
        B, L = hist.shape
        if L != self.L:
            raise ops.MiError("DIN: history width %d differs from max_len %d" % (L, self.L))
        self._buffers(B, hist)
        D = self.dim
        # gather + build, the unit's three layers over all B*L rows, weighted pooling: feat = [pooled | q]
        self.unit.forward(self.E, hist, qid, lengths, self.feat, self.ids_all, self.err)
        ops.linear_fwd(self.feat, self.head.w[:2 * D].view(2 * D, 1), self.head.w[2 * D:], out=self.logit.view(B, 1))
        if labels is not None:
            ops.bce_logits(self.logit, labels, 1.0 / B, self.loss_rows, self.g)
        return self.logit

    # -- one training step --------------------------------------------------------------------
    def train_step(self, hist, qid, lengths, labels, check_ids=False):
        B = hist.shape[0]
        D = self.dim
        self.forward(hist, qid, lengths, labels)
        # The sort of the step's row ids needs nothing but ids_all, which the forward has just written: it runs on
        # a side stream beside the output layer and the unit's backward (nine small launches that leave most of
        # the chip idle) and is joined in front of the table update.
        side = self._side_stream()
        if side is not None:
            main = torch.cuda.current_stream()
            self._ev_ids.record(main)
            side.wait_event(self._ev_ids)
            with torch.cuda.stream(side):
                self.plan.prepare(self.ids_all.view(-1))
                self._ev_sorted.record(side)
        ops.reduce_sum(self.loss_rows, 1.0 / B, out=self.loss)
        ops.adam_schedule_step(self.sched, self.lr, self.beta1, self.beta2)
        # output layer
        dfeat, dW, db = ops.linear_bwd(self.feat, self.head.w[:2 * D].view(2 * D, 1), self.g.view(B, 1))
        self.head.grad[:2 * D] = dW.view(-1)
        self.head.grad[2 * D:] = db
        ops.adam_tf_dense(self.head.w, self.head.m, self.head.v, self.head.grad, 0.0, self.beta1, self.beta2, self.eps,
                          lr_dev=self.sched)
        # the unit, output side first: data gradients with the weights the forward saw, every weight step, and the
        # per-position gradients of the gathered rows
        self.unit.backward(self.feat, dfeat, self.ids_all, self.dE, self.beta1, self.beta2, self.eps, lr_dev=self.sched)
        if side is not None:
            torch.cuda.current_stream().wait_event(self._ev_sorted)
        else:
            self.plan.prepare(self.ids_all.view(-1))
        ops.bwd_apply_adagrad_rowwise(self.plan, self.dE, self.E, self.E_acc, self.emb_lr, self.emb_eps, grad_ld=D)
        self.steps += 1
        if check_ids:
            ops.check_error_word(self.err, "DIN ids")
        return self.loss

    def flops_per_sample(self):
        """Multiply-adds x2 of the unit's contractions in one training step, per sample (padded widths)."""
        per_item = (2 if self.factored else 4) * self.dim * self.h1p + self.h1p * self.h2p + self.h2p
        return 3 * 2 * per_item * self.L
