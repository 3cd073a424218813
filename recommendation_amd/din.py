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
        if self.factored:
            self._derive_w1()

    def _derive_w1(self):
        """The tensors the factored first layer reads, from the master W1 = [Wa | Wb | Wc | Wd]
        (after initialisation and load_state; a training step refreshes them in its weight-step kernel)."""
        D = self.dim
        Wa, Wb, Wc, Wd = (self.W1.w[:, i * D:(i + 1) * D] for i in range(4))
        self.W1f = torch.cat([Wb - Wc, Wd], dim=1).to(torch.bfloat16).contiguous()
        self.W1fT = self.W1f.t().contiguous()
        self.Wq = (Wa + Wc).contiguous()
        self.WqT = self.Wq.t().contiguous()

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
        tiles = -(-n_out // 128) * -(-k_in // 128)
        t64 = M // 64
        s = max(1, min(512 // max(tiles, 1), t64))
        while s > 1 and -(-t64 // -(-t64 // s)) != s:
            s -= 1
        return s

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
        ac = (2 if self.factored else 4) * D
        self.A = torch.empty((M, ac), **bf)
        self.dA = torch.empty((M, ac), **bf)
        if self.factored:
            self.rowbias = torch.empty((B, self.h1p), **f32)
            self.S = torch.empty((B, self.h1p), **f32)            # per-sample sums of dZ1
            self.dq = torch.empty((B, D), **f32)
            self.qpart = torch.empty((ops.din_qterm_part_floats(B, self.h1p, D),), **f32)
            self.gWf = torch.empty((self.h1p, 2 * D), **f32)
            self.gq = torch.empty((D * self.h1p + self.h1p,), **f32)   # q^T S (transposed d Wa), then d b1
        self.H1, self.dZ1 = torch.empty((M, self.h1p), **bf), torch.empty((M, self.h1p), **bf)
        self.H2, self.dH2 = torch.empty((M, self.h2p), **bf), torch.empty((M, self.h2p), **bf)
        self.a, self.da = torch.empty((M,), **f32), torch.empty((M,), **f32)
        self.ids_all = torch.empty((B, L + 1), dtype=torch.int32, device=dev)
        self.dE = torch.empty((B * (L + 1), D), **f32)
        self.feat = torch.empty((B, 2 * D), **f32)            # [pooled | q]
        self.logit, self.loss_rows, self.g = (torch.empty((B,), **f32) for _ in range(3))
        self.loss = torch.empty((1,), **f32)
        self.split = {(self.h1p, ac): self._splits(self.h1p, ac, M),
                      (self.h2p, self.h1p): self._splits(self.h2p, self.h1p, M)}
        self.vec_slices = max(1, min(1024, M // 64))
        n_slab = max(max(s * k[0] * k[1] for k, s in self.split.items()), self.vec_slices * (self.h1p + 4))
        self.slab = torch.empty((n_slab,), **f32)
        self.plan = ops.SortedPlan(self.toffs, 1, self.rows, self.ids_all.view(-1), bag_len=1, runs=False)
        self._B = (B, hist.dtype)

    # -- forward ------------------------------------------------------------------------------
    def forward(self, hist, qid, lengths, labels=None):
        """hist [B, L] ids, qid [B] ids, lengths [B] int32 -> logits [B]; pooled interest vectors in
        self.feat[:, :dim]."""
        B, L = hist.shape
        if L != self.L:
            raise ops.MiError("DIN: history width %d differs from max_len %d" % (L, self.L))
        self._buffers(B, hist)
        D = self.dim
        ops.din_lau_build(self.E, hist, qid, lengths, self.A, self.ids_all, q_out=self.feat[:, D:], err=self.err)
        if self.factored:
            # the q term of the first layer, once per sample (fp32), enters as a bias row per L positions
            ops.din_qterm_fwd(self.feat[:, D:], self.Wq, self.b1.w, self.rowbias)
            ops.gemm_bf16(self.A, self.W1f, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=self.H1, bias=self.rowbias,
                          bias_rows=L)
        else:
            ops.gemm_bf16(self.A, self.W1.h, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=self.H1, bias=self.b1.w)
        ops.gemm_bf16(self.H1, self.W2.h, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=self.H2, bias=self.b2.w)
        ops.head_fwd(self.H2, self.w3.w[:self.h2p], self.w3.w[self.h2p:self.h2p + 1], None, 0.0, self.a, None, None)
        ops.din_lau_pool(self.A, self.a, self.ids_all, D, self.feat)
        ops.linear_fwd(self.feat, self.head.w[:2 * D].view(2 * D, 1), self.head.w[2 * D:], out=self.logit.view(B, 1))
        if labels is not None:
            ops.bce_logits(self.logit, labels, 1.0 / B, self.loss_rows, self.g)
        return self.logit

    # -- one training step --------------------------------------------------------------------
    def _apply(self, p, n_slabs, n, stride):
        ops.slab_adam(self.slab, n_slabs, n, stride, p.w, p.m, p.v, self.beta1, self.beta2, self.eps,
                      lr_dev=self.sched, w_bf16=p.h, grad_out=p.grad if self.keep_grads else None)

    def _weight_step(self, p, dY, X):
        n_out, k_in = p.w.shape
        s = self.split[(n_out, k_in)]
        ops.gemm_bf16(dY, X, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=n_out, N=k_in, K=dY.shape[0], k_splits=s,
                      out=self.slab[:s * n_out * k_in].view(s, n_out, k_in))
        self._apply(p, s, n_out * k_in, n_out * k_in)
        ops.transpose_bf16(p.h, p.hT)

    def _bias_step(self, p, dY):
        n = p.w.numel()
        part = self.slab[:self.vec_slices * n].view(self.vec_slices, n)
        ops.colsum_bf16(dY, self.vec_slices, part, N=n)
        self._apply(p, self.vec_slices, n, n)

    def train_step(self, hist, qid, lengths, labels, check_ids=False):
        B = hist.shape[0]
        D = self.dim
        self.forward(hist, qid, lengths, labels)
        ops.reduce_sum(self.loss_rows, 1.0 / B, out=self.loss)
        ops.adam_schedule_step(self.sched, self.lr, self.beta1, self.beta2)
        # output layer
        dfeat, dW, db = ops.linear_bwd(self.feat, self.head.w[:2 * D].view(2 * D, 1), self.g.view(B, 1))
        self.head.grad[:2 * D] = dW.view(-1)
        self.head.grad[2 * D:] = db
        ops.adam_tf_dense(self.head.w, self.head.m, self.head.v, self.head.grad, 0.0, self.beta1, self.beta2, self.eps,
                          lr_dev=self.sched)
        # the unit, output side first; data gradients with the OLD weights, then the weight steps
        ops.din_lau_pool_bwd(self.A, dfeat, self.ids_all, D, self.da)
        part = self.slab[:self.vec_slices * (self.h2p + 4)].view(self.vec_slices, self.h2p + 4)
        part.zero_()
        ops.head_bwd(self.da, self.w3.w[:self.h2p], self.H2, self.dH2, self.vec_slices, part)
        self._apply(self.w3, self.vec_slices, self.h2p + 4, self.h2p + 4)
        ops.gemm_bf16(self.dH2, self.W2.hT, ops.MI_EPI_MASK, out=self.dZ1, in0=self.H1)
        self._weight_step(self.W2, self.dH2, self.H1)
        self._bias_step(self.b2, self.dH2)
        if self.factored:
            self._first_layer_backward_factored(B, dfeat)
        else:
            ops.gemm_bf16(self.dZ1, self.W1.hT, ops.MI_EPI_MASK, out=self.dA)
            self._weight_step(self.W1, self.dZ1, self.A)
            self._bias_step(self.b1, self.dZ1)
            # embedding rows: per-position gradients, then sort by row + fixed-order sums + row-wise Adagrad
            ops.din_lau_scatter(self.A, self.dA, self.a, dfeat, dfeat[:, D:], self.ids_all, D, self.dE)
        self.plan.prepare(self.ids_all.view(-1))
        ops.bwd_apply_adagrad_rowwise(self.plan, self.dE, self.E, self.E_acc, self.emb_lr, self.emb_eps, grad_ld=D)
        self.steps += 1
        if check_ids:
            ops.check_error_word(self.err, "DIN ids")
        return self.loss

    def _first_layer_backward_factored(self, B, dfeat):
        """dZ1 -> gradients of [k | q*k], of q, of b1 and of the four blocks of W1, then the fused
        weight step; data gradients are formed with the weights as the forward saw them."""
        D, H = self.dim, self.h1p
        q = self.feat[:, D:]
        # S[b] = sum of dZ1 over the sample's positions: the column sums with one slice per sample
        ops.colsum_bf16(self.dZ1, B, self.S, N=H)
        # d q += S Wq; per 32 samples a partial of q^T S (-> d Wa) and of the column sums of S (-> d b1)
        ops.din_qterm_bwd(q, self.S, self.Wq, dfeat[:, D:], self.dq, self.qpart)
        stride = D * H + H
        chunks = self.qpart.numel() // stride
        ops.slab_adam(self.qpart, chunks, stride, stride, None, None, None, 0.0, 0.0, 0.0, grad_out=self.gq)   # chunks -> one
        ops.slab_adam(self.gq[D * H:], 1, H, H, self.b1.w, self.b1.m, self.b1.v, self.beta1, self.beta2, self.eps,
                      lr_dev=self.sched, grad_out=self.b1.grad if self.keep_grads else None)
        ops.gemm_bf16(self.dZ1, self.W1fT, ops.MI_EPI_MASK, out=self.dA)
        s = self.split[(H, 2 * D)]
        ops.gemm_bf16(self.dZ1, self.A, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=H, N=2 * D, K=self.dZ1.shape[0],
                      k_splits=s, out=self.slab[:s * H * 2 * D].view(s, H, 2 * D))
        ops.slab_adam(self.slab, s, H * 2 * D, H * 2 * D, None, None, None, 0.0, 0.0, 0.0, grad_out=self.gWf)   # slabs -> one
        ops.din_w1_step(self.gWf, 1, H * 2 * D, self.gq, 1, stride, self.W1.w, self.W1.m, self.W1.v, self.beta1,
                        self.beta2, self.eps, self.W1f, self.W1fT, self.Wq, self.WqT, lr_dev=self.sched,
                        grad_out=self.W1.grad if self.keep_grads else None)
        ops.din_lau_scatter_fact(self.A, self.dA, self.a, dfeat, self.dq, q, self.ids_all, D, self.dE)

    def flops_per_sample(self):
        """Multiply-adds x2 of the unit's contractions in one training step, per sample (padded widths)."""
        per_item = (2 if self.factored else 4) * self.dim * self.h1p + self.h1p * self.h2p + self.h2p
        return 3 * 2 * per_item * self.L
