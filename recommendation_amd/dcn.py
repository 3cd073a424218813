"""DCN-v2 on Criteo-shaped input: host side of the bf16 matrix-core path.

The reference has no DCN (recommendation/code/r_model.py:24-58 is a linear network trained with
tf.train.AdamOptimizer, recommendation/code/model_trainer.py:157); this is the interaction model
the benchmark definition adds on top of the same building blocks:

  x0       = [ 26 embedding rows (mi_embag_fwd, written as bf16) | 13 dense features | zero pad ]
  x_{l+1}  = x0 * (W_l x_l + b_l) + x_l        3 full-rank cross layers   (mi_gemm_bf16, fused epilogue)
  h_{i+1}  = relu(W_i h_i + b_i)               MLP                        (mi_gemm_bf16, fused epilogue)
  logit    = w . h + b, BCE loss                                          (mi_head_fwd)

The input width 26*16+13 = 429 is padded to 448 (a multiple of the 64-deep contraction tile);
padded weights and activations are zero and stay zero.  Activations and the weights' working
copies are bf16, accumulation and the master weights fp32.  The backward never writes a
transposed tensor: dX = dY*W and dW = dY^T*X read their operands as they lie in memory through
the transposing LDS read.  dW is produced as split-K fp32 slabs, which one kernel per parameter
sums in a fixed order, feeds to TensorFlow-form Adam and re-casts to bf16.  The embedding side is
the same deterministic sort-by-row scatter and fused row-wise Adagrad as everywhere else.
"""
import math

import torch

from . import ops


def _round_up(x, m):
    return -(-x // m) * m


class _Dense(object):
    """One dense parameter: fp32 master + Adam state (+ bf16 working copy for matrices)."""

    def __init__(self, value, device, bf16_copy):
        self.w = value.to(device=device, dtype=torch.float32).contiguous()
        self.m, self.v = torch.zeros_like(self.w), torch.zeros_like(self.w)
        self.grad = torch.zeros_like(self.w)
        self.h = self.w.to(torch.bfloat16) if bf16_copy else None
        # the same matrix the other way round, [in][out]: what the data-gradient contraction reads
        self.hT = self.h.t().contiguous() if bf16_copy else None


class DCNv2(object):
    def __init__(self, rows_per_table, dim, n_dense, device, n_cross=3, mlp=(1024, 1024, 512), lr=1e-3, emb_lr=1e-2,
                 beta1=0.9, beta2=0.999, eps=1e-8, emb_eps=1e-8, seed=0, keep_grads=False, embeddings=True,
                 grads_only=False, loss_ranks=1):
        """``embeddings=False``: no table here - the step is fed the pooled rows [B][T*dim] (a sharded table's
        output) and leaves dL/d pooled in ``self.dx0``.  ``grads_only=True``: the backward stores every dense
        gradient in ``self.grad_arena`` and changes no parameter; ``apply_dense()`` applies them (a data-parallel
        step adds the ranks' arenas in between).  ``loss_ranks``: ranks that share the global batch - the loss
        gradient is scaled by 1 / (B * loss_ranks), so the SUM of the ranks' gradients is the global mean's."""
        self.device = torch.device(device)
        self.rows_per_table = [int(r) for r in rows_per_table]
        self.T, self.dim, self.n_dense = len(self.rows_per_table), int(dim), int(n_dense)
        self.total_rows = sum(self.rows_per_table)
        self.toffs = ops.table_offsets(self.rows_per_table, self.device)
        self.d = self.T * self.dim + self.n_dense
        self.dp = _round_up(self.d, 64)
        self.n_cross, self.mlp = int(n_cross), [int(w) for w in mlp]
        for w in self.mlp:
            if w % 64:
                raise ValueError("MLP widths must be multiples of 64")
        self.lr, self.emb_lr, self.beta1, self.beta2, self.eps, self.emb_eps = lr, emb_lr, beta1, beta2, eps, emb_eps
        self.embeddings, self.grads_only, self.loss_ranks = bool(embeddings), bool(grads_only), int(loss_ranks)
        self.keep_grads = keep_grads or self.grads_only
        gen = torch.Generator(device='cpu')
        gen.manual_seed(seed)
        f32 = dict(dtype=torch.float32, device=self.device)
        if self.embeddings:
            self.V = torch.randn((self.total_rows, self.dim), generator=gen).mul_(0.05).to(self.device)
            self.V_acc = torch.zeros((self.total_rows,), **f32)
        else:
            self.V = self.V_acc = None
        d, dp = self.d, self.dp

        def padded(w, rows, cols):
            out = torch.zeros((rows, cols))
            out[:w.shape[0], :w.shape[1]] = w
            return out

        self.Wc = [_Dense(padded(torch.randn((d, d), generator=gen) / math.sqrt(d) * 0.5, dp, dp), self.device, True)
                   for _ in range(self.n_cross)]
        self.bc = [_Dense(padded(torch.randn((1, d), generator=gen) * 0.05, 1, dp).view(-1), self.device, False)
                   for _ in range(self.n_cross)]
        widths = [d] + self.mlp
        wpad = [dp] + self.mlp
        self.Wm = [_Dense(padded(torch.randn((widths[i + 1], widths[i]), generator=gen) * math.sqrt(2.0 / widths[i]),
                                 wpad[i + 1], wpad[i]), self.device, True) for i in range(len(self.mlp))]
        self.bm = [_Dense(torch.randn((widths[i + 1],), generator=gen) * 0.05, self.device, False)
                   for i in range(len(self.mlp))]
        # output layer: w[K] and the bias in one buffer so that one reduction covers both
        K = self.mlp[-1]
        wb = torch.zeros(K + 4)
        wb[:K] = torch.randn((K,), generator=gen) / math.sqrt(K)
        self.out = _Dense(wb, self.device, False)
        self._make_arenas()
        self.sched = torch.tensor([0.0, 1.0, 1.0, 0.0], **f32)
        self.err = ops.new_error_word(self.device)
        self.steps = 0
        self._graphs = {}
        self._B = None

    # -- parameters ---------------------------------------------------------------------------
    def _make_arenas(self):
        """Every dense parameter's w, m, v and gradient as views of four flat tensors: one all-reduce covers all
        gradients, one Adam launch all parameters."""
        ps = list(self.dense_params().values())
        n = sum(p.w.numel() for p in ps)
        f32 = dict(dtype=torch.float32, device=self.device)
        self.w_arena, self.m_arena, self.v_arena, self.grad_arena = (torch.zeros((n,), **f32) for _ in range(4))
        off = 0
        for p in ps:
            k, shape = p.w.numel(), p.w.shape
            for name, arena in (('w', self.w_arena), ('m', self.m_arena), ('v', self.v_arena), ('grad', self.grad_arena)):
                view = arena[off:off + k].view(shape)
                view.copy_(getattr(p, name))
                setattr(p, name, view)
            off += k

    def dense_params(self):
        """name -> _Dense, names as in the test oracle."""
        out = {}
        for l in range(self.n_cross):
            out['cross%d/W' % l], out['cross%d/b' % l] = self.Wc[l], self.bc[l]
        for i in range(len(self.mlp)):
            out['mlp%d/W' % i], out['mlp%d/b' % i] = self.Wm[i], self.bm[i]
        out['out/wb'] = self.out
        return out

    def load_state(self, st):
        """Copy parameters from an object with the test oracle's fields (unpadded numpy arrays)."""
        d = self.d
        if self.embeddings:
            self.V.copy_(torch.from_numpy(st.V))
        for l in range(self.n_cross):
            self.Wc[l].w.zero_()
            self.Wc[l].w[:d, :d] = torch.from_numpy(st.Wc[l]).to(self.device)
            self.bc[l].w.zero_()
            self.bc[l].w[:d] = torch.from_numpy(st.bc[l]).to(self.device)
        for i in range(len(self.mlp)):
            W = torch.from_numpy(st.Wm[i]).to(self.device)
            self.Wm[i].w.zero_()
            self.Wm[i].w[:, :W.shape[1]] = W
            self.bm[i].w.copy_(torch.from_numpy(st.bm[i]))
        K = self.mlp[-1]
        self.out.w.zero_()
        self.out.w[:K] = torch.from_numpy(st.w_out).to(self.device)
        self.out.w[K] = float(st.b_out[0])
        for p in self.Wc + self.Wm:
            ops.cast_bf16(p.w, p.h)
            ops.transpose_bf16(p.h, p.hT)

    # -- buffers ------------------------------------------------------------------------------
    def _splits(self, n_out, k_in, B):
        """K-slices for a dW contraction [n_out, k_in] over the batch: enough workgroups to fill the
        chip, each slice a whole number of 64-sample tiles."""
        return ops.gemm_wgrad_splits(n_out, k_in, B)

    def _buffers(self, B, ids):
        if self._B == (B, ids.dtype):
            return
        if B % 64:
            raise ops.MiError("DCNv2: batch size must be a multiple of 64 (the contraction tile depth)")
        dev, dp = self.device, self.dp
        bf = dict(dtype=torch.bfloat16, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        # the cross network runs behind mi_cross_fwd / mi_cross_bwd; its activations and gradients live in that
        # call pair's workspace and are used here as views (the MLP reads x_L and writes dx_L / du_{L-1} in place)
        L = self.n_cross
        self.cross = ops.CrossNetwork(self.Wc, self.bc, dp, B, dev, self.keep_grads) if L else None
        self.x = [torch.zeros((B, dp), **bf)] + [self.cross.part(ops.MI_CROSS_PART_X, l) for l in range(1, L + 1)]
        self.u = [self.cross.part(ops.MI_CROSS_PART_U, l) for l in range(L)]
        self.h = [torch.empty((B, w), **bf) for w in self.mlp]
        self.dz = [torch.empty((B, w), **bf) for w in self.mlp]
        if L:
            self.dx = [self.cross.part(ops.MI_CROSS_PART_DX, l) for l in range(L + 1)]     # dx[l] = dL/dx_l
            self.du = [self.cross.part(ops.MI_CROSS_PART_DU, i) for i in range(2)]
        else:
            self.dx = [torch.empty((B, dp), **bf)]
            self.du = []
        self.dx0 = torch.empty((B, dp), **f32)
        self.logit, self.loss_rows, self.g = (torch.empty((B,), **f32) for _ in range(3))
        self.loss = torch.empty((1,), **f32)
        shapes = [(dp, dp)] * self.n_cross + [tuple(p.w.shape) for p in self.Wm]
        self.split = {s: self._splits(s[0], s[1], B) for s in set(shapes)}
        self.vec_slices = max(1, min(256, B // 64))
        widest = max([dp] + self.mlp + [self.mlp[-1] + 4])
        n_slab = max(max(self.split[s] * s[0] * s[1] for s in shapes), self.vec_slices * widest)
        self.slab = torch.empty((n_slab,), **f32)
        self.plan = ops.SortedPlan(self.toffs, self.T, self.total_rows, ids, bag_len=1, runs=False) \
            if self.embeddings else None
        self._B = (B, ids.dtype)

    # -- forward ------------------------------------------------------------------------------
    def forward(self, ids, dense, labels=None, pooled=None):
        """logits [B] (fp32); with labels also the loss rows and dL/dlogit of the mean BCE loss.  Without a table
        of its own (``embeddings=False``) the rows come as ``pooled`` [B][T*dim] fp32."""
        B = ids.shape[0]
        self._buffers(B, ids)
        x0 = self.x[0]
        if self.embeddings:
            ops.embag_fwd(self.V, self.toffs, self.T, ids, bag_len=1, out=x0, out_ld=self.dp, err=self.err)
        else:
            if pooled is None or tuple(pooled.shape) != (B, self.T * self.dim):
                raise ops.MiError("DCNv2 without embeddings: pooled must be [B][T*dim]")
            ops.pack_dense_bf16(pooled, x0, 0, self.T * self.dim)
        ops.pack_dense_bf16(dense if self.n_dense else None, x0, self.T * self.dim, self.dp)
        if self.n_cross:
            self.cross.forward(x0)          # x_{l+1} = x0 * (x_l W_l^T + b_l) + x_l, one launch per layer
        a = self.x[-1]
        for i in range(len(self.mlp)):
            ops.gemm_bf16(a, self.Wm[i].h, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=self.h[i],
                          bias=self.bm[i].w)
            a = self.h[i]
        K = self.mlp[-1]
        ops.head_fwd(a, self.out.w[:K], self.out.w[K:K + 1], labels, 1.0 / (B * self.loss_ranks), self.logit,
                     self.loss_rows if labels is not None else None, self.g if labels is not None else None)
        return self.logit

    # -- one training step --------------------------------------------------------------------
    def _apply(self, p, n_slabs, n, stride):
        if self.grads_only:         # the slabs added up into the gradient, nothing else
            ops.slab_adam(self.slab, n_slabs, n, stride, None, None, None, 0.0, 0.0, 0.0, grad_out=p.grad)
            return
        ops.slab_adam(self.slab, n_slabs, n, stride, p.w, p.m, p.v, self.beta1, self.beta2, self.eps,
                      lr_dev=self.sched, w_bf16=p.h, grad_out=p.grad if self.keep_grads else None)

    def apply_dense(self):
        """After a ``grads_only`` backward: TensorFlow-form Adam on every dense parameter from ``grad_arena`` (one
        launch), then the bf16 working copies of the matrices in both orientations."""
        ops.adam_tf_dense(self.w_arena, self.m_arena, self.v_arena, self.grad_arena, 0.0, self.beta1, self.beta2, self.eps,
                          lr_dev=self.sched)
        for p in self.Wc + self.Wm:
            ops.cast_bf16(p.w, p.h)
            ops.transpose_bf16(p.h, p.hT)

    def _weight_step(self, p, dY, X):
        """dW = dY^T X as split-K slabs, then the fused reduction + Adam + bf16 re-cast."""
        n_out, k_in = p.w.shape
        s = self.split[(n_out, k_in)]
        ops.gemm_bf16(dY, X, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=n_out, N=k_in, K=dY.shape[0], k_splits=s,
                      out=self.slab[:s * n_out * k_in].view(s, n_out, k_in))
        self._apply(p, s, n_out * k_in, n_out * k_in)
        if not self.grads_only:
            ops.transpose_bf16(p.h, p.hT)

    def _bias_step(self, p, dY):
        n = p.w.numel()
        part = self.slab[:self.vec_slices * n].view(self.vec_slices, n)
        ops.colsum_bf16(dY, self.vec_slices, part, N=n)
        self._apply(p, self.vec_slices, n, n)

    def train_step(self, ids, dense, labels, check_ids=False, pooled=None, graph=False):
        """Forward, backward and every optimizer update; returns the mean loss (1-element tensor).
        ``graph=True`` replays a HIP graph of the step captured on first use for these input buffers (they must stay
        at the same addresses; their contents may change): the step's ~70 launches are one chain on one stream, so a
        replay only drops the gaps between them - the same launches, the same bits."""
        if graph:
            B = ids.shape[0]
            key = (ids.data_ptr(), dense.data_ptr() if dense is not None else 0, labels.data_ptr(), B, ids.dtype,
                   pooled.data_ptr() if pooled is not None else 0)
            g = self._graphs.get(key)
            if g is None:
                self._buffers(B, ids)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step(ids, dense, labels, pooled)
                self._graphs[key] = g
            g.replay()
        else:
            self._step(ids, dense, labels, pooled)
        self.steps += 1
        if check_ids:
            ops.check_error_word(self.err, "DCN ids")
        return self.loss

    def _step(self, ids, dense, labels, pooled):
        B = ids.shape[0]
        self.forward(ids, dense, labels, pooled=pooled)
        ops.reduce_sum(self.loss_rows, 1.0 / B, out=self.loss)
        ops.adam_schedule_step(self.sched, self.lr, self.beta1, self.beta2)
        x0 = self.x[0]
        n_mlp = len(self.mlp)
        K = self.mlp[-1]
        # output layer: dz of the last hidden layer, and the gradient of w and the bias
        part = self.slab[:self.vec_slices * (K + 4)].view(self.vec_slices, K + 4)
        part.zero_()
        ops.head_bwd(self.g, self.out.w[:K], self.h[-1], self.dz[-1], self.vec_slices, part)
        self._apply(self.out, self.vec_slices, K + 4, K + 4)
        # MLP, last layer first.  Per layer: the data gradient with the OLD weights, then the weight step.
        for i in reversed(range(n_mlp)):
            dz = self.dz[i]
            inp = self.h[i - 1] if i > 0 else self.x[-1]
            if i > 0:
                ops.gemm_bf16(dz, self.Wm[i].hT, ops.MI_EPI_MASK, out=self.dz[i - 1], in0=self.h[i - 1])
            else:
                # into the cross network: dx_L and du_{L-1} = dx_L * x0
                L = self.n_cross
                if L:
                    ops.gemm_bf16(dz, self.Wm[0].hT, ops.MI_EPI_CROSS_BWD, out=self.dx[L], out1=self.du[L & 1],
                                  in0=x0)
                else:
                    ops.gemm_bf16(dz, self.Wm[0].hT, ops.MI_EPI_CROSS_BWD, out=self.dx[0])
            self._weight_step(self.Wm[i], dz, inp)
            self._bias_step(self.bm[i], dz)
        # cross layers, last first: dx_l = du_l W_l + dx_{l+1}, du_{l-1} = dx_l * x0, every layer's weight and bias
        # step, and d x0 = dx_0 + sum_l dx_{l+1} * u_l in one pass over the stored tensors
        if self.n_cross:
            self.cross.backward(x0, self.dx0, self.beta1, self.beta2, self.eps, lr_dev=self.sched, du_last_ready=True,
                                apply_adam=not self.grads_only)
        else:
            ops.cross_input_grad(self.dx, self.u, self.dx0)
        # embeddings: sort by row, sum each row's gradients in a fixed order, row-wise Adagrad
        if self.embeddings:
            self.plan.prepare(ids)
            ops.bwd_apply_adagrad_rowwise(self.plan, self.dx0, self.V, self.V_acc, self.emb_lr, self.emb_eps,
                                          grad_ld=self.dp)

    # -- the step's contractions, one by one (for per-kernel timing) ----------------------------
    def contraction_calls(self):
        """[(class, name, fn, flops, (A, B))] for every mi_gemm_bf16 launch of one training step on the buffers
        the last step left behind: class 'row' = both operands stored reduction-contiguous (forward layers
        and data gradients), 'wgrad' = both reduction-major (weight gradients, split-K slabs).  The calls
        write the same tensors the step writes; weights are not touched.  (A, B): the two bf16 operands - class 'row'
        computes A B^T (+ epilogue), class 'wgrad' A^T B - for timing another library's product on the same data."""
        B = self.x[0].shape[0]
        x0, L, n_mlp = self.x[0], self.n_cross, len(self.mlp)
        calls = []
        for l in range(L):
            calls.append(('row', 'cross%d fwd' % l, (lambda l=l: ops.gemm_bf16(
                self.x[l], self.Wc[l].h, ops.MI_EPI_CROSS_FWD, out=self.x[l + 1], out1=self.u[l], bias=self.bc[l].w, in0=x0,
                in1=self.x[l])), 2.0 * B * self.dp * self.dp, (self.x[l], self.Wc[l].h)))
        a = self.x[-1]
        for i in range(n_mlp):
            calls.append(('row', 'mlp%d fwd' % i, (lambda i=i, a=a: ops.gemm_bf16(
                a, self.Wm[i].h, ops.MI_EPI_BIAS_ACT, flags=ops.MI_EPI_RELU, out=self.h[i], bias=self.bm[i].w)),
                2.0 * B * self.Wm[i].w.shape[0] * self.Wm[i].w.shape[1], (a, self.Wm[i].h)))
            a = self.h[i]
        for i in reversed(range(n_mlp)):
            dz = self.dz[i]
            inp = self.h[i - 1] if i > 0 else self.x[-1]
            n_out, k_in = self.Wm[i].w.shape
            if i > 0:
                calls.append(('row', 'mlp%d dX' % i, (lambda i=i, dz=dz: ops.gemm_bf16(
                    dz, self.Wm[i].hT, ops.MI_EPI_MASK, out=self.dz[i - 1], in0=self.h[i - 1])), 2.0 * B * n_out * k_in,
                    (dz, self.Wm[i].hT)))
            else:
                calls.append(('row', 'mlp0 dX', (lambda dz=dz: ops.gemm_bf16(
                    dz, self.Wm[0].hT, ops.MI_EPI_CROSS_BWD, out=self.dx[L], out1=self.du[L & 1], in0=x0)),
                    2.0 * B * n_out * k_in, (dz, self.Wm[0].hT)))
            s = self.split[(n_out, k_in)]
            calls.append(('wgrad', 'mlp%d dW' % i, (lambda dz=dz, inp=inp, s=s, n_out=n_out, k_in=k_in: ops.gemm_bf16(
                dz, inp, ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=n_out, N=k_in, K=dz.shape[0], k_splits=s,
                out=self.slab[:s * n_out * k_in].view(s, n_out, k_in))), 2.0 * B * n_out * k_in, (dz, inp)))
        for l in reversed(range(L)):
            du = self.du[(l + 1) & 1]
            if l > 0:
                calls.append(('row', 'cross%d dX' % l, (lambda l=l, du=du: ops.gemm_bf16(
                    du, self.Wc[l].hT, ops.MI_EPI_CROSS_BWD, out=self.dx[l], in1=self.dx[l + 1], out1=self.du[l & 1],
                    in0=x0)), 2.0 * B * self.dp * self.dp, (du, self.Wc[l].hT)))
            else:
                calls.append(('row', 'cross0 dX', (lambda du=du: ops.gemm_bf16(
                    du, self.Wc[0].hT, ops.MI_EPI_CROSS_BWD, out=self.dx[0], in1=self.dx[1])), 2.0 * B * self.dp * self.dp,
                    (du, self.Wc[0].hT)))
            s = self.split[(self.dp, self.dp)]
            calls.append(('wgrad', 'cross%d dW' % l, (lambda l=l, du=du, s=s: ops.gemm_bf16(
                du, self.x[l], ops.MI_EPI_SLAB, a_rm=True, b_rm=True, M=self.dp, N=self.dp, K=du.shape[0], k_splits=s,
                out=self.slab[:s * self.dp * self.dp].view(s, self.dp, self.dp))), 2.0 * B * self.dp * self.dp, (du, self.x[l])))
        return calls

    # -- accounting ---------------------------------------------------------------------------
    def flops_per_sample(self):
        """Multiply-adds x2 of the contractions of one training step (forward + both backward products)."""
        fwd = self.n_cross * self.dp * self.dp
        w = [self.dp] + self.mlp
        fwd += sum(w[i] * w[i + 1] for i in range(len(self.mlp)))
        return 3 * 2 * fwd
