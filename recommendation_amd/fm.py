"""Factorisation machine on Criteo-shaped input: host side of the FM hot path.

The reference has no FM (recommendation/code/r_model.py:24-140 is a linear network); this
is the extension the benchmark definition names.  It reuses the colour model's building
blocks - row gather, sort-by-row sparse backward, TensorFlow-form Adam
(recommendation/code/model_trainer.py:157) - on 26 large tables.

One training step is five pieces of the C ABI and no torch arithmetic:
  mi_adam_schedule_step the optimizer's scalar state (alpha_t, beta powers), kept on the device
  mi_fm_fwd            gather + pairwise term in registers + BCE loss rows + dL/dlogit
  mi_embag_bwd_prepare stable sort of (row, position)
  mi_fm_bwd_apply      per distinct row: sum_i g_i*(S_i - v) and the fused optimizer
  mi_fm_dense_step     dense-feature weight / bias gradients, their Adam update, mean loss
Up to 8192 samples per batch the first three share ONE launch (mi_fm_fwd_sort: they need only the
step's inputs) and the last two another (mi_fm_bwd_apply_dense: both start from dL/dlogit), so a
step is two launches, each as long as its longest member; results are bit-identical either way.
Parameters and optimizer state live in one record per row (ops.FmRecords), so a row's update
touches one or two 128-byte lines.  All buffers are allocated once per batch shape: a step
allocates nothing and can be captured in a HIP graph.
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
        self.table = ops.FmRecords(self.total_rows, self.dim, _OPT[opt], self.device)
        # initialise V and w1 in place, chunked so the temporary stays small next to a 8 GB table
        chunk = 1 << 22
        for lo in range(0, self.total_rows, chunk):
            hi = min(lo + chunk, self.total_rows)
            self.table.rec[lo:hi, :self.dim + 1] = torch.randn((hi - lo, self.dim + 1), generator=gen,
                                                               **f32).mul_(init_scale)
        # dense side: wd[n_dense] followed by the bias, one flat buffer so one launch covers both
        self.dp = torch.zeros((self.n_dense + 1,), **f32)
        self.dp[:self.n_dense] = torch.randn((self.n_dense,), generator=gen, **f32).mul_(init_scale)
        self.dm, self.dv = torch.zeros_like(self.dp), torch.zeros_like(self.dp)
        # tf.train.AdamOptimizer's {alpha_t, beta1_power, beta2_power}, kept on the device so that a
        # captured graph of the step can be replayed unchanged step after step
        self.sched = torch.tensor([0.0, 1.0, 1.0, 0.0], **f32)
        self.steps = 0
        self.err = ops.new_error_word(self.device)
        self._buf_B = None
        self._bufsets = {}                     # (B, id dtype) -> (buffers, loss, plan); kept, so captured graphs stay valid
        self._graphs = {}
        self._side = None
        self._prefetched = None                # (address, B, dtype) of the ids whose sort the last step left behind
        self._cur = 0                          # which of the two plans of a buffer set holds it
        self._multi_memo = None                # the last train_steps call: (lists, length, end addresses, graph)
        self.fused = True                      # set False to run the step as five separate launches
        self._fused_max = ops.fm_fwd_sort_max_batch()
        # beyond this many samples a table is sorted by several workgroups that each walk all its ids: only with the
        # ids also given field-major (``ids_tm``)
        self._fused_any_layout_max = ops.fm_lds_sort_single_max()

    # -- views ------------------------------------------------------------------------------
    @property
    def V(self):
        return self.table.V

    @property
    def w1(self):
        return self.table.w1

    @property
    def wd(self):
        return self.dp[:self.n_dense]

    @property
    def bias(self):
        return self.dp[self.n_dense:]

    def load_state(self, st):
        """Copy parameters from an object with numpy V, w1, wd, bias (the test oracle's state)."""
        self.table.rec[:, :self.dim] = torch.from_numpy(st.V).to(self.device)
        self.table.rec[:, self.table.off_w1] = torch.from_numpy(st.w1).to(self.device)
        self.dp[:self.n_dense].copy_(torch.from_numpy(st.wd))
        self.dp[self.n_dense] = float(st.bias)

    def state_bytes(self):
        return self.table.rec.numel() * 4

    # -- buffers ----------------------------------------------------------------------------
    def _buffers(self, ids):
        B = ids.shape[0]
        key = (B, ids.dtype)
        if self._buf_B != key:
            # One buffer set per batch shape, never freed while the model lives: a HIP graph captured
            # for one shape keeps pointing at that shape's buffers after steps of another shape (a
            # ragged last batch) ran in between.
            if key not in self._bufsets:
                f32 = dict(dtype=torch.float32, device=self.device)
                buf = dict(logit=torch.empty((B,), **f32), S=torch.empty((B, self.dim), **f32),
                           loss_rows=torch.empty((B,), **f32), g=torch.empty((B,), **f32))
                loss = torch.empty((1,), **f32)
                plan = ops.SortedPlan(self.toffs, self.T, self.total_rows, ids, bag_len=1, runs=False)
                # a second plan, so that the NEXT batch's ids can be sorted while this batch's plan is in use
                plan2 = ops.SortedPlan(self.toffs, self.T, self.total_rows, ids, bag_len=1, runs=False) \
                    if B <= self._fused_max else None
                self._bufsets[key] = (buf, loss, [plan, plan2])
            self.buf, self.loss, self._plans = self._bufsets[key]
            self.plan = self._plans[0]
            self._buf_B = key
        return self.buf

    # -- steps ------------------------------------------------------------------------------
    def forward(self, ids, dense, labels=None):
        """logits [B]; with labels also fills loss rows and dL/dlogit of the mean BCE loss."""
        buf = self._buffers(ids)
        B = ids.shape[0]
        return ops.fm_fwd(self.table, self.toffs, ids, dense if self.n_dense else None,
                          self.wd if self.n_dense else None, self.bias, labels, 1.0 / B, self.err, out=buf)

    def _step_prefetched(self, ids, dense, labels, cur, next_ids, next_ids_tm, sorted_now, ids_tm):
        """The step with the sort off its critical path: the sort needs nothing but ids, so the NEXT
        batch's ids are sorted on a side stream while this batch's forward and update run; the step
        itself is forward (+ schedule) and update (+ dense side).  ``cur``: which of the two plans
        holds this batch's sort; ``sorted_now``: it does not yet (first step, or the caller passed
        other ids than it announced), so this batch is sorted here first."""
        main = torch.cuda.current_stream()
        adam = self.opt == 'adam_tf'
        B = ids.shape[0]
        plan, nxt = self._plans[cur], self._plans[cur ^ 1]
        if sorted_now:
            ops.fm_sort_ids(self.toffs, ids, plan, ids_tm=ids_tm)
        side = self._side[0]
        side.wait_stream(main)
        with torch.cuda.stream(side):
            ops.fm_sort_ids(self.toffs, next_ids, nxt, ids_tm=next_ids_tm)
        r = ops.fm_fwd_sort(self.table, self.toffs, ids, None, dense if self.n_dense else None,
                            self.wd if self.n_dense else None, self.bias, labels, 1.0 / B, self.err,
                            out=self.buf, sched=self.sched, lr=self.lr, beta1=self.beta1, beta2=self.beta2, ids_tm=ids_tm)
        ops.fm_bwd_apply_dense(plan, self.table, r['g'], r['S'], _OPT[self.opt], self.lr, self.beta1, self.beta2,
                               self.eps, dense if self.n_dense else None, r['loss_rows'], self.dp, self.dm, self.dv,
                               self.loss, lr_dev=self.sched if adam else None, dense_lr_dev=self.sched)
        main.wait_stream(side)
        self.plan = plan

    def _fused_ok(self, B, ids_tm):
        return self.fused and (B <= self._fused_any_layout_max or (ids_tm is not None and B <= self._fused_max))

    def _step_body(self, ids, dense, labels, streams=None, ids_tm=None, stage=None):
        """The launches of one step.  With ``streams`` (two side streams, used under graph capture)
        the sort runs beside the forward - it only needs the ids - and the dense-side update beside
        the sparse one."""
        main = torch.cuda.current_stream()
        adam = self.opt == 'adam_tf'
        B = ids.shape[0]
        if self._fused_ok(B, ids_tm):
            # two launches: {schedule, forward, sort} and {sparse update, dense side}
            r = ops.fm_fwd_sort(self.table, self.toffs, ids, self.plan, dense if self.n_dense else None,
                                self.wd if self.n_dense else None, self.bias, labels, 1.0 / B, self.err,
                                out=self._buffers(ids), sched=self.sched, lr=self.lr, beta1=self.beta1,
                                beta2=self.beta2, ids_tm=ids_tm)
            ops.fm_bwd_apply_dense(self.plan, self.table, r['g'], r['S'], _OPT[self.opt], self.lr, self.beta1,
                                   self.beta2, self.eps, dense if self.n_dense else None, r['loss_rows'], self.dp,
                                   self.dm, self.dv, self.loss, lr_dev=self.sched if adam else None,
                                   dense_lr_dev=self.sched, stage=stage)
            return
        if stage is not None:
            raise ops.MiError("train_step: stage needs the two-launch step (fused, this batch size and id layout)")
        if streams:
            streams[0].wait_stream(main)
            with torch.cuda.stream(streams[0]):
                self.plan.prepare(ids)
        ops.adam_schedule_step(self.sched, self.lr, self.beta1, self.beta2)
        r = self.forward(ids, dense, labels)
        if not streams:
            self.plan.prepare(ids)

        def dense_side():
            ops.fm_dense_step(r['g'], dense if self.n_dense else None, r['loss_rows'], self.dp, self.dm, self.dv, 0.0,
                              self.beta1, self.beta2, self.eps, self.loss, lr_dev=self.sched)

        if streams:
            streams[1].wait_stream(main)
            with torch.cuda.stream(streams[1]):
                dense_side()
            main.wait_stream(streams[0])
        ops.fm_bwd_apply(self.plan, self.table, r['g'], r['S'], _OPT[self.opt], self.lr, self.beta1, self.beta2,
                         self.eps, lr_dev=self.sched if adam else None)
        if streams:
            main.wait_stream(streams[1])
        else:
            dense_side()

    def train_step(self, ids, dense, labels, check_ids=False, graph=False, ids_tm=None, next_ids=None,
                   next_ids_tm=None, stage=None):
        """One pass of the hot path; returns the mean loss as a 1-element device tensor.
        ``graph=True`` replays a HIP graph of the step captured on first use for these input
        buffers (the inputs must stay at the same addresses; their contents may change).
        ``ids_tm``: the same ids as a contiguous [T][B] tensor (one row per field, the way the
        reference feeds one tensor per feature); the two-launch step then reads that layout, which
        lets a table's sort load its ids as whole lines.
        ``next_ids`` (+ ``next_ids_tm``): the ids of the batch the NEXT call will train on, same
        shape.  The sort by row needs nothing but ids, so it is taken off this step's critical path:
        the next batch is sorted on a side stream beside this step's forward and update, and the next
        call (which must pass that same tensor, contents unchanged) finds its sort done.  Results are
        bit-identical to the step without ``next_ids``.
        ``stage`` = (device block, pinned host block): the step's second launch also copies the host block - the
        NEXT batch as the caller lays it out - into the device block, which must not be one this step reads (two
        blocks, taken in turn); eager steps only (capture them yourself for a graph), same bits."""
        self._buffers(ids)
        B = ids.shape[0]
        if stage is not None and (graph or next_ids is not None):
            raise ops.MiError("train_step: stage goes with the plain eager step")
        pre = next_ids is not None and self._fused_ok(B, ids_tm) and self._fused_ok(B, next_ids_tm)
        if pre:
            if tuple(next_ids.shape) != tuple(ids.shape) or next_ids.dtype != ids.dtype:
                raise ops.MiError("train_step: next_ids must have the shape and dtype of ids")
            if self._side is None:
                self._side = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
            want = (ids.data_ptr(), B, ids.dtype)
            sorted_now = self._prefetched != want
            cur = self._cur if not sorted_now else 0
            self._prefetched = (next_ids.data_ptr(), B, ids.dtype)
            self._cur = cur ^ 1
        else:
            self._prefetched = None
        if graph:
            key = (ids.data_ptr(), dense.data_ptr(), labels.data_ptr(), ids.shape[0], ids.dtype,
                   ids_tm.data_ptr() if ids_tm is not None else 0)
            if pre:
                key += (next_ids.data_ptr(), next_ids_tm.data_ptr() if next_ids_tm is not None else 0, cur, sorted_now)
            g = self._graphs.get(key)
            if g is None:
                if self._side is None:
                    self._side = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    if pre:
                        self._step_prefetched(ids, dense, labels, cur, next_ids, next_ids_tm, sorted_now, ids_tm)
                    else:
                        self._step_body(ids, dense, labels, streams=self._side, ids_tm=ids_tm)
                self._graphs[key] = g
            g.replay()
            if pre:
                self.plan = self._plans[cur]
        elif pre:
            self._step_prefetched(ids, dense, labels, cur, next_ids, next_ids_tm, sorted_now, ids_tm)
        else:
            self._step_body(ids, dense, labels, ids_tm=ids_tm, stage=stage)
        self.steps += 1
        if check_ids:
            ops.check_error_word(self.err, "FM ids")
        return self.loss

    def train_steps(self, batches, ids_tms=None, check_ids=False):
        """Several consecutive steps - one per (ids, dense, labels) of ``batches``, in order - replayed as ONE HIP
        graph captured on first use for these input buffers (which must stay at the same addresses; their contents
        may change).  Between two replays the device idles for the 4-5 us a graph launch costs; inside a replay a
        launch follows its predecessor at once, so a training loop that has its next few batches at hand pays
        that gap once per replay instead of once per step.  Same launches in the same order as calling
        ``train_step`` on each batch: the same bits.  Returns the steps' mean losses as one tensor.
        Calling again with the SAME list objects (same length, same first and last id tensors) replays the graph
        found for them last time without looking at the batches in between: build a new list when one of its
        batches is replaced by another tensor."""
        K = len(batches)
        # The host's share of a replay is kept short (the device idles while it runs): a list object that was passed
        # before, with the same length and the same first and last id tensors, is taken to hold the same buffers
        # (pass a NEW list after replacing a batch in the middle of one).
        memo = self._multi_memo
        if memo is not None and memo[0] is batches and memo[1] is ids_tms and memo[2] == K and \
                memo[3] == (batches[0][0].data_ptr(), batches[K - 1][0].data_ptr()):
            ent = memo[4]
        else:
            shapes = set((tuple(b[0].shape), b[0].dtype) for b in batches)
            if len(shapes) != 1:
                raise ops.MiError("train_steps: the batches of one replay must share a shape")
            key = ('multi',) + tuple((b[0].data_ptr(), b[1].data_ptr(), b[2].data_ptr()) for b in batches) + \
                  tuple(t.data_ptr() if t is not None else 0 for t in (ids_tms or [None] * K))
            ent = self._graphs.get(key)
            if ent is None:
                self._buffers(batches[0][0])
                if self._side is None:
                    self._side = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
                losses = torch.zeros((K,), dtype=torch.float32, device=self.device)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                keep = self.loss
                with torch.cuda.graph(g):
                    for j, (ids, dense, labels) in enumerate(batches):
                        self.loss = losses[j:j + 1]
                        self._step_body(ids, dense, labels, streams=self._side, ids_tm=ids_tms[j] if ids_tms else None)
                    keep.copy_(losses[K - 1:K])   # the model's own loss word, inside the graph: no launch behind the replay
                self.loss = keep
                ent = self._graphs[key] = (g, losses, keep)
            self._multi_memo = (batches, ids_tms, K, (batches[0][0].data_ptr(), batches[K - 1][0].data_ptr()), ent)
        self._buffers(batches[0][0])
        self._prefetched = None
        g, losses, keep = ent
        g.replay()
        if self.loss is not keep:   # (another shape's buffer set was current when this graph was captured)
            self.loss.copy_(losses[K - 1:K])
        self.steps += K
        if check_ids:
            ops.check_error_word(self.err, "FM ids")
        return losses
