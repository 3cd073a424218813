"""Train step of the colour model: the arithmetic of the reference's hot loop
(``recommendation/code/model_trainer.py:138-158,170-174`` and the identical single-process
``recommendation/local_train_utils.py:33-63``) without the parameter-server cluster around it.

The reference distributes with TensorFlow's asynchronous parameter server (2 workers + 1 ps,
model_trainer.py:124-136): every variable read and Adam apply is a gRPC hop and workers'
updates interleave unsynchronised.  Here N ranks run synchronous data parallelism instead:
each rank takes its own file shard (files[index::parallelism], as the reference does), and the
flat gradient buffer is all-reduced (mean) before the single fused Adam launch, so all
replicas hold identical parameters after every step.
"""
import torch

from . import ops
from .r_model import COLUMNS, RecommendationModel, Sample


class AdamOptimizer(object):
    """``tf.train.AdamOptimizer``: alpha_t = lr*sqrt(1-b2^t)/(1-b1^t); m, v dense on every
    variable; theta -= m*alpha_t/(sqrt(v)+eps) - one launch over the flat parameter buffer."""

    def __init__(self, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.learning_rate, self.beta1, self.beta2, self.epsilon = learning_rate, beta1, beta2, epsilon
        self.m = self.v = None
        self._schedule = ops.TFAdamSchedule(learning_rate, beta1, beta2)

    @property
    def t(self):
        return self._schedule.t

    def _slots_for(self, params):
        if self.m is None or self.m.numel() != params.numel():
            self.m = torch.zeros_like(params, requires_grad=False)
            self.v = torch.zeros_like(params, requires_grad=False)

    def apply_gradients(self, params, grad):
        self._slots_for(params)
        ops.adam_tf_dense(params.detach(), self.m, self.v, grad.contiguous(), self._schedule.step(), self.beta1,
                          self.beta2, self.epsilon)

    def minimize(self, loss, model, global_step=None, process_group=None):
        """Backward through the model, (optionally) average gradients over the ranks, update.
        Returns the new global step."""
        params = model.parameters()
        model.zero_grad()
        loss.backward()
        model.backward_done()
        grad = model.grad_arena()
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            world = torch.distributed.get_world_size(process_group)
            if world > 1:
                torch.distributed.all_reduce(grad, group=process_group)
                grad = grad / world
        self.apply_gradients(params, grad)
        return self.t if global_step is None else global_step + 1


def columns_to_features(columns):
    """The dict the reference builds from the dataset columns (model_trainer.py:142-146)."""
    features = {k: columns[i] for i, k in enumerate(COLUMNS[:6])}
    labels = columns[6] if len(columns) > 6 else None
    return features, labels


def train_step(model, optimizer, columns, process_group=None):
    """One ``mon_sess.run([train_op, acc, labels, top_indices_real, loss, global_step])``
    (model_trainer.py:147-158, 173-174).  Returns (acc, labels, top_indices, loss, global_step)."""
    m = model
    features, labels = columns_to_features(columns)
    label_tensor = m.input_to_one_hot_plus(labels, m.colour_count)
    fs = m.features(features)
    last_layer = m.forward(fs)
    top_indices, top_values = m.output(last_layer)
    top_indices_real = top_indices.to(torch.int32)
    acc = m.accuracy(labels, top_indices_real, m.recommend_num)
    loss = m.loss(logits=last_layer, labels=label_tensor)
    global_step = optimizer.minimize(loss, m, process_group=process_group)
    return acc, labels, top_indices_real, loss.detach(), global_step


class GraphedTrainStep(object):
    """``train_step`` for a fixed batch size captured ONCE as a HIP graph and replayed: the colour model's
    step is ~70 launches of a 5,910-parameter model (encoder checks, 11 layers forward and backward through
    autograd, heads, Adam), each a few microseconds of device work behind tens of microseconds of Python and
    launch overhead, so the eager step is bound by the host.  Replaying the captured launches removes that;
    the arithmetic is the same kernels in the same order, so results equal the eager step's bit for bit.

    What a captured step cannot do is read anything back or take a scalar from the host, so
    * Adam's step size lives on the device (mi_adam_schedule_step advances {alpha_t, beta1_power,
      beta2_power} inside the graph; the host copy of the schedule is advanced beside it for checkpoints),
    * the id error word is read every ``check_every`` replays (and by ``check()``) instead of in every
      ``features`` call: a bad id is never dereferenced meanwhile, it is just reported a few steps later.
    A batch of another shape (the ragged last batch of an epoch, recommend lists of another width) runs the eager step.  The tensors returned by
    a replay are the graph's own output buffers: the next call overwrites them.

    Data parallel (``process_group`` of more than one rank): the step is captured as TWO graphs around the
    all-reduce - {encoders, forward, heads, loss, backward} and {schedule, Adam} - and a call is replay,
    ``all_reduce`` of the flat gradient, division by the world size, replay: the same kernels and the same
    collective as ``AdamOptimizer.minimize(process_group=...)``, so the same bits.  The collective itself stays
    outside the graphs: torch.distributed's work objects are not capturable from Python, and this build has no
    second GPU to exercise a captured RCCL call between two ranks."""

    def __init__(self, model, optimizer, columns, check_every=64, process_group=None):
        up = torch.distributed.is_available() and torch.distributed.is_initialized()
        self.group = process_group
        self.world = torch.distributed.get_world_size(process_group) if (up or process_group is not None) else 1
        m = self.model = model
        self.opt = optimizer
        self.check_every = int(check_every)
        self.calls = 0
        self.static_in = tuple(m._ids(c).clone() for c in columns)
        self.batch = int(self.static_in[0].shape[0])
        features, labels = columns_to_features(self.static_in)
        with torch.no_grad():                       # variables, table offsets and the error word exist before capture
            m.input_to_one_hot_plus(labels, m.colour_count)
            m.forward(m.features(features))
        params = m.parameters()
        optimizer._slots_for(params)
        self.sched = torch.zeros(3, dtype=torch.float32, device=m.device)
        self._sched_to_device()
        saved = [t.detach().clone() for t in (params, optimizer.m, optimizer.v, self.sched)]
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream()
        side.wait_stream(cur)
        m._in_graph = True
        try:
            with torch.cuda.stream(side):
                for _ in range(3):                  # warm-up off the capture, then put everything back
                    self._body()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():
                for dst, src in zip((params, optimizer.m, optimizer.v, self.sched), saved):
                    dst.copy_(src)
            self.graph = torch.cuda.CUDAGraph()
            if self.world == 1:
                with torch.cuda.graph(self.graph):
                    self.out = self._body()
            else:
                self.gbuf = torch.empty_like(m.grad_arena())     # the averaged gradient the second graph reads
                with torch.cuda.graph(self.graph):
                    self.out = self._backward_part()
                self.graph_apply = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_apply):
                    self._apply_part(self.gbuf)
        finally:
            m._in_graph = False

    def _sched_to_device(self):
        sch = self.opt._schedule
        self.sched.copy_(torch.tensor([0.0, float(sch.b1p), float(sch.b2p)], dtype=torch.float32))

    def _body(self):
        out = self._backward_part()
        grad = self.model.grad_arena()
        if self.world > 1:      # (warm-up of the data-parallel form: the ranks' gradients are NOT exchanged here)
            grad = grad / self.world
        self._apply_part(grad)
        return out

    def _apply_part(self, grad):
        m, opt = self.model, self.opt
        ops.adam_schedule_step(self.sched, opt.learning_rate, opt.beta1, opt.beta2)
        ops.adam_tf_dense(m.parameters().detach(), opt.m, opt.v, grad, 0.0, opt.beta1, opt.beta2, opt.epsilon,
                          lr_dev=self.sched)

    def _backward_part(self):
        m = self.model
        features, labels = columns_to_features(self.static_in)
        label_tensor = m.input_to_one_hot_plus(labels, m.colour_count)
        fs = m.features(features)
        last_layer = m.forward(fs)
        top_indices, _ = m.output(last_layer)
        top_indices_real = top_indices.to(torch.int32)
        acc = m.accuracy(labels, top_indices_real, m.recommend_num)
        loss = m.loss(logits=last_layer, labels=label_tensor)
        m.zero_grad()
        loss.backward()
        m.backward_done()
        return acc, top_indices_real, loss.detach()

    def check(self):
        self.model.check_errors()

    def __call__(self, columns):
        """Same result tuple as train_step: (acc, labels, top_indices, loss, global_step)."""
        m = self.model
        ins = [m._ids(c) for c in columns]
        # any column of another shape than the captured one - a ragged last batch, or a batch whose recommend
        # lists have another width - takes the eager step; the device's schedule is brought back in line after it
        if len(ins) != len(self.static_in) or any(tuple(a.shape) != tuple(b.shape) for a, b in zip(ins, self.static_in)):
            res = train_step(m, self.opt, columns, process_group=self.group)
            self._sched_to_device()
            return res
        for dst, src in zip(self.static_in, ins):
            dst.copy_(src, non_blocking=True)
        m._in_graph = True
        try:
            self.graph.replay()
            if self.world > 1:
                grad = m.grad_arena()
                torch.distributed.all_reduce(grad, group=self.group)
                torch.div(grad, self.world, out=self.gbuf)
                self.graph_apply.replay()
        finally:
            m._in_graph = False
        self.opt._schedule.step()                   # the host's copy of the schedule (checkpoints, global_step)
        self.calls += 1
        if self.check_every and self.calls % self.check_every == 0:
            m.check_errors()
        acc, top, loss = self.out
        return acc, self.static_in[6], top, loss, self.opt.t


def train(files, batch_size=300, last_step=5000, learning_rate=1e-4, device='cuda:0', seed=None, parallelism=None,
          index=None, log_every=100, model=None, process_group=None, graph=False, checkpoint_dir=None,
          save_checkpoint_steps=None, checkpoint_format='npz', resume_input=False, is_chief=None):
    """The loop of local_train_utils.train (:30-66) / ModelTrainer.train's worker branch
    (model_trainer.py:138-179): same model constants, batch size, optimizer and stop step.
    ``graph=True`` replays the step as a captured HIP graph (single process; see GraphedTrainStep).

    ``checkpoint_dir`` is MonitoredTrainingSession's (local_train_utils.py:54-58, model_trainer.py:160-169): when
    the directory holds a checkpoint, variables, Adam slots, schedule and global step are restored on entry and
    the loop runs on until the GLOBAL step reaches ``last_step`` (StopAtStepHook(last_step=...)); a checkpoint is
    written every ``save_checkpoint_steps`` steps (the reference saves by wall-clock time - every 600 s by default,
    30 / 60 s in its batch / stream jobs, model_trainer.py:230-233,196-199 - which has no reproducible counterpart)
    and when the loop ends, by the chief only (rank 0 unless ``is_chief`` says otherwise; model_trainer.py:164).
    As in the reference the input starts over after a restore; ``resume_input=True`` instead skips the batches the
    restored steps consumed, which makes an interrupted run continue bit for bit like an uninterrupted one."""
    from . import checkpoint
    m = model or RecommendationModel(colour_count=128, recommend_num=6, user_count=100, country_count=20,
                                     device=device, seed=seed)
    dataset = Sample.read_label_data(files, batch_size=batch_size, parallelism=parallelism, index=index)
    iterator = dataset.make_one_shot_iterator()
    optimizer = AdamOptimizer(learning_rate=learning_rate)
    if is_chief is None:
        is_chief = not index
    step = 0
    if checkpoint_dir is not None and checkpoint.latest(checkpoint_dir) is not None:
        step = checkpoint.load(checkpoint_dir, m, optimizer)
        if resume_input:
            for _ in range(step):
                iterator.get_next()
    history = []
    graphed = None
    saved_at = step
    while step < last_step:
        columns = iterator.get_next()
        step += 1
        if graph:
            if graphed is None:
                graphed = GraphedTrainStep(m, optimizer, columns, process_group=process_group)
            acc_res, _, _, loss_res, global_step_res = graphed(columns)
        else:
            acc_res, _, _, loss_res, global_step_res = train_step(m, optimizer, columns, process_group)
        if log_every and step % log_every == 0:
            history.append((step, float(loss_res), float(acc_res)))
            print("Train step %d, loss: %f accuracy: %f" % (step, float(loss_res), float(acc_res)))
        if checkpoint_dir is not None and is_chief and save_checkpoint_steps and step % save_checkpoint_steps == 0:
            if graphed is not None:
                graphed.check()                     # never save parameters a bad id may have spoilt unnoticed
            checkpoint.save(checkpoint_dir, m, optimizer, step=step, fmt=checkpoint_format)
            saved_at = step
    if graphed is not None:
        graphed.check()
    if checkpoint_dir is not None and is_chief and saved_at != step:
        checkpoint.save(checkpoint_dir, m, optimizer, step=step, fmt=checkpoint_format)
    return m, optimizer, history
