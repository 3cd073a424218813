"""Checkpoints under the reference's variable names.

The reference saves through tf.train.MonitoredTrainingSession(checkpoint_dir=...)
(recommendation/code/model_trainer.py:165): a TensorFlow bundle holding
  dense/kernel, dense/bias, dense_1/kernel, ... dense_10/bias          (tf.layers.dense, r_model.py:52-58)
  <variable>/Adam, <variable>/Adam_1                                   (AdamOptimizer's m and v slots)
  beta1_power, beta2_power                                             (its fp32 schedule variables)
  global_step
No checkpoint ships with the reference and TensorFlow is not available to write one, so the
container here is a plain ``.npz`` whose KEYS are exactly those names ('/' kept): converting a
bundle to it is one tf.train.load_checkpoint loop on a machine that has TensorFlow.  The first-layer
kernels keep the reference's shapes ([256, 8] for the concatenated recommend/click towers), so
nothing has to be split or merged.
"""
import os

import numpy as np
import torch


def variables_of(model, optimizer=None):
    """name -> numpy array for everything a reference checkpoint would hold."""
    out = {}
    for name, v in model.variables.items():
        out[name] = v.cpu().numpy().copy()
    if optimizer is not None and optimizer.m is not None:
        for name, (off, shape) in model._slots.items():
            n = int(np.prod(shape))
            out[name + '/Adam'] = optimizer.m[off:off + n].view(shape).cpu().numpy().copy()
            out[name + '/Adam_1'] = optimizer.v[off:off + n].view(shape).cpu().numpy().copy()
        out['beta1_power'] = np.float32(optimizer._schedule.b1p)
        out['beta2_power'] = np.float32(optimizer._schedule.b2p)
    out['global_step'] = np.int64(optimizer.t if optimizer is not None else 0)
    return out


def save(checkpoint_dir, model, optimizer=None, step=None):
    """Write ``model.ckpt-<step>.npz`` under checkpoint_dir and point ``checkpoint`` at it (the file
    TensorFlow keeps there, model_trainer.py:55-63 rewrites the same file)."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    values = variables_of(model, optimizer)
    step = int(values['global_step']) if step is None else int(step)
    name = 'model.ckpt-%d' % step
    np.savez(os.path.join(checkpoint_dir, name + '.npz'), **values)
    with open(os.path.join(checkpoint_dir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))
    return os.path.join(checkpoint_dir, name + '.npz')


def latest(checkpoint_dir):
    """Path of the newest checkpoint named by the ``checkpoint`` file, or None."""
    index = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.exists(index):
        return None
    for line in open(index):
        if line.startswith('model_checkpoint_path:'):
            name = line.split(':', 1)[1].strip().strip('"')
            path = os.path.join(checkpoint_dir, os.path.basename(name) + '.npz')
            return path if os.path.exists(path) else None
    return None


def load(path_or_dir, model, optimizer=None):
    """Restore variables (and, when given an optimizer, its slots and schedule).  The model must
    have been wired (one forward) or takes its variable shapes from the file.  Returns global_step."""
    path = latest(path_or_dir) if os.path.isdir(path_or_dir) else path_or_dir
    if path is None:
        raise FileNotFoundError("no checkpoint under %s" % path_or_dir)
    with np.load(path, allow_pickle=False) as z:
        values = {k: z[k] for k in z.files}
    slots = ('/Adam', '/Adam_1')
    model.load_variables({k: v for k, v in values.items()
                          if not k.endswith(slots) and k not in ('beta1_power', 'beta2_power', 'global_step')})
    step = int(values.get('global_step', 0))
    if optimizer is not None:
        optimizer._slots_for(model.parameters())
        for name, (off, shape) in model._slots.items():
            n = int(np.prod(shape))
            for suffix, flat in (('/Adam', optimizer.m), ('/Adam_1', optimizer.v)):
                if name + suffix in values:
                    flat[off:off + n] = torch.from_numpy(values[name + suffix].reshape(-1)).to(flat.device)
        sch = optimizer._schedule
        sch.t = step
        if 'beta1_power' in values:
            sch.b1p, sch.b2p = np.float32(values['beta1_power']), np.float32(values['beta2_power'])
        else:
            sch.b1p, sch.b2p = np.float32(sch.beta1) ** step, np.float32(sch.beta2) ** step
    return step
