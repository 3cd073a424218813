"""Checkpoints under the reference's variable names.

The reference saves through tf.train.MonitoredTrainingSession(checkpoint_dir=...)
(recommendation/code/model_trainer.py:165): a TensorFlow bundle holding
  dense/kernel, dense/bias, dense_1/kernel, ... dense_10/bias          (tf.layers.dense, r_model.py:52-58)
  <variable>/Adam, <variable>/Adam_1                                   (AdamOptimizer's m and v slots)
  beta1_power, beta2_power                                             (its fp32 schedule variables)
  global_step
Two containers, both keyed by exactly those names ('/' kept):
* ``fmt='bundle'``: TensorFlow's own on-disk form, ``model.ckpt-<step>.index`` +
  ``model.ckpt-<step>.data-00000-of-00001`` (tf_bundle.py restates the format), which is what a
  reference-trained checkpoint directory holds and what ``load`` / ``latest`` look for first;
* ``fmt='npz'`` (the default for ``save``): a plain ``.npz``.
No checkpoint ships with the reference and TensorFlow is not available here, so the bundle path has been
exercised only on files this module wrote (checksums, known answers and round trips pin the format
code).  The first-layer kernels keep the reference's shapes ([256, 8] for the concatenated recommend /
click towers); ``split_first_layer`` / ``merge_first_layer`` convert between that and the two embedding
tables R and K the gather formulation speaks of (r_model.py:31-40).
"""
import os

import numpy as np
import torch

from . import tf_bundle


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


def split_first_layer(kernel, colour_count):
    """``dense_2/kernel`` or ``dense_5/kernel`` [2*colour_count, 8] -> (R, K): the rows the n-hot recommend
    input selects and the rows the one-hot click input selects (concat order of r_model.py:36-38)."""
    if kernel.shape[0] != 2 * colour_count:
        raise ValueError("kernel has %d rows, expected %d" % (kernel.shape[0], 2 * colour_count))
    return kernel[:colour_count], kernel[colour_count:]


def merge_first_layer(R, K):
    return np.concatenate([R, K], axis=0)


def save(checkpoint_dir, model, optimizer=None, step=None, fmt='npz'):
    """Write ``model.ckpt-<step>`` under checkpoint_dir - as ``.npz`` or, with ``fmt='bundle'``, as a
    TensorFlow bundle (.index + .data-00000-of-00001) - and point ``checkpoint`` at it (the file
    TensorFlow keeps there, model_trainer.py:55-63 rewrites the same file)."""
    os.makedirs(checkpoint_dir, exist_ok=True)
    values = variables_of(model, optimizer)
    step = int(values['global_step']) if step is None else int(step)
    values['global_step'] = np.int64(step)           # the file's name and its global_step agree
    name = 'model.ckpt-%d' % step
    if fmt not in ('npz', 'bundle'):
        raise ValueError("fmt must be 'npz' or 'bundle'")
    # one container per step: ``latest`` prefers the bundle of a name, so a bundle left from an earlier save of
    # this step would shadow the .npz written now (and the other way round a stale .npz would linger)
    stale = [name + '.npz'] if fmt == 'bundle' else [name + '.index', name + '.data-00000-of-00001']
    for f in stale:
        if os.path.exists(os.path.join(checkpoint_dir, f)):
            os.remove(os.path.join(checkpoint_dir, f))
    if fmt == 'bundle':
        path = tf_bundle.write_bundle(os.path.join(checkpoint_dir, name), {k: np.asarray(v) for k, v in values.items()})
    else:
        path = os.path.join(checkpoint_dir, name + '.npz')
        np.savez(path, **values)
    with open(os.path.join(checkpoint_dir, 'checkpoint'), 'w') as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (name, name))
    return path


def latest(checkpoint_dir):
    """Path of the newest checkpoint named by the ``checkpoint`` file, or None: the bundle prefix when
    ``<name>.index`` exists (what tf.train.latest_checkpoint returns), else ``<name>.npz``."""
    index = os.path.join(checkpoint_dir, 'checkpoint')
    if not os.path.exists(index):
        return None
    for line in open(index):
        if line.startswith('model_checkpoint_path:'):
            name = line.split(':', 1)[1].strip().strip('"')
            base = os.path.join(checkpoint_dir, os.path.basename(name))
            if os.path.exists(base + '.index'):
                return base
            return base + '.npz' if os.path.exists(base + '.npz') else None
    return None


def read_values(path, skipped=None):
    """name -> array of a checkpoint: a bundle prefix (or its .index file) or an .npz.  Bundle entries this reader
    cannot represent are left out (tf_bundle.read_bundle; their names go to ``skipped``)."""
    if path.endswith('.index'):
        path = path[:-len('.index')]
    if os.path.exists(path + '.index'):
        return tf_bundle.read_bundle(path, skipped=skipped)
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def load(path_or_dir, model, optimizer=None):
    """Restore variables (and, when given an optimizer, its slots and schedule).  The model must
    have been wired (one forward) or takes its variable shapes from the file.  Returns global_step."""
    path = latest(path_or_dir) if os.path.isdir(path_or_dir) else path_or_dir
    if path is None:
        raise FileNotFoundError("no checkpoint under %s" % path_or_dir)
    skipped = []
    values = read_values(path, skipped)
    needed = set(model._slots) if getattr(model, '_slots', None) else set()
    lost = [(n, why) for n, why in skipped if n in needed]
    if lost:
        raise tf_bundle.BundleError("checkpoint %s: %s" % (path, "; ".join("%s: %s" % nw for nw in lost)))
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
