"""The callers either side of the hot path: ModelInference post-processing
(inference_service.py:36-93), checkpoints under TensorFlow's variable names
(model_trainer.py:165), and the synthetic train-to-imitate pipeline (data.py:67-296)."""
import os

import numpy as np
import pytest
import torch

from oracle import colour_model as cm
from tests.test_colour_model_gpu import model_with, oracle_params, rel_err, to_dev

pytestmark = pytest.mark.gpu


def test_model_inference_strings_and_threshold(dev):
    from recommendation_amd.inference import ModelInference, threshold
    params = oracle_params(5)
    batch = cm.synthetic_batch(64, seed=9, with_label=False)
    records = [cm.format_record(batch, i, with_label=False) for i in range(64)]
    mi = ModelInference(model=model_with(params, dev))
    got = mi.inference(records)
    p64 = {n: p.astype(np.float64) for n, p in params.items()}
    logits = cm.forward(cm.features(batch, dtype=np.float64), p64)
    idx, val = cm.output(logits, 6)
    assert got == [','.join(str(x) for x in sorted(r)) for r in idx.tolist()]
    click = mi.inference_click(records)
    want = [int(i[0]) if v[0] > threshold else -1 for i, v in zip(idx, val)]
    near = np.abs(val[:, 0] - threshold) < 1e-6                   # a probability on the threshold may fall either way
    assert [c for c, n in zip(click, near) if not n] == [w for w, n in zip(want, near) if not n]
    assert -1 in click or max(val[:, 0]) > threshold               # the synthetic weights exercise at least one side
    assert mi.inference_click(records[0]) == click[:1]             # a single record string (scalar squeeze path)


@pytest.mark.parametrize("fmt", ["npz", "bundle"])
def test_checkpoint_roundtrip_continues_training_bitwise(dev, tmp_path, fmt):
    """Both containers: the .npz and TensorFlow's bundle (.index + .data-00000-of-00001, what the reference's
    MonitoredTrainingSession writes, model_trainer.py:165)."""
    from recommendation_amd import checkpoint
    from recommendation_amd.trainer import AdamOptimizer, train_step
    params = oracle_params(7)
    a = model_with(params, dev)
    opt_a = AdamOptimizer(1e-3)
    batches = [tuple(torch.from_numpy(v).to(dev) for v in cm.synthetic_batch(128, seed=20 + i).values())
               for i in range(6)]
    for b in batches[:3]:
        train_step(a, opt_a, b)
    path = checkpoint.save(str(tmp_path), a, opt_a, fmt=fmt)
    if fmt == 'npz':
        with np.load(path) as z:
            names = set(z.files)
    else:
        from recommendation_amd import tf_bundle
        assert checkpoint.latest(str(tmp_path)) == path == str(tmp_path / 'model.ckpt-3')
        names = set(n for n, _, _ in tf_bundle.list_variables(path))
    # TensorFlow's names: 11 dense layers, their Adam slots, the schedule and the step
    for k in range(11):
        base = 'dense' if k == 0 else 'dense_%d' % k
        for v in ('/kernel', '/bias'):
            assert base + v in names and base + v + '/Adam' in names and base + v + '/Adam_1' in names
    assert {'beta1_power', 'beta2_power', 'global_step'} <= names
    assert len(names) == 11 * 2 * 3 + 3
    b_model = model_with(oracle_params(8), dev)                    # different weights, then restored
    with torch.no_grad():
        b_model.forward(b_model.features({k: v for k, v in zip(cm._KEYS[:6], batches[0][:6])}))
    opt_b = AdamOptimizer(1e-3)
    assert checkpoint.load(str(tmp_path), b_model, opt_b) == 3
    for b in batches[3:]:
        ra, rb = train_step(a, opt_a, b), train_step(b_model, opt_b, b)
        assert torch.equal(ra[3], rb[3])
    assert torch.equal(a.parameters(), b_model.parameters())
    assert torch.equal(opt_a.m, opt_b.m) and torch.equal(opt_a.v, opt_b.v)


def test_imitation_pipeline_trains(dev, tmp_path):
    """create_data -> teacher top-6 -> pair by user -> teacher label; a student then learns from it."""
    from recommendation_amd import data
    from recommendation_amd.r_model import RecommendationModel, Sample
    from recommendation_amd.trainer import AdamOptimizer, train_step
    d = str(tmp_path)
    teacher, path = data.pipeline(d, num=3000, device=dev, seed=3)
    for name in ('users.csv', 'sample.csv', 'org_sample_1.csv', 'no_label_sample_1.csv', 'train_sample_1.csv'):
        assert os.path.getsize(os.path.join(d, name)) > 0
    lines = open(path).read().splitlines()
    assert 1000 <= len(lines) <= 1500                              # pairs of records of the same user
    first = lines[0].split(' ')
    assert len(first) == 7 and len(first[2].split(',')) == 6 and -1 <= int(first[6]) < 128
    # every label is what the teacher says about that record
    cols = next(iter(Sample.read_label_data(path, batch_size=len(lines))))
    with torch.no_grad():
        logits = teacher.forward(teacher.features({k: cols[i] for i, k in enumerate(cm._KEYS[:6])}))
    idx, val = teacher.top_one_output(logits)
    want = torch.where(val > data.threshold, idx.to(torch.int32), torch.full_like(idx, -1).to(torch.int32))
    assert torch.equal(want.cpu(), cols[6])
    student = RecommendationModel(128, 6, 100, 20, device=dev, seed=99)
    opt = AdamOptimizer(1e-2)
    it = iter(Sample.read_label_data(path, batch_size=256))
    losses = [float(train_step(student, opt, next(it))[3]) for _ in range(60)]
    assert np.mean(losses[-10:]) < np.mean(losses[:10])


def _record_file(path, n, seed):
    batch = cm.synthetic_batch(n, seed=seed)
    with open(path, 'w') as f:
        for i in range(n):
            f.write(cm.format_record(batch, i) + "\n")
    return path


@pytest.mark.parametrize("graph", [False, True])
def test_train_resumes_from_checkpoint_dir_bit_for_bit(dev, tmp_path, graph):
    """train(checkpoint_dir=...) is MonitoredTrainingSession(checkpoint_dir=...) of the reference's loop
    (local_train_utils.py:54-58): a run stopped at step 7 and started again on the same directory restores
    variables, Adam slots, schedule and global step, runs on to the same last_step, and - with the input resumed
    where the restored steps left it - ends with exactly the bits of the uninterrupted run.  With the input
    starting over (the reference's behaviour) it still stops at last_step."""
    from recommendation_amd import checkpoint
    from recommendation_amd.trainer import train
    path = _record_file(str(tmp_path / 'train.csv'), 700, seed=41)
    kw = dict(batch_size=50, learning_rate=1e-3, device=dev, seed=5, log_every=0, graph=graph)
    whole, opt_w, _ = train(path, last_step=12, **kw)
    d = str(tmp_path / 'ckpt')
    part, opt_p, _ = train(path, last_step=7, checkpoint_dir=d, save_checkpoint_steps=5, **kw)
    assert sorted(f for f in os.listdir(d) if f.endswith('.npz')) == ['model.ckpt-5.npz', 'model.ckpt-7.npz']
    assert checkpoint.latest(d).endswith('model.ckpt-7.npz') and opt_p.t == 7
    resumed, opt_r, _ = train(path, last_step=12, checkpoint_dir=d, save_checkpoint_steps=5, resume_input=True, **kw)
    assert opt_r.t == 12 and checkpoint.latest(d).endswith('model.ckpt-12.npz')
    for n in whole.variables:
        assert torch.equal(whole.variables[n], resumed.variables[n]), n
    assert torch.equal(opt_w.m, opt_r.m) and torch.equal(opt_w.v, opt_r.v)
    # the reference's way: the input starts over after a restore; the loop still ends at the global last_step
    again, opt_a, _ = train(path, last_step=14, checkpoint_dir=d, **kw)
    assert opt_a.t == 14 and checkpoint.latest(d).endswith('model.ckpt-14.npz')
    # nothing left to do: the restored step already is last_step
    _, opt_n, _ = train(path, last_step=14, checkpoint_dir=d, **kw)
    assert opt_n.t == 14


def test_graphed_step_takes_the_eager_path_for_any_other_column_shape(dev):
    """Same batch size, recommend lists of another width: the captured step's static inputs do not fit, so the call
    must run the eager step (it used to try to reshape the column into the captured buffer)."""
    from recommendation_amd.r_model import InvalidArgumentError
    from recommendation_amd.trainer import AdamOptimizer, GraphedTrainStep, train_step
    params = oracle_params(3)
    a, b = model_with(params, dev), model_with(params, dev)
    opt_a, opt_b = AdamOptimizer(1e-3), AdamOptimizer(1e-3)
    cols = [tuple(torch.from_numpy(v).to(dev) for v in cm.synthetic_batch(64, seed=70 + i).values()) for i in range(3)]
    g = GraphedTrainStep(a, opt_a, cols[0])
    for c in cols[:2]:
        ra, rb = g(c), train_step(b, opt_b, c)
        assert torch.equal(ra[3], rb[3])
    odd = list(cols[2])
    odd[2] = odd[2][:, :5].contiguous()                      # five recommended colours instead of six
    # the model's bag width is recommend_num = 6 (r_model.py:72): both paths must refuse the column the same way
    with pytest.raises((InvalidArgumentError, RuntimeError, ValueError)):
        train_step(b, opt_b, tuple(odd))
    with pytest.raises((InvalidArgumentError, RuntimeError, ValueError)):
        g(tuple(odd))
    ra, rb = g(cols[2]), train_step(b, opt_b, cols[2])          # and the graphed step is intact afterwards
    assert torch.equal(ra[3], rb[3]) and torch.equal(a.parameters(), b.parameters())
