"""Host-side logic that needs no GPU: the record reader, file sharding, lazy one-hot
features, and the rule that the product path fails loudly without the HIP library/device."""
import numpy as np
import pytest
import torch

from oracle import colour_model as cm
from recommendation_amd import ops
from recommendation_amd.r_model import (Dataset, InvalidArgumentError, OutOfRangeError, Sample, SparseFeature,
                                        concat, parse_record)


def write_records(path, batch, n, with_label=True):
    with open(path, 'w') as f:
        for i in range(n):
            f.write(cm.format_record(batch, i, with_label) + '\n')


def test_reader_matches_oracle_grammar(tmp_path):
    batch = cm.synthetic_batch(10, seed=3)
    p = tmp_path / 'train_sample_1.csv'
    write_records(p, batch, 10)
    ds = Sample.read_label_data(str(p), batch_size=4)
    it = ds.make_one_shot_iterator()
    cols = it.get_next()
    assert len(cols) == 7 and all(c.dtype == torch.int32 for c in cols)
    want = next(cm.batches_from_lines(open(p).readlines(), 4))
    for c, k in zip(cols, want):
        assert np.array_equal(c.numpy(), want[k]), k
    # repeat(-1): 10 records, batch 4 -> 4,4,4(2 old + 2 from the next epoch)...
    sizes = [it.get_next()[0].shape[0] for _ in range(5)]
    assert sizes == [4, 4, 4, 4, 4]


def test_read_csv_one_epoch_keeps_partial_batch_then_out_of_range(tmp_path):
    batch = cm.synthetic_batch(7, seed=4, with_label=False)
    p = tmp_path / 'sample.csv'
    write_records(p, batch, 7, with_label=False)
    it = Sample.read_csv(str(p), batch_size=5).make_one_shot_iterator()
    a, b = it.get_next(), it.get_next()
    assert len(a) == 6 and a[0].shape[0] == 5 and b[0].shape[0] == 2 and a[2].shape == (5, 6)
    with pytest.raises(OutOfRangeError):
        it.get_next()


def test_file_sharding_files_index_parallelism(tmp_path):
    paths = []
    for k in range(5):
        b = cm.synthetic_batch(2, seed=k)
        b['user'][:] = k
        p = tmp_path / ('f%d.csv' % k)
        write_records(p, b, 2)
        paths.append(str(p))
    for index, want in ((0, [0, 0, 2, 2, 4, 4]), (1, [1, 1, 3, 3])):
        it = Sample.read_label_data(paths, batch_size=len(want), parallelism=2, index=index).make_one_shot_iterator()
        assert it.get_next()[0].tolist() == want


def test_defaults_and_errors():
    assert parse_record("  1,2,3,4,5,6    ", 7) == [0, 0, [1, 2, 3, 4, 5, 6], 0, [0, 0, 0, 0, 0, 0], 0, 0]
    with pytest.raises(InvalidArgumentError):
        parse_record("1 2 3", 6)
    with pytest.raises(InvalidArgumentError):
        parse_record("1 2 1,2,3,4,5,6 x 1,2,3,4,5,6 0", 6)
    with pytest.raises(InvalidArgumentError):
        parse_record("99999999999 2 1,2,3,4,5,6 0 1,2,3,4,5,6 0", 6)


def test_flink_dataset_reads_any_iterable():
    batch = cm.synthetic_batch(3, seed=5)
    lines = [cm.format_record(batch, i) for i in range(3)]
    cols = Sample.read_flink_dataset(lines, 2).make_one_shot_iterator().get_next()
    assert np.array_equal(cols[6].numpy(), batch['label'][:2])


def test_sparse_feature_concat_is_the_one_hot_concat():
    rec = torch.tensor([[0, 2, 5], [1, 3, 4]], dtype=torch.int32)
    click = torch.tensor([[2], [-1]], dtype=torch.int32)
    f = concat([SparseFeature(rec, 6), SparseFeature(click, 6)], axis=1)
    assert f.size == 12 and f.ids.tolist() == [[0, 2, 5, 8], [1, 3, 4, -1]]
    dense = np.concatenate([cm.input_to_n_hot(rec.numpy(), 6, 3), cm.input_to_one_hot_plus(click.numpy(), 6)], 1)
    rebuilt = np.zeros_like(dense)
    for b in range(2):
        for v in f.ids[b].tolist():
            if v >= 0:
                rebuilt[b, v] = 1.0
    assert np.array_equal(rebuilt, dense)


def test_no_cpu_fallback():
    with pytest.raises(ops.MiError):
        ops.embag_fwd(torch.zeros((4, 4)), torch.tensor([0, 4]), 1, torch.zeros((2, 1), dtype=torch.int32))
    with pytest.raises(ops.MiError):
        ops.linear_fwd(torch.zeros((2, 3)), torch.zeros((3, 2)))


def test_gradient_sink_allows_a_store_only_for_a_single_use_of_a_clean_slice():
    """The rule behind storing a variable's gradient straight into the flat buffer (r_model._GradSink): exactly
    one use since the last finished backward and nothing written since the slice was zeroed."""
    from recommendation_amd.r_model import _GradSink
    view = object()
    s = _GradSink(view)
    assert s.take() is None and s.dirty          # never used in a forward: nothing to store, and now marked written
    s.dirty = False
    s.uses = 1
    assert s.take() is view and s.dirty          # the one use of a clean slice: store
    assert s.take() is None                      # a second backward through the same op: accumulate
    s.dirty, s.uses = False, 2
    assert s.take() is None and s.take() is None  # two ops share the variable: both accumulate
    s.uses = 1                                   # backward_done() + one forward, but no zero_grad in between
    assert s.dirty and s.take() is None
