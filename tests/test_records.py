"""The host-side tokenizer of the record wire format against the per-record Python parser
(recommendation/code/r_model.py:143-177 restated): same integers, same errors."""
import numpy as np
import pytest

from recommendation_amd.r_model import Dataset, InvalidArgumentError, parse_record, parse_records


def _records(n, seed, n_fields=7):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        f = [str(rng.integers(0, 100)), str(rng.integers(0, 20)),
             ','.join(str(x) for x in sorted(rng.choice(128, 6, replace=False))), str(rng.integers(-1, 128)),
             ','.join(str(x) for x in sorted(rng.choice(128, 6, replace=False))), str(rng.integers(-1, 128))]
        if n_fields == 7:
            f.append(str(rng.integers(-1, 128)))
        out.append(' '.join(f))
    return out


@pytest.mark.parametrize("n_fields", [6, 7])
def test_tokenizer_equals_python_parser(n_fields):
    lines = _records(500, 1, n_fields)
    lines[3] = lines[3].replace(' ', '  ', 1)[:0] + ' '.join(['', ''] + lines[3].split(' ')[2:])   # empty scalars -> 0
    lines[5] = ' '.join(lines[5].split(' ')[:2] + [''] + lines[5].split(' ')[3:])                  # empty list -> zeros
    text = '\r\n'.join(lines[:250]) + '\n\n' + '\n'.join(lines[250:]) + '\n'
    cols, used = parse_records(text, n_fields)
    assert used == len(text)
    want = [parse_record(l, n_fields) for l in lines]
    for k in range(n_fields):
        assert cols[k].dtype == np.int32
        assert np.array_equal(cols[k], np.asarray([w[k] for w in want], dtype=np.int32)), k


def test_tokenizer_stops_at_max_rows_and_reports_consumed():
    lines = _records(10, 2)
    text = '\n'.join(lines) + '\n'
    cols, used = parse_records(text, 7, max_rows=4)
    assert len(cols[0]) == 4 and used == len('\n'.join(lines[:4])) + 1
    rest, _ = parse_records(text[used:], 7)
    assert len(rest[0]) == 6


@pytest.mark.parametrize("bad", ["1 2 3,4,5,6,7,8 9 1,2,3,4,5,6",                 # 5 fields
                                 "1 2 3,4,5,6,7,8 9 1,2,3,4,5,6 7 8 9",            # 8 fields
                                 "1 x 3,4,5,6,7,8 9 1,2,3,4,5,6 7 8",              # not a number
                                 "1 2 3,4,5,6,7 9 1,2,3,4,5,6 7 8",                # short list
                                 "1 2 3,4,5,6,7,8,9 9 1,2,3,4,5,6 7 8",            # long list
                                 "1 2 3,4,5,6,7,8 99999999999 1,2,3,4,5,6 7 8",    # not an int32
                                 "1 2 3,4,5,6,7,8 9 1,2,3,4,5,a 7 8"])
def test_tokenizer_rejects_what_decode_csv_rejects(bad):
    good = _records(2, 3)
    with pytest.raises(InvalidArgumentError):
        parse_records('\n'.join([good[0], bad, good[1]]), 7)


def test_dataset_fast_path_equals_python_path():
    lines = _records(23, 4)
    src = lambda: iter(l + '\n' for l in lines)
    fast = list(Dataset(src, 7, 5, 2))
    slow = list(Dataset(src, 7, 5, 2, fast=False))
    assert len(fast) == len(slow) == 10          # 46 records in batches of 5, final partial batch kept
    for a, b in zip(fast, slow):
        for x, y in zip(a, b):
            assert x.dtype == y.dtype and x.shape == y.shape and bool((x == y).all())
