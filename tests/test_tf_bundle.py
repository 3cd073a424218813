"""TensorFlow checkpoint-bundle reader / writer (recommendation_amd/tf_bundle.py), CPU only.

No bundle ships with the reference (model_trainer.py:165 would write one) and TensorFlow is absent, so the format
code is pinned by the format's own published known answers (CRC-32C check value, LevelDB's CRC mask, the table
magic, protobuf wire encodings spelled out by hand below) and by round trips: parity with files written by
TensorFlow itself is unpinned.
"""
import os
import struct

import numpy as np
import pytest

from recommendation_amd import tf_bundle as tb


def test_crc32c_known_answers():
    assert tb.crc32c(b'123456789') == 0xE3069283          # the CRC-32C check value (RFC 3720 appendix B.4 family)
    assert tb.crc32c(b'\x00' * 32) == 0x8A9136AA           # RFC 3720 B.4: 32 bytes of zeros
    assert tb.crc32c(b'\xff' * 32) == 0x62A8AB43           # RFC 3720 B.4: 32 bytes of ones
    assert tb.crc32c(bytes(range(32))) == 0x46DD794E       # RFC 3720 B.4: incrementing bytes
    # continuation == one pass
    assert tb.crc32c(b'6789', tb.crc32c(b'12345')) == 0xE3069283


def test_crc_mask_is_leveldbs_and_inverts():
    crc = tb.crc32c(b'foo')
    m = tb.mask_crc(crc)
    assert m != crc and tb.unmask_crc(m) == crc
    assert tb.mask_crc(0) == 0xa282ead8
    assert tb.unmask_crc(tb.unmask_crc(tb.mask_crc(tb.mask_crc(crc)))) == crc


def test_varint_and_entry_wire_bytes():
    out = bytearray()
    tb._put_varint(out, 300)
    assert bytes(out) == b'\xac\x02'
    assert tb._get_varint(b'\xac\x02', 0) == (300, 2)
    # BundleEntryProto{dtype: DT_FLOAT(1), shape{dim{size:256} dim{size:8}}, offset: 4, size: 8192, crc32c: 0x01020304}
    want = (b'\x08\x01' + b'\x12\x09' + b'\x12\x03\x08\x80\x02' + b'\x12\x02\x08\x08' + b'\x20\x04' + b'\x28\x80\x40' +
            b'\x35\x04\x03\x02\x01')
    got = tb.encode_entry(1, (256, 8), 0, 4, 8192, 0x01020304)
    assert got == want
    e = tb.decode_entry(want)
    assert (e['dtype'], e['shape'], e['shard_id'], e['offset'], e['size'], e['crc32c']) == (1, [256, 8], 0, 4, 8192, 0x01020304)
    # scalar: empty shape message
    assert tb.decode_entry(tb.encode_entry(9, (), 0, 0, 8, 7))['shape'] == []
    assert tb.decode_header(tb.encode_header(1)) == {'num_shards': 1, 'endianness': 0, 'producer': 1}
    assert tb.encode_header(1) == b'\x08\x01\x1a\x02\x08\x01'


def test_table_layout_and_prefix_compression():
    pairs = [(b'', b'H'), (b'dense/bias', b'1'), (b'dense/bias/Adam', b'22'), (b'dense/bias/Adam_1', b'333'),
             (b'dense/kernel', b'4444'), (b'global_step', b'5')]
    raw = tb.write_table(pairs)
    assert struct.unpack('<Q', raw[-8:])[0] == 0xdb4775248b80fb57
    assert len(raw) > 48 and tb.read_table(raw) == pairs
    # the second key shares nothing with "", the third shares the whole second key: shared=10, non_shared=5, vlen=2
    assert b'\x0a\x05\x02/Adam22' in raw
    # a flipped byte in a data block is caught by the block checksum
    bad = bytearray(raw)
    bad[5] ^= 0x40
    with pytest.raises(tb.BundleError):
        tb.read_table(bytes(bad))
    with pytest.raises(tb.BundleError):
        tb.read_table(raw[:-1] + b'\x00')
    with pytest.raises(tb.BundleError):
        tb.write_table([(b'b', b''), (b'a', b'')])


def test_table_with_many_blocks_and_restarts():
    rng = np.random.default_rng(0)
    keys = sorted({('var_%05d/slot_%d' % (rng.integers(0, 3000), rng.integers(0, 4))).encode() for _ in range(4000)})
    pairs = [(k, bytes(rng.integers(0, 256, size=int(rng.integers(0, 60)), dtype=np.uint8))) for k in keys]
    raw = tb.write_table(pairs, block_size=512, restart_interval=4)
    assert tb.read_table(raw) == pairs
    raw1 = tb.write_table(pairs[:1])
    assert tb.read_table(raw1) == pairs[:1]


def test_snappy_decoder_on_hand_built_streams():
    # literal "abcd", then a 1-byte-offset copy of length 4 from 4 back, then a 2-byte-offset copy of length 5
    # reaching 8 back: "abcd" "abcd" "abcda"
    stream = bytes([13]) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((4 - 4) << 2) | 1, 4]) + bytes([((5 - 1) << 2) | 2, 8, 0])
    assert tb.snappy_decompress(stream) == b'abcdabcdabcda'
    # overlapping copy (run-length): literal "x", copy length 7 at offset 1
    assert tb.snappy_decompress(bytes([8, 0]) + b'x' + bytes([((7 - 4) << 2) | 1, 1])) == b'x' * 8
    # a table whose data block is marked snappy decodes through it
    pairs = [(b'', b'hdr'), (b'k', b'v' * 40)]
    block = tb._BlockBuilder(16)
    for k, v in pairs:
        block.add(k, v)
    plain = block.finish()
    lit = bytearray()
    tb._put_varint(lit, len(plain))
    lit += bytes([60 << 2, len(plain) - 1]) + plain          # one literal with a 1-byte length
    comp = bytes(lit)
    out = bytearray(comp) + b'\x01' + struct.pack('<I', tb.mask_crc(tb.crc32c(comp + b'\x01')))
    h = bytearray()
    tb._put_varint(h, 0)
    tb._put_varint(h, len(comp))
    idx = tb._BlockBuilder(1)
    idx.add(b'l', bytes(h))
    ib = idx.finish()
    ioff = len(out)
    out += ib + b'\x00' + struct.pack('<I', tb.mask_crc(tb.crc32c(ib + b'\x00')))
    mb = tb._BlockBuilder(16).finish()
    moff = len(out)
    out += mb + b'\x00' + struct.pack('<I', tb.mask_crc(tb.crc32c(mb + b'\x00')))
    footer = bytearray()
    for off, size in ((moff, len(mb)), (ioff, len(ib))):
        tb._put_varint(footer, off)
        tb._put_varint(footer, size)
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', tb.TABLE_MAGIC)
    assert tb.read_table(bytes(out + footer)) == pairs


def test_bundle_round_trip_with_the_reference_names(tmp_path):
    """Everything a reference checkpoint holds (the 22 variables of r_model.py:31-58 under tf.layers.dense's names,
    Adam's slots and schedule variables, global_step) written as a bundle and read back bit for bit."""
    from oracle import colour_model as cm
    params = cm.init_params(seed=3, dtype=np.float32)
    rng = np.random.default_rng(1)
    values = dict(params)
    for n, p in params.items():
        values[n + '/Adam'] = rng.standard_normal(p.shape).astype(np.float32)
        values[n + '/Adam_1'] = rng.random(p.shape).astype(np.float32)
    values['beta1_power'] = np.float32(0.9) ** 7
    values['beta2_power'] = np.float32(0.999) ** 7
    values['global_step'] = np.int64(7)
    prefix = str(tmp_path / 'model.ckpt-7')
    tb.write_bundle(prefix, values)
    assert sorted(os.listdir(tmp_path)) == ['model.ckpt-7.data-00000-of-00001', 'model.ckpt-7.index']
    got = tb.read_bundle(prefix)
    assert sorted(got) == sorted(values) and len(got) == 22 * 3 + 3
    for n, v in values.items():
        assert got[n].dtype == np.asarray(v).dtype and got[n].shape == np.asarray(v).shape, n
        assert np.array_equal(got[n], v), n
    listed = dict((n, (s, d)) for n, s, d in tb.list_variables(prefix))
    assert listed['dense_2/kernel'] == ((256, 8), np.float32) and listed['global_step'] == ((), np.int64)
    assert sum(int(np.prod(s)) for n, (s, d) in listed.items() if n in params) == 5910
    # the data file is the tensors back to back in key order
    names = sorted(values, key=lambda s: s.encode())
    blob = b''.join(np.asarray(values[n]).tobytes() for n in names)
    assert open(prefix + '.data-00000-of-00001', 'rb').read() == blob
    # corruption of a tensor byte is caught by the entry checksum
    with open(prefix + '.data-00000-of-00001', 'r+b') as f:
        f.seek(100)
        b = f.read(1)
        f.seek(100)
        f.write(bytes([b[0] ^ 1]))
    with pytest.raises(tb.BundleError):
        tb.read_bundle(prefix)
    assert tb.read_bundle(prefix, verify=False)['global_step'] == 7


def test_first_layer_split_and_merge():
    from recommendation_amd import checkpoint as ck
    k = np.arange(256 * 8, dtype=np.float32).reshape(256, 8)
    R, K = ck.split_first_layer(k, 128)
    assert R.shape == K.shape == (128, 8) and np.array_equal(R[5], k[5]) and np.array_equal(K[5], k[133])
    assert np.array_equal(ck.merge_first_layer(R, K), k)
    with pytest.raises(ValueError):
        ck.split_first_layer(k, 100)


def test_entries_the_reader_cannot_represent_are_skipped_not_fatal(tmp_path):
    """A bundle that carries, beside the model's variables, a DT_STRING entry (TensorFlow 2 writes
    _CHECKPOINTABLE_OBJECT_GRAPH that way) or a bfloat16 tensor still reads: those entries are left out and named."""
    prefix = str(tmp_path / 'model.ckpt-7')
    w = np.arange(12, dtype=np.float32).reshape(3, 4)
    blob = b'\x0a\x03abc'
    bf = np.array([1, 2, 3], dtype=np.uint16).tobytes()
    raw = w.tobytes() + blob + bf
    with open(tb._shard_name(prefix, 0, 1), 'wb') as f:
        f.write(raw)
    pairs = [(b'', tb.encode_header(1)),
             (b'_CHECKPOINTABLE_OBJECT_GRAPH', tb.encode_entry(7, (), 0, len(w.tobytes()), len(blob), tb.mask_crc(tb.crc32c(blob)))),
             (b'dense/kernel', tb.encode_entry(1, w.shape, 0, 0, len(w.tobytes()), tb.mask_crc(tb.crc32c(w.tobytes())))),
             (b'half', tb.encode_entry(14, (3,), 0, len(w.tobytes()) + len(blob), len(bf), tb.mask_crc(tb.crc32c(bf))))]
    with open(prefix + '.index', 'wb') as f:
        f.write(tb.write_table(pairs))
    skipped = []
    got = tb.read_bundle(prefix, skipped=skipped)
    assert list(got) == ['dense/kernel'] and np.array_equal(got['dense/kernel'], w)
    assert [n for n, _ in skipped] == ['_CHECKPOINTABLE_OBJECT_GRAPH', 'half']
    assert tb.read_bundle(prefix).keys() == got.keys()          # no list given: silently left out


class _FakeModel(object):
    """What checkpoint.save / load touch of a model."""

    def __init__(self, values):
        import torch
        self.variables = {k: torch.from_numpy(np.array(v, dtype=np.float32)) for k, v in values.items()}
        self._slots = {}

    def load_variables(self, values):
        import torch
        for k, v in values.items():
            self.variables[k] = torch.from_numpy(np.array(v, dtype=np.float32))


def test_a_step_saved_twice_keeps_one_container(tmp_path):
    """latest() prefers the bundle of a name; a later .npz save of the same step must not be shadowed by it."""
    from recommendation_amd import checkpoint
    d = str(tmp_path)
    checkpoint.save(d, _FakeModel({'dense/kernel': [[1.0, 2.0]]}), step=5, fmt='bundle')
    assert os.path.exists(os.path.join(d, 'model.ckpt-5.index'))
    checkpoint.save(d, _FakeModel({'dense/kernel': [[3.0, 4.0]]}), step=5, fmt='npz')
    assert not os.path.exists(os.path.join(d, 'model.ckpt-5.index'))
    assert not os.path.exists(os.path.join(d, 'model.ckpt-5.data-00000-of-00001'))
    m = _FakeModel({})
    assert checkpoint.load(d, m) == 5
    assert m.variables['dense/kernel'].tolist() == [[3.0, 4.0]]
    checkpoint.save(d, _FakeModel({'dense/kernel': [[5.0, 6.0]]}), step=5, fmt='bundle')
    assert not os.path.exists(os.path.join(d, 'model.ckpt-5.npz'))
    m = _FakeModel({})
    checkpoint.load(d, m)
    assert m.variables['dense/kernel'].tolist() == [[5.0, 6.0]]
