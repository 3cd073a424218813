"""TensorFlow checkpoint bundles (``<prefix>.index`` + ``<prefix>.data-00000-of-00001``) without TensorFlow.

The reference saves and restores through tf.train.MonitoredTrainingSession(checkpoint_dir=...)
(recommendation/code/model_trainer.py:165) and lists what it wrote through tf.train.get_checkpoint_state
(model_trainer.py:47-65); what lands on disk is TensorFlow's "tensor bundle" (V2 saver format):

* ``<prefix>.index`` - an immutable sorted string table in the LevelDB table format: data blocks of
  prefix-compressed (key, value) entries with a restart array, each block followed by a 5-byte trailer
  (compression type, masked CRC-32C), an index block of block handles, and a 48-byte footer that ends in
  the magic number 0xdb4775248b80fb57.  Key "" holds a BundleHeaderProto, every other key is a variable
  name and holds a BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}.
* ``<prefix>.data-SSSSS-of-NNNNN`` - the raw little-endian tensor bytes, back to back, at those offsets.

This module restates that published format (TensorFlow and LevelDB are third-party dependencies the reference
does not vendor; no bundle ships with the reference and none can be produced here, so files written by
TensorFlow itself have not been read by this code: the reader is pinned by the format's own checksums and
known answers, and by round trips through the writer).  Snappy-compressed blocks (TensorFlow writes bundles
uncompressed, other table writers may not) are decoded by a small pure-Python decompressor.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
           17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
_DTYPE_CODES = {np.dtype(v): k for k, v in _DTYPES.items()}


class BundleError(ValueError):
    """The files are not a readable tensor bundle (bad magic, checksum, or an unsupported feature)."""


# ---- CRC-32C (Castagnoli), as LevelDB and TensorFlow mask it ------------------------------------------------
def _make_crc_table():
    tab = np.zeros(256, dtype=np.uint32)
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
        tab[i] = c
    return tab


_CRC_TABLE = _make_crc_table()
_CRC_LIST = [int(x) for x in _CRC_TABLE]


def crc32c(data, crc=0):
    """CRC-32C of ``data`` (bytes-like), continuing from ``crc``."""
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_LIST
    for b in memoryview(data).cast('B'):
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(crc):
    """LevelDB's masking of a stored CRC: rotate right by 15 bits and add a constant."""
    return (((crc >> 15) | (crc << 17)) + _MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - _MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ---- varints and the few protobuf messages a bundle uses -------------------------------------------------
def _put_varint(out, v):
    v &= (1 << 64) - 1
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)


def _get_varint(buf, pos):
    shift, v = 0, 0
    while True:
        if pos >= len(buf):
            raise BundleError("truncated varint")
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7
        if shift > 63:
            raise BundleError("varint too long")


def _pb_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message; length-delimited values as bytes."""
    pos = 0
    while pos < len(buf):
        tag, pos = _get_varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            if len(v) != n:
                raise BundleError("truncated protobuf field")
            pos += n
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise BundleError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _pb_tag(out, field, wt):
    _put_varint(out, (field << 3) | wt)


def encode_header(num_shards=1, producer=1):
    """BundleHeaderProto {num_shards = 1; endianness = 2 (LITTLE = 0, omitted); version = 3 {producer = 1}}."""
    out = bytearray()
    _pb_tag(out, 1, 0)
    _put_varint(out, num_shards)
    ver = bytearray()
    _pb_tag(ver, 1, 0)
    _put_varint(ver, producer)
    _pb_tag(out, 3, 2)
    _put_varint(out, len(ver))
    out += ver
    return bytes(out)


def decode_header(buf):
    h = {'num_shards': 0, 'endianness': 0, 'producer': 0}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            h['num_shards'] = v
        elif f == 2:
            h['endianness'] = v
        elif f == 3:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 1:
                    h['producer'] = v2
    return h


def encode_entry(dtype_code, shape, shard_id, offset, size, masked_crc):
    """BundleEntryProto; proto3 omits zero-valued scalars, as TensorFlow's serializer does."""
    out = bytearray()
    _pb_tag(out, 1, 0)
    _put_varint(out, dtype_code)
    sh = bytearray()
    for d in shape:
        dim = bytearray()
        if d:
            _pb_tag(dim, 1, 0)
            _put_varint(dim, d)
        _pb_tag(sh, 2, 2)
        _put_varint(sh, len(dim))
        sh += dim
    _pb_tag(out, 2, 2)
    _put_varint(out, len(sh))
    out += sh
    if shard_id:
        _pb_tag(out, 3, 0)
        _put_varint(out, shard_id)
    if offset:
        _pb_tag(out, 4, 0)
        _put_varint(out, offset)
    if size:
        _pb_tag(out, 5, 0)
        _put_varint(out, size)
    _pb_tag(out, 6, 5)
    out += struct.pack('<I', masked_crc)
    return bytes(out)


def decode_entry(buf):
    e = {'dtype': 0, 'shape': [], 'shard_id': 0, 'offset': 0, 'size': 0, 'crc32c': None, 'slices': 0}
    for f, _, v in _pb_fields(buf):
        if f == 1:
            e['dtype'] = v
        elif f == 2:
            for f2, _, v2 in _pb_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, v3 in _pb_fields(v2):
                        if f3 == 1:
                            size = v3 - (1 << 64) if v3 >> 63 else v3
                    e['shape'].append(size)
                elif f2 == 3 and v2:
                    raise BundleError("tensor of unknown rank in a bundle")
        elif f == 3:
            e['shard_id'] = v
        elif f == 4:
            e['offset'] = v
        elif f == 5:
            e['size'] = v
        elif f == 6:
            e['crc32c'] = v
        elif f == 7:
            e['slices'] += 1
    return e


# ---- Snappy (raw format) decompression, for tables written with compressed blocks ------------------------------
def snappy_decompress(buf):
    n, pos = _get_varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        if off == 0 or off > len(out):
            raise BundleError("bad snappy back reference")
        for _ in range(ln):              # a copy may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise BundleError("snappy length mismatch")
    return bytes(out)


# ---- LevelDB table ---------------------------------------------------------------------------------
def _read_block(buf, offset, size, verify=True):
    """Contents of the block at (offset, size): trailer checked, decompressed if need be."""
    end = offset + size
    if end + 5 > len(buf):
        raise BundleError("block handle past the end of the table")
    body = buf[offset:end]
    ctype = buf[end]
    if verify:
        stored = unmask_crc(struct.unpack_from('<I', buf, end + 1)[0])
        if crc32c(buf[offset:end + 1]) != stored:
            raise BundleError("block checksum mismatch at offset %d" % offset)
    if ctype == 0:
        return bytes(body)
    if ctype == 1:
        return snappy_decompress(bytes(body))
    raise BundleError("unknown block compression type %d" % ctype)


def _block_entries(block):
    """(key, value) pairs of one block, in order: keys are stored as (shared prefix length, suffix)."""
    if len(block) < 4:
        raise BundleError("block too short")
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    limit = len(block) - 4 - 4 * n_restarts
    if limit < 0:
        raise BundleError("bad restart array")
    pos, key = 0, b''
    while pos < limit:
        shared, pos = _get_varint(block, pos)
        non_shared, pos = _get_varint(block, pos)
        vlen, pos = _get_varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise BundleError("corrupt block entry")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _handle(buf, pos):
    off, pos = _get_varint(buf, pos)
    size, pos = _get_varint(buf, pos)
    return off, size, pos


def read_table(buf, verify=True):
    """All (key, value) pairs of a LevelDB-format table held in ``buf``, in key order."""
    if len(buf) < 48:
        raise BundleError("file too short for a table footer")
    footer = buf[-48:]
    if struct.unpack_from('<Q', footer, 40)[0] != TABLE_MAGIC:
        raise BundleError("bad table magic: not a TensorFlow checkpoint index")
    _, _, pos = _handle(footer, 0)                 # metaindex block (filters; unused by bundles)
    ioff, isize, _ = _handle(footer, pos)
    out = []
    for _, hv in _block_entries(_read_block(buf, ioff, isize, verify)):
        boff, bsize, _ = _handle(hv, 0)
        out.extend(_block_entries(_read_block(buf, boff, bsize, verify)))
    return out


class _BlockBuilder(object):
    def __init__(self, restart_interval):
        self.interval = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last = b''

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            m = min(len(key), len(self.last))
            while shared < m and key[shared] == self.last[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        _put_varint(self.buf, shared)
        _put_varint(self.buf, len(key) - shared)
        _put_varint(self.buf, len(value))
        self.buf += key[shared:]
        self.buf += value
        self.last = key
        self.count += 1

    def finish(self):
        out = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
        return out

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4


def _shortest_separator(a, b):
    """A short key k with a <= k < b (LevelDB's BytewiseComparator::FindShortestSeparator)."""
    m = min(len(a), len(b))
    i = 0
    while i < m and a[i] == b[i]:
        i += 1
    if i < m and a[i] < 0xFF and a[i] + 1 < b[i]:
        return a[:i] + bytes([a[i] + 1])
    return a


def _short_successor(a):
    for i, c in enumerate(a):
        if c != 0xFF:
            return a[:i] + bytes([c + 1])
    return a


def write_table(pairs, block_size=256 * 1024, restart_interval=16):
    """Serialise sorted (key, value) byte pairs as an uncompressed LevelDB-format table."""
    out = bytearray()
    index = _BlockBuilder(1)
    pending = None                                  # (last key of the finished block, its handle)

    def emit(contents):
        off = len(out)
        out.extend(contents)
        out.append(0)                               # kNoCompression
        out.extend(struct.pack('<I', mask_crc(crc32c(contents + b'\x00'))))
        h = bytearray()
        _put_varint(h, off)
        _put_varint(h, len(contents))
        return bytes(h)

    blk = _BlockBuilder(restart_interval)
    prev = None
    for key, value in pairs:
        if prev is not None and key <= prev:
            raise BundleError("table keys must be strictly increasing")
        if pending is not None:
            index.add(_shortest_separator(pending[0], key), pending[1])
            pending = None
        blk.add(key, value)
        prev = key
        if blk.size() >= block_size:
            pending = (key, emit(blk.finish()))
            blk = _BlockBuilder(restart_interval)
    if blk.buf:
        pending = (prev, emit(blk.finish()))
    if pending is not None:
        index.add(_short_successor(pending[0]), pending[1])
    meta_handle = emit(_BlockBuilder(restart_interval).finish())
    index_handle = emit(index.finish())
    footer = bytearray(meta_handle + index_handle)
    footer.extend(b'\x00' * (40 - len(footer)))
    footer.extend(struct.pack('<Q', TABLE_MAGIC))
    out.extend(footer)
    return bytes(out)


# ---- bundles -------------------------------------------------------------------------------------------
def _shard_name(prefix, shard, n):
    return '%s.data-%05d-of-%05d' % (prefix, shard, n)


def list_variables(prefix, verify=True):
    """[(name, shape, numpy dtype)] of a bundle, in key order (what tf.train.list_variables returns)."""
    with open(prefix + '.index', 'rb') as f:
        pairs = read_table(f.read(), verify)
    out = []
    for k, v in pairs:
        if k == b'':
            continue
        e = decode_entry(v)
        out.append((k.decode(), tuple(e['shape']), _DTYPES.get(e['dtype'])))
    return out


def read_bundle(prefix, verify=True, skipped=None):
    """name -> numpy array for every tensor of the bundle ``prefix`` (.index + .data-* files) that this reader can
    represent.  Entries it cannot - a dtype without a numpy counterpart here (DT_STRING such as TensorFlow 2's
    ``_CHECKPOINTABLE_OBJECT_GRAPH``, bfloat16 ...) or a partitioned (sliced) variable - are left out, their names
    and the reason appended to ``skipped`` when a list is given: a checkpoint that carries such an entry beside
    the model's variables still restores; whoever needs the entry finds it missing."""
    with open(prefix + '.index', 'rb') as f:
        pairs = read_table(f.read(), verify)
    if not pairs or pairs[0][0] != b'':
        raise BundleError("bundle index has no header entry")
    header = decode_header(pairs[0][1])
    if header['endianness'] != 0:
        raise BundleError("big-endian bundles are not supported")
    n_shards = max(1, header['num_shards'])
    shards = {}
    out = {}
    for k, v in pairs[1:]:
        e = decode_entry(v)
        name = k.decode()
        if e['slices'] or e['dtype'] not in _DTYPES:
            if skipped is not None:
                skipped.append((name, "partitioned (sliced) variable" if e['slices']
                                else "unsupported dtype code %d" % e['dtype']))
            continue
        dt = np.dtype(_DTYPES[e['dtype']])
        count = int(np.prod(e['shape'], dtype=np.int64)) if e['shape'] else 1
        if count * dt.itemsize != e['size']:
            raise BundleError("%s: size %d does not match shape %s of %s" % (name, e['size'], e['shape'], dt))
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap(_shard_name(prefix, sid, n_shards), dtype=np.uint8, mode='r') \
                if os.path.getsize(_shard_name(prefix, sid, n_shards)) else np.zeros(0, np.uint8)
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if len(raw) != e['size']:
            raise BundleError("%s: data shard %d is too short" % (name, sid))
        raw = np.array(raw)                            # off the memory map
        if verify and e['crc32c'] is not None and crc32c(raw.tobytes()) != unmask_crc(e['crc32c']):
            raise BundleError("%s: tensor checksum mismatch" % name)
        out[name] = raw.view(dt).reshape(e['shape'])
    return out


def write_bundle(prefix, tensors):
    """Write name -> array as a one-shard bundle: ``prefix.index`` and ``prefix.data-00000-of-00001``."""
    names = sorted(tensors, key=lambda s: s.encode())
    pairs = [(b'', encode_header(1))]
    offset = 0
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    with open(_shard_name(prefix, 0, 1), 'wb') as f:
        for name in names:
            a = np.asarray(tensors[name])          # (ascontiguousarray would turn a scalar into shape (1,))
            if a.dtype not in _DTYPE_CODES:
                raise BundleError("%s: dtype %s has no TensorFlow code here" % (name, a.dtype))
            if a.dtype.byteorder == '>':
                a = a.astype(a.dtype.newbyteorder('<'))
            raw = a.tobytes()
            f.write(raw)
            pairs.append((name.encode(), encode_entry(_DTYPE_CODES[a.dtype], a.shape, 0, offset, len(raw),
                                                       mask_crc(crc32c(raw)))))
            offset += len(raw)
    with open(prefix + '.index', 'wb') as f:
        f.write(write_table(pairs))
    return prefix
