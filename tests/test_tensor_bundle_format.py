"""keras_files.read_tensor_bundle against index / data files assembled HERE from the published format constants - LevelDB
table format (table/format.cc: block trailer, footer, magic 0xdb4775248b80fb57), snappy format_description.txt,
tensorflow/core/protobuf/tensor_bundle.proto field numbers, types.proto dtype codes, CRC-32C known answers - not by the
repository's own fixture writer (tests/golden/make_savedmodel_fixture.py), which shares its author's reading of the format
with the reader.  Covers what a TensorFlow-written `variables/` directory can contain that the committed fixture does not:
several shards, snappy-compressed index blocks, prefix-compressed keys across restart points, DT_HALF / DT_DOUBLE /
DT_BFLOAT16 variables, DT_STRING entries (the object graph), partitioned (sliced) variables, a big-endian header."""
import os
import struct

import numpy as np
import pytest

TABLE_MAGIC = 0xdb4775248b80fb57
DT_FLOAT, DT_DOUBLE, DT_STRING, DT_BFLOAT16, DT_HALF = 1, 2, 7, 14, 19


@pytest.fixture(scope='module')
def kf():
    from dl_channel_estimation_mamimo_amd import keras_files
    return keras_files


# ------------------------------------------------------------------------------------------------ spec-level helpers
def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def crc32c_ref(data):
    """bitwise CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) - the slow textbook form"""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def mask_crc(crc):
    """leveldb util/crc32c.h Mask(): rotate right by 15 bits, add 0xa282ead8"""
    return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xFFFFFFFF


def pb_varint_field(fn, v):
    return varint(fn << 3 | 0) + varint(v)


def pb_bytes_field(fn, b):
    return varint(fn << 3 | 2) + varint(len(b)) + b


def pb_fixed32_field(fn, v):
    return varint(fn << 3 | 5) + struct.pack('<I', v)


def shape_proto(shape):
    return b''.join(pb_bytes_field(2, pb_varint_field(1, d)) for d in shape)       # TensorShapeProto.dim = 2, Dim.size = 1


def entry_proto(dtype, shape, shard, offset, size, crc, slices=False):
    e = pb_varint_field(1, dtype) + pb_bytes_field(2, shape_proto(shape))
    if shard:
        e += pb_varint_field(3, shard)
    if offset:
        e += pb_varint_field(4, offset)
    e += pb_varint_field(5, size) + pb_fixed32_field(6, mask_crc(crc))
    if slices:                                                                      # BundleEntryProto.slices = 7 (TensorSliceProto)
        e += pb_bytes_field(7, pb_bytes_field(1, pb_varint_field(1, 0) + pb_varint_field(2, 2)))
    return e


def header_proto(num_shards, endianness=0):
    h = pb_varint_field(1, num_shards)
    if endianness:
        h += pb_varint_field(2, endianness)
    return h + pb_bytes_field(3, pb_varint_field(1, 1))                              # VersionDef.producer = 1


def snappy_literal_only(data):
    """valid snappy stream: preamble + literals of at most 60 bytes (tag = (len - 1) << 2)"""
    out = bytearray(varint(len(data)))
    for i in range(0, len(data), 60):
        chunk = data[i:i + 60]
        out.append((len(chunk) - 1) << 2)
        out += chunk
    return bytes(out)


def table_block(entries, restart_interval=2):
    """LevelDB block: prefix-compressed entries, restart array, restart count"""
    out, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts) or 1)
    return bytes(out)


def write_table(path, entries, compress=False, entries_per_block=3):
    """entries sorted by key -> a LevelDB table file: data blocks, empty metaindex, index block, 48-byte footer"""
    buf = bytearray()

    def put_block(raw):
        ctype = 0
        if compress:
            raw, ctype = snappy_literal_only(raw), 1
        off = len(buf)
        buf.extend(raw)
        buf.append(ctype)
        buf.extend(struct.pack('<I', mask_crc(crc32c_ref(raw + bytes([ctype])))))
        return off, len(raw)

    index = []
    for i in range(0, len(entries), entries_per_block):
        chunk = entries[i:i + entries_per_block]
        off, size = put_block(table_block(chunk))
        index.append((chunk[-1][0], varint(off) + varint(size)))                     # separator key >= last key of the block
    moff, msize = put_block(table_block([]))
    ioff, isize = put_block(table_block(index, restart_interval=1))
    footer = varint(moff) + varint(msize) + varint(ioff) + varint(isize)
    buf.extend(footer.ljust(40, b'\0') + struct.pack('<Q', TABLE_MAGIC))
    with open(path, 'wb') as f:
        f.write(bytes(buf))


def write_bundle(prefix, tensors, num_shards=1, compress=False, endianness=0, sliced=None, extra_entries=()):
    """tensors: [(key, dtype code, array-as-stored)] -> prefix.index + prefix.data-*; shards assigned round-robin"""
    shards = [bytearray() for _ in range(num_shards)]
    entries = [(b'', header_proto(num_shards, endianness))]
    for i, (key, dtype, arr) in enumerate(sorted(tensors, key=lambda t: t[0].encode())):
        raw = arr.tobytes()
        sid = i % num_shards
        off = len(shards[sid])
        shards[sid] += raw + b'\0' * 3                                                # gaps between tensors are legal
        entries.append((key.encode(), entry_proto(dtype, arr.shape, sid, off, len(raw), crc32c_ref(raw), slices=(key == sliced))))
    entries += list(extra_entries)
    entries.sort(key=lambda kv: kv[0])
    write_table(prefix + '.index', entries, compress=compress)
    for sid, data in enumerate(shards):
        with open('%s.data-%05d-of-%05d' % (prefix, sid, num_shards), 'wb') as f:
            f.write(bytes(data))


# ------------------------------------------------------------------------------------------------ known answers first
def test_crc32c_and_mask_known_answers(kf):
    assert crc32c_ref(b'123456789') == 0xE3069283                      # the standard CRC-32C check value
    assert kf.crc32c(b'123456789') == 0xE3069283
    assert kf.crc32c(bytes(32)) == 0x8A9136AA                           # RFC 3720 B.4: 32 bytes of zeros
    assert kf.crc32c(bytes([0xff] * 32)) == 0x62A8AB43                  # ... 32 bytes of 0xff
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert kf._unmask_crc(mask_crc(c)) == c


def test_varint_and_snappy_known_streams(kf):
    assert kf._varint(b'\xac\x02', 0) == (300, 2) and varint(300) == b'\xac\x02'
    # format_description.txt: literal 'abcd' (tag 0x0c), then copy with 2-byte offset: length 12, offset 4 -> 16 bytes
    assert kf._snappy_decompress(bytes([16, 0x0c]) + b'abcd' + bytes([((12 - 1) << 2) | 2, 4, 0])) == b'abcd' * 4
    # copy with 1-byte offset (kind 1): length 4..11, 11-bit offset: len 7, offset 3 -> 'xyz' + 'xyzxyzx'
    assert kf._snappy_decompress(bytes([10, 0x08]) + b'xyz' + bytes([((7 - 4) << 2) | 1 | ((3 >> 8) << 5), 3])) == b'xyzxyzxyzx'
    # literal with an explicit 1-byte length (tag 60 << 2): 70 bytes
    blob = bytes(range(70))
    assert kf._snappy_decompress(bytes([70, 60 << 2, 69]) + blob) == blob
    with pytest.raises(kf.KerasFileError):
        kf._snappy_decompress(bytes([4, ((4 - 1) << 2) | 2, 9, 0]))     # copy that reaches in front of the output


# ------------------------------------------------------------------------------------------------ bundles
def _tensors(rng):
    f16 = rng.standard_normal((5, 3)).astype('<f2')
    f64 = rng.standard_normal((2, 2, 2)).astype('<f8')
    f32 = rng.standard_normal((7,)).astype('<f4')
    bf = (rng.standard_normal((4,)).astype('<f4').view('<u4') >> 16).astype('<u2')     # bfloat16 bit patterns
    return [('layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE', DT_HALF, f16),
            ('layer_with_weights-0/bias/.ATTRIBUTES/VARIABLE_VALUE', DT_DOUBLE, f64),
            ('layer_with_weights-1/gamma/.ATTRIBUTES/VARIABLE_VALUE', DT_FLOAT, f32),
            ('save_counter/.ATTRIBUTES/VARIABLE_VALUE', DT_BFLOAT16, bf)]


@pytest.mark.parametrize('num_shards,compress', [(1, False), (3, False), (2, True)])
def test_bundle_shards_compression_and_dtypes(kf, tmp_path, num_shards, compress):
    rng = np.random.default_rng(num_shards)
    ts = _tensors(rng)
    graph = ('_CHECKPOINTABLE_OBJECT_GRAPH'.encode(), entry_proto(DT_STRING, (), 0, 0, 0, 0))       # ignored: not a weight
    prefix = str(tmp_path / 'variables')
    write_bundle(prefix, ts, num_shards=num_shards, compress=compress, extra_entries=[graph])
    got = kf.read_tensor_bundle(prefix)                                   # CRC verification is the default
    assert sorted(got) == sorted(k for k, _, _ in ts)
    for key, dtype, arr in ts:
        if dtype == DT_BFLOAT16:
            want = (arr.astype(np.uint32) << 16).view(np.float32)
            assert got[key].dtype == np.float32 and np.array_equal(got[key], want)
        else:
            assert got[key].dtype == arr.dtype and got[key].shape == arr.shape and np.array_equal(got[key], arr)
    # a flipped payload byte is caught by the per-tensor checksum
    shard0 = '%s.data-%05d-of-%05d' % (prefix, 0, num_shards)
    raw = bytearray(open(shard0, 'rb').read())
    raw[1] ^= 0x40
    open(shard0, 'wb').write(bytes(raw))
    with pytest.raises(kf.KerasFileError, match='checksum'):
        kf.read_tensor_bundle(prefix)
    assert kf.read_tensor_bundle(prefix, verify_crc=False)               # ... and only by it


def test_bundle_rejections(kf, tmp_path):
    rng = np.random.default_rng(0)
    ts = _tensors(rng)
    p1 = str(tmp_path / 'big')
    write_bundle(p1, ts, endianness=1)
    with pytest.raises(kf.KerasFileError, match='big-endian'):
        kf.read_tensor_bundle(p1)
    p2 = str(tmp_path / 'sliced')
    write_bundle(p2, ts, sliced=ts[2][0])
    with pytest.raises(kf.KerasFileError, match='slices'):
        kf.read_tensor_bundle(p2)
    p3 = str(tmp_path / 'lost')
    write_bundle(p3, ts, num_shards=2)
    os.remove(p3 + '.data-00001-of-00002')
    with pytest.raises(kf.KerasFileError, match='missing'):
        kf.read_tensor_bundle(p3)
    p4 = str(tmp_path / 'magic')
    write_bundle(p4, ts)
    raw = bytearray(open(p4 + '.index', 'rb').read())
    raw[-1] ^= 0xff
    open(p4 + '.index', 'wb').write(bytes(raw))
    with pytest.raises(kf.KerasFileError, match='magic'):
        kf.read_tensor_bundle(p4)


def test_savedmodel_layer_mapping_from_an_independent_writer(kf, tmp_path):
    """Object-based checkpoint keys of a Dense - BatchNormalization - Dense model (layer_with_weights-<i>/<attr>) -> the keras
    paths load-by-topology uses; two shards, compressed index."""
    rng = np.random.default_rng(5)
    f = lambda *s: rng.standard_normal(s).astype('<f4')
    layers = [{'kernel': f(6, 4), 'bias': f(4)}, {'gamma': f(4), 'beta': f(4), 'moving_mean': f(4), 'moving_variance': np.abs(f(4)) + 0.1},
              {'kernel': f(4, 3), 'bias': f(3)}]
    ts = [('layer_with_weights-%d/%s/.ATTRIBUTES/VARIABLE_VALUE' % (i, k), DT_FLOAT, v) for i, lw in enumerate(layers) for k, v in lw.items()]
    ts.append(('optimizer/iter/.ATTRIBUTES/VARIABLE_VALUE', 9, np.array(7, dtype='<i8')))                # DT_INT64, not a layer weight
    mdir = tmp_path / 'real_keras_model'
    (mdir / 'variables').mkdir(parents=True)
    write_bundle(str(mdir / 'variables' / 'variables'), ts, num_shards=2, compress=True)
    w = kf.read_savedmodel_variables(str(mdir))
    assert sorted(w) == sorted(['fc_dense0/kernel:0', 'fc_dense0/bias:0', 'batch_normalization/gamma:0', 'batch_normalization/beta:0',
                                'batch_normalization/moving_mean:0', 'batch_normalization/moving_variance:0',
                                'fc_regressor/kernel:0', 'fc_regressor/bias:0'])
    assert np.array_equal(w['fc_dense0/kernel:0'], layers[0]['kernel']) and np.array_equal(w['fc_regressor/bias:0'], layers[2]['bias'])
    assert np.array_equal(w['batch_normalization/moving_variance:0'], layers[1]['moving_variance'])
