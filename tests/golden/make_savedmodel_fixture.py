#!/usr/bin/env python3
"""Writes tests/golden/savedmodel_fixture/{real,imag}_keras_model/variables/variables.{index,data-00000-of-00001}
in the TensorBundle format of a TF-2 SavedModel (``CSI_predictor.save(<d>_keras_model)``,
massiveMIMO_CSI_prediction_DNN.py:411; read back by inference.py:15-16).

TensorFlow is NOT available here, so - unlike the HDF5 fixture, which the genuine libhdf5 writes - this file is
produced by this script's own writer from the published formats (tensorflow/core/util/tensor_bundle: one
``BundleHeaderProto`` under the empty key, one ``BundleEntryProto`` {dtype, shape, shard_id, offset, size, masked
crc32c} per tensor, keys sorted; tensorflow/core/lib/io/table = the LevelDB table format: prefix-compressed
entries, restart array, 5-byte block trailer, index block, 48-byte footer with magic 0xdb4775248b80fb57).  The
reader test is therefore a self-consistency test of keras_files.read_savedmodel_variables and says so; a real
SavedModel written by TensorFlow has not been through it (SURVEY 8 a-10 stays "partial" for this container).

Keys as object-based Keras checkpoints name them: ``layer_with_weights-<i>/<var>/.ATTRIBUTES/VARIABLE_VALUE`` for
the i-th layer that owns weights, plus what a compiled model drags along (``_CHECKPOINTABLE_OBJECT_GRAPH`` string
tensor, optimizer hyper-parameters, one optimizer slot) - the reader has to step over those."""
import os
import struct
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))


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


def pb_varint(field, v):
    return varint(field << 3) + varint(v)


def pb_bytes(field, b):
    return varint((field << 3) | 2) + varint(len(b)) + b


def mask_crc(c):
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


class TableBuilder:
    def __init__(self, block_size=512, restart_interval=16):
        self.out = bytearray()
        self.block_size, self.ri = block_size, restart_interval
        self.index = []
        self._reset()

    def _reset(self):
        self.buf, self.restarts, self.n, self.last = bytearray(), [0], 0, b''

    def _emit(self, contents):
        from dl_channel_estimation_mamimo_amd.keras_files import crc32c
        off = len(self.out)
        self.out += contents + b'\x00' + struct.pack('<I', mask_crc(crc32c(bytes(contents) + b'\x00')))
        return off, len(contents)

    def _finish_block(self):
        body = bytes(self.buf) + b''.join(struct.pack('<I', r) for r in self.restarts) + struct.pack('<I', len(self.restarts))
        handle = self._emit(body)
        self.index.append((self.last, handle))
        self._reset()

    def add(self, key, value):
        if self.n and self.n % self.ri == 0:
            self.restarts.append(len(self.buf))
            shared = 0
        else:
            shared = 0
            while shared < min(len(key), len(self.last)) and key[shared] == self.last[shared]:
                shared += 1
            if self.n == 0:
                shared = 0
        self.buf += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        self.last, self.n = key, self.n + 1
        if len(self.buf) >= self.block_size:
            self._finish_block()

    def finish(self):
        if self.n:
            self._finish_block()
        meta = self._emit(struct.pack('<II', 0, 1))                       # empty metaindex block
        ib = bytearray()
        for key, (off, size) in self.index:                               # one restart per entry keeps it simple
            ib += varint(0) + varint(len(key)) + varint(len(varint(off) + varint(size))) + key + varint(off) + varint(size)
        body = bytes(ib) + struct.pack('<I', 0) + struct.pack('<I', 1)
        idx = self._emit(body)
        foot = varint(meta[0]) + varint(meta[1]) + varint(idx[0]) + varint(idx[1])
        self.out += foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', 0xdb4775248b80fb57)
        return bytes(self.out)


DT = {np.dtype('float32'): 1, np.dtype('int64'): 9}


def write_bundle(prefix, tensors, object_graph=b'\n\x05dummy'):
    """tensors: {key: ndarray}; written in key order like BundleWriter (std::map)."""
    from dl_channel_estimation_mamimo_amd.keras_files import crc32c
    os.makedirs(os.path.dirname(prefix), exist_ok=True)
    entries, data = {}, bytearray()
    # the object graph: a scalar DT_STRING tensor = varint length, masked crc of the lengths, bytes
    items = dict(tensors)
    for key in sorted(list(items) + ['_CHECKPOINTABLE_OBJECT_GRAPH']):
        if key == '_CHECKPOINTABLE_OBJECT_GRAPH':
            lens = varint(len(object_graph))
            raw = lens + struct.pack('<I', mask_crc(crc32c(struct.pack('<Q', len(object_graph))))) + object_graph
            dtype, shape = 7, ()
        else:
            arr = np.ascontiguousarray(items[key])
            raw, dtype, shape = arr.tobytes(), DT[arr.dtype], arr.shape
        shape_pb = b''.join(pb_bytes(2, pb_varint(1, int(s))) for s in shape)
        e = pb_varint(1, dtype) + pb_bytes(2, shape_pb)
        if len(data):
            e += pb_varint(4, len(data))                                   # offset (proto3: zero is omitted)
        e += pb_varint(5, len(raw)) + varint((6 << 3) | 5) + struct.pack('<I', mask_crc(crc32c(raw)))
        entries[key.encode()] = e
        data += raw
    tb = TableBuilder()
    header = pb_varint(1, 1) + pb_bytes(3, pb_varint(1, 1))               # num_shards = 1, endianness LITTLE (0, omitted), version.producer = 1
    tb.add(b'', header)
    for key in sorted(entries):
        tb.add(key, entries[key])
    with open(prefix + '.index', 'wb') as f:
        f.write(tb.finish())
    with open(prefix + '.data-00000-of-00001', 'wb') as f:
        f.write(bytes(data))


def checkpoint_tensors(w):
    """container-named weights -> object-based checkpoint keys of the reference's model (5 layers own weights)"""
    sfx = '/.ATTRIBUTES/VARIABLE_VALUE'
    out, li, i = {}, 0, 0
    while f'fc_dense{i}.kernel' in w:
        out[f'layer_with_weights-{li}/kernel{sfx}'] = w[f'fc_dense{i}.kernel']
        out[f'layer_with_weights-{li}/bias{sfx}'] = w[f'fc_dense{i}.bias']
        li += 1
        for v in ('gamma', 'beta', 'moving_mean', 'moving_variance'):
            out[f'layer_with_weights-{li}/{v}{sfx}'] = w[f'bn{i}.{v}']
        li += 1
        i += 1
    out[f'layer_with_weights-{li}/kernel{sfx}'] = w['fc_regressor.kernel']
    out[f'layer_with_weights-{li}/bias{sfx}'] = w['fc_regressor.bias']
    # what model.compile(Adam) adds
    out['optimizer/iter' + sfx] = np.array(0, dtype=np.int64)
    for h, v in (('beta_1', 0.9), ('beta_2', 0.999), ('decay', 0.0), ('learning_rate', 1e-4)):
        out[f'optimizer/{h}{sfx}'] = np.array(v, dtype=np.float32)
    out[f'layer_with_weights-0/bias/.OPTIMIZER_SLOT/optimizer/m{sfx}'] = np.zeros_like(w['fc_dense0.bias'])
    return out


def main():
    exp = np.load(os.path.join(OUT, 'keras_weights_expected.npz'))       # the same tensors as the HDF5 fixture
    for d in ('real', 'imag'):
        w = {k[len(d) + 1:]: exp[k] for k in exp.files if k.startswith(d + '.')}
        prefix = os.path.join(OUT, 'savedmodel_fixture', f'{d}_keras_model', 'variables', 'variables')
        write_bundle(prefix, checkpoint_tensors(w))
        # a SavedModel directory also holds saved_model.pb (the graph); the reader does not need it
        open(os.path.join(OUT, 'savedmodel_fixture', f'{d}_keras_model', 'saved_model.pb'), 'wb').write(b'')
    print('wrote savedmodel_fixture/{real,imag}_keras_model/variables')


if __name__ == '__main__':
    main()
