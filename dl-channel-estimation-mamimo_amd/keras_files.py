"""Readers (and an HDF5 writer) for the model files the REFERENCE writes, with no TensorFlow / h5py dependency:

  * ``<d>_weights-improvement.hdf5`` - the Keras-2.3 HDF5 weight checkpoint of
    massiveMIMO_CSI_prediction_DNN.py:279-281,319 (``ModelCheckpoint`` / ``save_weights``), loaded there by
    ``load_weights`` (:334).  Layout (keras ``hdf5_format.save_weights_to_hdf5_group``): root attribute
    ``layer_names`` (fixed-length strings, model layer order), one group per layer with attribute
    ``weight_names`` and one contiguous little-endian dataset per weight at ``<layer>/<weight name>`` -
    weight names contain a '/', so the datasets sit one group deeper (``fc_dense0/fc_dense0/kernel:0``).
    Whole-model ``.h5`` files (``model.save('x.h5')``) carry the same tree under ``/model_weights``.
  * ``<d>_keras_model/`` - the TF SavedModel directory of :411 (``CSI_predictor.save``), read by
    inference.py:15-16: the tensors live in ``variables/variables.index`` (an SSTable of
    ``BundleEntryProto`` records) + ``variables/variables.data-?????-of-?????`` (raw little-endian bytes).

Both return ``{keras variable path: float32 ndarray}`` in model order; ``model.normalize_keras_names`` maps the
paths to the container's names (BatchNormalization layers by order, as load-by-topology does).

Only what those writers emit is understood (HDF5: superblock 0-3, version-1/2 object headers, symbol-table
groups and compact link messages, contiguous / compact datasets of fixed-point or IEEE float type,
fixed- and variable-length string attributes; TensorBundle: uncompressed or snappy-compressed index blocks,
DT_FLOAT / DT_HALF / DT_DOUBLE / integer tensors without slices).  Anything else raises ``KerasFileError`` naming
the construct - never a silent wrong answer."""
import os
import struct

import numpy as np


class KerasFileError(ValueError):
    pass


# =====================================================================================================
# HDF5 (the subset libhdf5 1.8 / 1.10 writes for h5py files without chunking or compression)
# =====================================================================================================
_H5_SIG = b'\x89HDF\r\n\x1a\n'
_UNDEF = 0xFFFFFFFFFFFFFFFF


class _H5Type:
    def __init__(self, cls, size, dtype=None, strpad=0, base=None, vlen_string=False):
        self.cls, self.size, self.dtype, self.strpad, self.base, self.vlen_string = cls, size, dtype, strpad, base, vlen_string


class Hdf5File:
    """Minimal read-only HDF5 object model: ``f[path]`` -> ``Hdf5Group`` / ``Hdf5Dataset``; ``.attrs`` dicts;
    ``group.keys()`` in stored (name) order."""

    def __init__(self, path):
        with open(path, 'rb') as fh:
            self.buf = fh.read()
        self.path = path
        base = 0
        while True:
            if self.buf[base:base + 8] == _H5_SIG:
                break
            base = 512 if base == 0 else base * 2
            if base + 8 > len(self.buf):
                raise KerasFileError(f'{path}: not an HDF5 file (no superblock signature)')
        ver = self.buf[base + 8]
        if ver in (0, 1):
            self.so, self.sl = self.buf[base + 13], self.buf[base + 14]
            p = base + 24 + (4 if ver == 1 else 0)
            self.base = self._off(p)
            p += 4 * self.so                                   # base, free-space, end-of-file, driver-info addresses
            # root group symbol-table entry: link name offset, object header address, cache type, reserved, scratch
            self.root_addr = self._off(p + self.so)
        elif ver in (2, 3):
            self.so, self.sl = self.buf[base + 9], self.buf[base + 10]
            p = base + 12
            self.base = self._off(p)
            self.root_addr = self._off(p + 3 * self.so)
        else:
            raise KerasFileError(f'{path}: HDF5 superblock version {ver} is not supported')
        if self.so not in (4, 8) or self.sl not in (4, 8):
            raise KerasFileError(f'{path}: HDF5 offset / length sizes {self.so}/{self.sl} are not supported')
        self.root = Hdf5Group(self, self.root_addr, '/')

    # ---- primitive readers
    def _u(self, p, n):
        return int.from_bytes(self.buf[p:p + n], 'little')

    def _off(self, p):
        return self._u(p, self.so)

    def _len(self, p):
        return self._u(p, self.sl)

    def __getitem__(self, path):
        node = self.root
        for part in [q for q in path.split('/') if q]:
            node = node[part]
        return node

    # ---- object headers -> list of (type, flags, payload offset, payload size)
    def messages(self, addr):
        a = addr + self.base
        buf = self.buf
        out = []
        if buf[a:a + 4] == b'OHDR':
            if buf[a + 4] != 2:
                raise KerasFileError(f'{self.path}: object header version {buf[a + 4]} at {addr:#x}')
            flags = buf[a + 5]
            p = a + 6
            if flags & 0x20:
                p += 16                                        # access / modification / change / birth times
            if flags & 0x10:
                p += 4                                         # max compact / min dense attributes
            szb = 1 << (flags & 3)
            chunk = self._u(p, szb)
            p += szb
            blocks = [(p, p + chunk)]
            while blocks:
                q, end = blocks.pop(0)
                while q + 4 <= end:
                    mtype, msize, mflags = buf[q], self._u(q + 1, 2), buf[q + 3]
                    q += 4 + (2 if flags & 0x04 else 0)
                    if q + msize > end:
                        break
                    if mtype == 0x10:
                        co, cl = self._off(q) + self.base, self._len(q + self.so)
                        if buf[co:co + 4] != b'OCHK':
                            raise KerasFileError(f'{self.path}: bad object header continuation at {co:#x}')
                        blocks.append((co + 4, co + cl - 4))
                    elif mtype != 0:
                        out.append((mtype, mflags, q, msize))
                    q += msize
            return out
        if buf[a] != 1:
            raise KerasFileError(f'{self.path}: object header version {buf[a]} at {addr:#x} is not supported')
        nmsg, hsize = self._u(a + 2, 2), self._u(a + 8, 4)
        blocks = [(a + 16, a + 16 + hsize)]
        while blocks and len(out) < nmsg + 64:
            q, end = blocks.pop(0)
            while q + 8 <= end:
                mtype, msize, mflags = self._u(q, 2), self._u(q + 2, 2), buf[q + 4]
                q += 8
                if mtype == 0x10:
                    blocks.append((self._off(q) + self.base, self._off(q) + self.base + self._len(q + self.so)))
                elif mtype != 0:
                    out.append((mtype, mflags, q, msize))
                q += msize
        return out

    # ---- datatype / dataspace messages
    def parse_type(self, p):
        buf = self.buf
        cls, ver = buf[p] & 0x0f, buf[p] >> 4
        b0, b1 = buf[p + 1], buf[p + 2]
        size = self._u(p + 4, 4)
        if cls == 0:                                           # fixed point
            dt = np.dtype(('>' if b0 & 1 else '<') + ('i' if b0 & 8 else 'u') + str(size))
            return _H5Type(cls, size, dt)
        if cls == 1:                                           # IEEE float (libhdf5 native float types only)
            if size not in (2, 4, 8):
                raise KerasFileError(f'{self.path}: {size}-byte floating-point type')
            exp_size, mant_size = buf[p + 13], buf[p + 15]
            if (size, exp_size, mant_size) not in ((2, 5, 10), (4, 8, 23), (8, 11, 52)):
                raise KerasFileError(f'{self.path}: non-IEEE float layout (exponent {exp_size}, mantissa {mant_size} bits)')
            return _H5Type(cls, size, np.dtype(('>' if b0 & 1 else '<') + 'f' + str(size)))
        if cls == 3:                                           # fixed-length string
            return _H5Type(cls, size, np.dtype('S' + str(size)), strpad=b0 & 0x0f)
        if cls == 9:                                           # variable length: (type 1 = string) of a base type
            base = self.parse_type(p + 8)
            return _H5Type(cls, size, None, base=base, vlen_string=(b0 & 0x0f) == 1)
        raise KerasFileError(f'{self.path}: HDF5 datatype class {cls} is not supported')

    def parse_space(self, p):
        buf = self.buf
        ver, rank, flags = buf[p], buf[p + 1], buf[p + 2]
        if ver == 1:
            q = p + 8
        elif ver == 2:
            if buf[p + 3] == 2:
                return None                                    # null dataspace
            q = p + 4
        else:
            raise KerasFileError(f'{self.path}: dataspace message version {ver}')
        return tuple(self._len(q + i * self.sl) for i in range(rank))

    def global_heap_object(self, addr, index):
        a = addr + self.base
        if self.buf[a:a + 4] != b'GCOL':
            raise KerasFileError(f'{self.path}: bad global heap collection at {addr:#x}')
        end = a + self._len(a + 8)
        q = a + 8 + self.sl
        while q + 8 + self.sl <= end:
            idx, size = self._u(q, 2), self._len(q + 8)
            if idx == 0:
                break
            if idx == index:
                return self.buf[q + 8 + self.sl:q + 8 + self.sl + size]
            q += 8 + self.sl + ((size + 7) & ~7)
        raise KerasFileError(f'{self.path}: global heap object {index} not found at {addr:#x}')

    def decode(self, typ, shape, raw):
        """raw bytes of `shape` elements of `typ` -> ndarray / list of bytes"""
        n = int(np.prod(shape)) if shape else 1
        if typ.cls == 9:
            if not typ.vlen_string:
                raise KerasFileError(f'{self.path}: variable-length sequences are not supported')
            vals, step = [], 4 + self.so + 4
            for i in range(n):
                q = i * step
                length = int.from_bytes(raw[q:q + 4], 'little')
                addr = int.from_bytes(raw[q + 4:q + 4 + self.so], 'little')
                idx = int.from_bytes(raw[q + 4 + self.so:q + 8 + self.so], 'little')
                vals.append(self.global_heap_object(addr, idx)[:length] if length else b'')
            return np.array(vals, dtype=object).reshape(shape) if shape else vals[0]
        arr = np.frombuffer(raw, dtype=typ.dtype, count=n)
        if typ.cls == 3 and typ.strpad == 2:
            # numpy 'S' drops trailing NULs itself (null-terminated / null-padded strings); space padding is stripped here
            arr = np.array([bytes(v).rstrip(b' ') for v in arr], dtype=typ.dtype)
        return arr.reshape(shape) if shape else arr.reshape(())

    def attributes(self, addr):
        out = {}
        for mtype, _, p, size in self.messages(addr):
            if mtype != 0x0c:
                continue
            buf = self.buf
            ver = buf[p]
            nsz, tsz, ssz = self._u(p + 2, 2), self._u(p + 4, 2), self._u(p + 6, 2)
            q = p + 8 + (1 if ver == 3 else 0)
            pad = (lambda x: (x + 7) & ~7) if ver == 1 else (lambda x: x)
            if ver not in (1, 2, 3):
                raise KerasFileError(f'{self.path}: attribute message version {ver}')
            name = buf[q:q + nsz].split(b'\0')[0].decode('utf8')
            q += pad(nsz)
            if ver >= 2 and buf[p + 1] & 3:
                raise KerasFileError(f'{self.path}: attribute {name!r} uses a shared datatype / dataspace')
            typ = self.parse_type(q)
            q += pad(tsz)
            shape = self.parse_space(q)
            q += pad(ssz)
            if shape is None:
                out[name] = None
                continue
            n = int(np.prod(shape)) if shape else 1
            out[name] = self.decode(typ, shape, buf[q:q + n * typ.size])
        return out


class Hdf5Dataset:
    def __init__(self, f, addr, name):
        self.file, self.addr, self.name = f, addr, name
        typ = shape = layout = None
        for mtype, _, p, size in f.messages(addr):
            if mtype == 0x03:
                typ = f.parse_type(p)
            elif mtype == 0x01:
                shape = f.parse_space(p)
            elif mtype == 0x08:
                layout = (p, size)
            elif mtype == 0x0b:
                raise KerasFileError(f'{f.path}: dataset {name} has a filter pipeline (compression) - not what keras save_weights writes')
        if typ is None or shape is None or layout is None:
            raise KerasFileError(f'{f.path}: {name} is not a dataset (no datatype / dataspace / layout message)')
        self.type, self.shape, self._layout = typ, shape, layout

    @property
    def attrs(self):
        return self.file.attributes(self.addr)

    def read(self):
        f = self.file
        p, _ = self._layout
        buf = f.buf
        ver = buf[p]
        nbytes = (int(np.prod(self.shape)) if self.shape else 1) * self.type.size
        if ver in (3, 4):
            cls = buf[p + 1]
            if cls == 1:                                       # contiguous
                addr = f._off(p + 2)
                if addr == _UNDEF & ((1 << (8 * f.so)) - 1):
                    return np.zeros(self.shape, self.type.dtype)        # never written: fill value 0
                raw = buf[addr + f.base:addr + f.base + nbytes]
            elif cls == 0:                                     # compact
                raw = buf[p + 4:p + 4 + f._u(p + 2, 2)][:nbytes]
            else:
                raise KerasFileError(f'{f.path}: dataset {self.name} is chunked - not what keras save_weights writes')
        elif ver in (1, 2):
            rank, cls = buf[p + 1], buf[p + 2]
            if cls == 1:
                addr = f._off(p + 8)
                raw = buf[addr + f.base:addr + f.base + nbytes]
            elif cls == 0:
                q = p + 8 + 4 * rank
                raw = buf[q + 4:q + 4 + f._u(q, 4)][:nbytes]
            else:
                raise KerasFileError(f'{f.path}: dataset {self.name} is chunked - not what keras save_weights writes')
        else:
            raise KerasFileError(f'{f.path}: data layout message version {ver}')
        if len(raw) != nbytes:
            raise KerasFileError(f'{f.path}: dataset {self.name} is truncated ({len(raw)} of {nbytes} bytes)')
        return f.decode(self.type, self.shape, raw)


class Hdf5Group:
    def __init__(self, f, addr, name):
        self.file, self.addr, self.name = f, addr, name
        self._links = None

    @property
    def attrs(self):
        return self.file.attributes(self.addr)

    def _load(self):
        if self._links is not None:
            return
        f, links = self.file, {}
        for mtype, _, p, size in f.messages(self.addr):
            if mtype == 0x11:                                  # symbol table: B-tree v1 of symbol nodes + local heap of names
                btree, heap = f._off(p), f._off(p + f.so)
                h = heap + f.base
                if f.buf[h:h + 4] != b'HEAP':
                    raise KerasFileError(f'{f.path}: bad local heap at {heap:#x}')
                data = f._off(h + 8 + 2 * f.sl) + f.base
                self._walk_btree(btree, data, links)
            elif mtype == 0x06:                                # link message (new-style compact group)
                buf = f.buf
                flags = buf[p + 1]
                q = p + 2
                ltype = 0
                if flags & 0x08:
                    ltype = buf[q]
                    q += 1
                if flags & 0x04:
                    q += 8
                if flags & 0x10:
                    q += 1
                lsz = 1 << (flags & 3)
                nlen = f._u(q, lsz)
                q += lsz
                nm = buf[q:q + nlen].decode('utf8')
                q += nlen
                if ltype == 0:
                    links[nm] = f._off(q)
            elif mtype == 0x02:
                # link info: dense storage when the fractal-heap address is defined
                buf = f.buf
                q = p + 2 + (8 if buf[p + 1] & 1 else 0)
                if f._off(q) != _UNDEF & ((1 << (8 * f.so)) - 1):
                    raise KerasFileError(f'{f.path}: group {self.name} uses dense link storage (libver="latest" with > 8 links) - not supported')
        self._links = links

    def _walk_btree(self, addr, heap_data, links):
        f = self.file
        a = addr + f.base
        buf = f.buf
        if buf[a:a + 4] == b'SNOD':
            n = f._u(a + 6, 2)
            q = a + 8
            esz = 2 * f.so + 24
            for i in range(n):
                noff, oaddr = f._off(q), f._off(q + f.so)
                s = heap_data + noff
                links[buf[s:buf.index(b'\0', s)].decode('utf8')] = oaddr
                q += esz
            return
        if buf[a:a + 4] != b'TREE' or buf[a + 4] != 0:
            raise KerasFileError(f'{f.path}: bad group B-tree node at {addr:#x}')
        n = f._u(a + 6, 2)
        q = a + 8 + 2 * f.so + f.sl                           # skip siblings and key 0
        for i in range(n):
            self._walk_btree(f._off(q), heap_data, links)
            q += f.so + f.sl

    def keys(self):
        self._load()
        return list(self._links)

    def __contains__(self, k):
        self._load()
        return k in self._links

    def __getitem__(self, k):
        self._load()
        if k not in self._links:
            raise KeyError(f'{self.name}: no member {k!r}')
        addr = self._links[k]
        full = self.name.rstrip('/') + '/' + k
        kinds = {m[0] for m in self.file.messages(addr)}
        return Hdf5Dataset(self.file, addr, full) if 0x08 in kinds else Hdf5Group(self.file, addr, full)


def _as_names(v):
    if v is None:
        return []
    return [bytes(x).decode('utf8') if not isinstance(x, str) else x for x in np.asarray(v).reshape(-1).tolist()]


def read_keras_hdf5_weights(path):
    """{'<layer>/<weight name>': float32 ndarray} of a Keras HDF5 weights / whole-model file, in the order keras
    itself walks them when loading by topology: root attribute ``layer_names``, per layer ``weight_names``
    (attributes split into ``name0, name1, ...`` chunks by keras for very long lists are re-joined)."""
    f = Hdf5File(path)
    root = f.root
    if 'layer_names' not in root.attrs and 'model_weights' in root:
        root = root['model_weights']                                            # model.save('x.h5')

    def listed(attrs, key):
        if key in attrs:
            return _as_names(attrs[key])
        out, i = [], 0
        while f'{key}{i}' in attrs:
            out += _as_names(attrs[f'{key}{i}'])
            i += 1
        if i == 0:
            raise KerasFileError(f'{path}: no {key!r} attribute - not a keras weights file')
        return out

    out = {}
    for layer in listed(root.attrs, 'layer_names'):
        g = root[layer]
        for wname in listed(g.attrs, 'weight_names'):
            node = g
            for part in wname.split('/'):
                node = node[part]
            if not isinstance(node, Hdf5Dataset):
                raise KerasFileError(f'{path}: {layer}/{wname} is not a dataset')
            out[f'{layer}/{wname}' if not wname.startswith(layer + '/') else wname] = np.asarray(node.read(), dtype=np.float32)
    return out


# =====================================================================================================
# TF SavedModel variables (TensorBundle): variables.index (SSTable) + variables.data-*
# =====================================================================================================
_TABLE_MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto: DT_FLOAT 1, DT_DOUBLE 2, DT_INT32 3, DT_UINT8 4, DT_INT16 5, DT_INT8 6, DT_INT64 9, DT_BOOL 10,
# DT_BFLOAT16 14 (widened to float32 below), DT_UINT16 17, DT_HALF 19, DT_UINT32 22, DT_UINT64 23
_DT = {1: '<f4', 2: '<f8', 3: '<i4', 4: 'u1', 5: '<i2', 6: 'i1', 9: '<i8', 10: '?', 14: '<u2', 17: '<u2', 19: '<f2', 22: '<u4', 23: '<u8'}


def _varint(b, p):
    v = shift = 0
    while True:
        c = b[p]
        p += 1
        v |= (c & 0x7f) << shift
        if c < 0x80:
            return v, p
        shift += 7


def _snappy_decompress(src):
    n, p = _varint(src, 0)
    out = bytearray()
    while p < len(src):
        tag = src[p]
        p += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[p:p + nb], 'little')
                p += nb
            ln += 1
            out += src[p:p + ln]
            p += ln
            continue
        if kind == 1:
            ln, off = ((tag >> 2) & 7) + 4, ((tag >> 5) << 8) | src[p]
            p += 1
        elif kind == 2:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[p:p + 2], 'little')
            p += 2
        else:
            ln, off = (tag >> 2) + 1, int.from_bytes(src[p:p + 4], 'little')
            p += 4
        if off == 0 or off > len(out):
            raise KerasFileError('corrupt snappy block in variables.index')
        for _ in range(ln):
            out.append(out[-off])
    if len(out) != n:
        raise KerasFileError('corrupt snappy block in variables.index (length)')
    return bytes(out)


def _table_block(buf, off, size):
    """contents of the block at (off, size): 1-byte compression type + 4-byte crc follow it"""
    raw, ctype = buf[off:off + size], buf[off + size]
    if ctype == 1:
        raw = _snappy_decompress(raw)
    elif ctype != 0:
        raise KerasFileError(f'variables.index: block compression type {ctype}')
    return raw


def _block_entries(block):
    """(key, value) pairs of one SSTable block (prefix-compressed keys, restart array at the end)"""
    nrestart = struct.unpack('<I', block[-4:])[0]
    end = len(block) - 4 - 4 * nrestart
    p, key = 0, b''
    while p < end:
        shared, p = _varint(block, p)
        non_shared, p = _varint(block, p)
        vlen, p = _varint(block, p)
        key = key[:shared] + block[p:p + non_shared]
        p += non_shared
        yield key, block[p:p + vlen]
        p += vlen


def _proto_fields(b):
    """flat protobuf wire parse -> list of (field number, wire type, value)"""
    p, out = 0, []
    while p < len(b):
        tag, p = _varint(b, p)
        fn, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = _varint(b, p)
        elif wt == 1:
            v, p = b[p:p + 8], p + 8
        elif wt == 2:
            ln, p = _varint(b, p)
            v, p = b[p:p + ln], p + ln
        elif wt == 5:
            v, p = b[p:p + 4], p + 4
        else:
            raise KerasFileError(f'variables.index: protobuf wire type {wt}')
        out.append((fn, wt, v))
    return out


_CRC32C_TABLE = None


_CRC32C_NATIVE = None


def _crc32c_native():
    """csi_crc32c of the HIP library (host code: the SSE4.2 crc32 instruction, ~10 GB/s) when the library is built; the pure-Python
    table walk below (~10 MB/s: 30 s for the two models of a Nt = 32 SavedModel) stays as the fallback and as its cross-check."""
    global _CRC32C_NATIVE
    if _CRC32C_NATIVE is None:
        try:
            import ctypes
            from . import _lib
            fn = _lib.load_library().csi_crc32c
            _CRC32C_NATIVE = lambda buf, crc: int(fn((ctypes.c_char * len(buf)).from_buffer_copy(buf) if not isinstance(buf, bytes) else buf, len(buf), crc)) & 0xFFFFFFFF
        except Exception:
            _CRC32C_NATIVE = False
    return _CRC32C_NATIVE


def crc32c(data, crc=0, native=True):
    """CRC-32C (Castagnoli), the checksum TensorBundle stores (masked) per tensor."""
    global _CRC32C_TABLE
    if native and _crc32c_native():
        buf = data if isinstance(data, bytes) else bytes(memoryview(data).cast('B'))
        return _crc32c_native()(buf, int(crc))
    if _CRC32C_TABLE is None:
        t = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
        _CRC32C_TABLE = t
    t = _CRC32C_TABLE
    crc = np.uint32(crc ^ 0xFFFFFFFF)
    # byte-serial table walk, vectorised 4 Ki bytes at a time is not possible (serial dependency): plain loop on a
    # python int is fast enough for the few MB checked here
    c = int(crc)
    tab = t.tolist()
    for byte in memoryview(data).cast('B') if not isinstance(data, (bytes, bytearray)) else data:
        c = tab[(c ^ byte) & 0xff] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _unmask_crc(masked):
    rot = (masked - 0xa282ead8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


def read_tensor_bundle(prefix, verify_crc=True):
    """{key: ndarray} of the tensor bundle ``<prefix>.index`` + ``<prefix>.data-xxxxx-of-yyyyy`` in key order
    (keys of object-based checkpoints look like ``layer_with_weights-0/kernel/.ATTRIBUTES/VARIABLE_VALUE``)."""
    with open(prefix + '.index', 'rb') as fh:
        buf = fh.read()
    if len(buf) < 48 or struct.unpack('<Q', buf[-8:])[0] != _TABLE_MAGIC:
        raise KerasFileError(f'{prefix}.index: not a TensorBundle index (table magic)')
    foot = buf[-48:]
    _, p = _varint(foot, 0)                                    # metaindex handle: offset, size
    _, p = _varint(foot, p)
    ioff, p = _varint(foot, p)
    isize, p = _varint(foot, p)
    entries = []
    for _, handle in _block_entries(_table_block(buf, ioff, isize)):
        boff, q = _varint(handle, 0)
        bsize, q = _varint(handle, q)
        entries += list(_block_entries(_table_block(buf, boff, bsize)))
    num_shards, shards = 1, {}
    out = {}
    for key, val in entries:
        if key == b'':                                         # BundleHeaderProto: num_shards = 1, endianness = 2
            for fn, wt, v in _proto_fields(val):
                if fn == 1:
                    num_shards = v
                if fn == 2 and v != 0:                         # BundleHeaderProto.Endianness: LITTLE = 0, BIG = 1
                    raise KerasFileError(f'{prefix}.index: big-endian bundle')
            continue
        dtype = shard = offset = size = 0
        shape, crc, sliced = [], None, False
        for fn, wt, v in _proto_fields(val):
            if fn == 1:
                dtype = v
            elif fn == 2:
                for f2, _, v2 in _proto_fields(v):             # TensorShapeProto.dim
                    if f2 == 2:
                        dim = [x for f3, _, x in _proto_fields(v2) if f3 == 1]
                        shape.append(dim[0] if dim else 0)
            elif fn == 3:
                shard = v
            elif fn == 4:
                offset = v
            elif fn == 5:
                size = v
            elif fn == 6:
                crc = struct.unpack('<I', v)[0]
            elif fn == 7:
                sliced = True
        name = key.decode('utf8')
        if dtype == 7 or dtype == 21 or dtype not in _DT:      # DT_STRING (the object graph) / DT_VARIANT / others: not weights
            continue
        if sliced:
            raise KerasFileError(f'{prefix}.index: tensor {name} is stored in slices (partitioned variable)')
        if shard not in shards:
            if not 0 <= shard < max(num_shards, 1):
                raise KerasFileError(f'{prefix}.index: tensor {name} sits in shard {shard} of {num_shards}')
            fn_ = '%s.data-%05d-of-%05d' % (prefix, shard, num_shards)
            if not os.path.exists(fn_):
                raise KerasFileError(f'{fn_}: shard file of tensor {name} is missing')
            with open(fn_, 'rb') as fh:
                shards[shard] = fh.read()
        raw = shards[shard][offset:offset + size]
        dt = np.dtype(_DT[dtype])
        n = int(np.prod(shape)) if shape else 1
        if len(raw) != size or n * dt.itemsize != size:
            raise KerasFileError(f'{prefix}: tensor {name} has {size} bytes for shape {shape} of {dt}')
        if verify_crc and crc is not None and crc32c(raw) != _unmask_crc(crc):
            raise KerasFileError(f'{prefix}: checksum mismatch in tensor {name}')
        arr = np.frombuffer(raw, dtype=dt).reshape(shape).copy()
        if dtype == 14:                                          # DT_BFLOAT16: the upper half of a float32
            arr = (arr.astype(np.uint32) << 16).view(np.float32)
        out[name] = arr
    return out


_BN_VARS = ('gamma', 'beta', 'moving_mean', 'moving_variance')


def read_savedmodel_variables(model_dir, verify_crc=True):
    """Weights of a Keras model saved as a TF SavedModel directory (DNN.py:411), as keras-style paths in model
    order.  The object-based checkpoint names the tensors by position - ``layer_with_weights-<i>/<attribute>`` with i
    counting the layers that own weights, in model order - which is all load-by-topology needs: a layer with
    kernel + bias is a Dense layer, one with gamma / beta / moving_mean / moving_variance a BatchNormalization;
    the Dense layers are fc_dense0.. in order and the last one is fc_regressor (DNN.py:211-227)."""
    prefix = os.path.join(model_dir, 'variables', 'variables')
    if not os.path.exists(prefix + '.index'):
        raise KerasFileError(f'{model_dir}: no variables/variables.index - not a SavedModel directory')
    tensors = read_tensor_bundle(prefix, verify_crc=verify_crc)
    layers = {}
    for key, val in tensors.items():
        parts = key.split('/')
        if len(parts) >= 4 and parts[0].startswith('layer_with_weights-') and parts[2:4] == ['.ATTRIBUTES', 'VARIABLE_VALUE']:
            layers.setdefault(int(parts[0].split('-')[1]), {})[parts[1]] = val
    if not layers:
        raise KerasFileError(f'{model_dir}: no layer_with_weights-* variables in the checkpoint')
    dense = [i for i in sorted(layers) if 'kernel' in layers[i]]
    out, n_dense, n_bn = {}, 0, 0
    for i in sorted(layers):
        lw = layers[i]
        if 'kernel' in lw:
            name = 'fc_regressor' if i == dense[-1] else f'fc_dense{n_dense}'
            n_dense += 1
            out[f'{name}/kernel:0'] = np.asarray(lw['kernel'], np.float32)
            if 'bias' in lw:
                out[f'{name}/bias:0'] = np.asarray(lw['bias'], np.float32)
        elif all(v in lw for v in _BN_VARS):
            name = 'batch_normalization' + (f'_{n_bn}' if n_bn else '')
            n_bn += 1
            for v in _BN_VARS:
                out[f'{name}/{v}:0'] = np.asarray(lw[v], np.float32)
        else:
            raise KerasFileError(f'{model_dir}: layer_with_weights-{i} has variables {sorted(lw)} - neither Dense nor BatchNormalization')
    return out


# =====================================================================================================
# Writing a Keras HDF5 weights file (what DNN.py:319 ``CSI_predictor.save_weights(<d>_weights-improvement.hdf5)``
# leaves behind), so that weights trained on the MI355X go back into the reference's own pipeline - keras
# ``load_weights`` (DNN.py:334) and the MATLAB evaluation behind it.  Same subset of the format as the reader, the
# objects libhdf5 1.10 itself writes for such a file (superblock 0, version-1 object headers, symbol-table groups:
# B-tree node + symbol nodes + local heap, contiguous little-endian datasets, fixed-length string attributes); the
# byte patterns of the datatype / dataspace / fill-value messages are the library's own (checked against
# tests/golden/keras_weights_real.hdf5, which libhdf5 wrote).  tests/test_host.py re-opens the result with the genuine
# libhdf5 where one is installed.
# =====================================================================================================
_LEAF_K, _INTERNAL_K = 4, 16
_UNDEF8 = b'\xff' * 8


def _pad8(b):
    return b + b'\0' * (-len(b) % 8)


def _h5_msg(mtype, data, flags=0):
    data = _pad8(data)
    return struct.pack('<HHB3x', mtype, len(data), flags) + data


def _h5_dataspace(shape):
    if shape == ():
        return struct.pack('<BBBB4x', 1, 0, 0, 0)
    dims = b''.join(struct.pack('<Q', int(d)) for d in shape)
    return struct.pack('<BBBB4x', 1, len(shape), 1, 0) + dims + dims            # flags 1: maximum dimensions = dimensions


_H5_F32LE = bytes.fromhex('11201f00040000000000200017080017' '7f000000')
_H5_F64LE = bytes.fromhex('11203f0008000000000040003' '40b0034ff030000')


def _h5_string_type(n):
    return struct.pack('<BBBBI', 0x13, 0x01, 0, 0, max(int(n), 1))            # class 3 v1, null-padded ASCII


def _h5_attr(name, dtype, space, payload):
    nm = name.encode() + b'\0'
    return _h5_msg(0x0c, struct.pack('<BBHHH', 1, 0, len(nm), len(dtype), len(space)) + _pad8(nm) + _pad8(dtype) + _pad8(space) + payload)


def _h5_attr_strings(name, items):
    """list of str -> what h5py stores for a list of bytes: a 1-D array of fixed-length strings; an empty list is
    stored as float64 of shape (0,) (numpy's dtype for an empty list) - as keras / h5py do for layers without weights"""
    if not items:
        return _h5_attr(name, _H5_F64LE, _h5_dataspace((0,)), b'')
    raw = [s.encode('utf8') for s in items]
    n = max(len(r) for r in raw)
    return _h5_attr(name, _h5_string_type(n), _h5_dataspace((len(raw),)), b''.join(r.ljust(n, b'\0') for r in raw))


def _h5_attr_scalar_string(name, value):
    raw = value.encode('utf8')
    return _h5_attr(name, _h5_string_type(len(raw)), _h5_dataspace(()), raw)


def _h5_object_header(messages):
    body = b''.join(messages)
    return struct.pack('<BBHII4x', 1, 0, len(messages), 1, len(body)) + body


class _H5Writer:
    def __init__(self):
        self.buf = bytearray(96)                    # superblock, filled in last

    def alloc(self, data):
        off = len(self.buf)
        self.buf += _pad8(bytes(data))
        return off

    def dataset(self, arr):
        arr = np.ascontiguousarray(arr, dtype='<f4')
        data = self.alloc(arr.tobytes() if arr.size else b'\0' * 8)
        msgs = [_h5_msg(0x01, _h5_dataspace(arr.shape)), _h5_msg(0x03, _H5_F32LE, flags=1),
                _h5_msg(0x05, bytes([2, 2, 2, 1, 0, 0, 0, 0]), flags=1),                  # fill value: late allocation, written if set, default
                _h5_msg(0x08, struct.pack('<BBQQ', 3, 1, data, arr.nbytes))]
        return self.alloc(_h5_object_header(msgs)), None

    def group(self, members, attr_msgs):
        """members: {name: (object header address, (btree, heap) or None)} -> (object header address, (btree, heap))"""
        names = sorted(members, key=lambda s: s.encode('utf8'))
        # local heap: offset 0 = the empty name, then the names; no free block (free-list head = 1 = H5HL_FREE_NULL)
        seg, off = bytearray(8), {}
        for nme in names:
            off[nme] = len(seg)
            seg += _pad8(nme.encode('utf8') + b'\0')
        seg_addr = self.alloc(seg)
        heap = self.alloc(b'HEAP' + bytes(4) + struct.pack('<QQQ', len(seg), 1, seg_addr))
        # symbol nodes of <= 2 * leaf K entries, sorted by name
        per = 2 * _LEAF_K
        chunks = [names[i:i + per] for i in range(0, len(names), per)] or [[]]
        if len(chunks) > 2 * _INTERNAL_K:
            raise KerasFileError('more than %d members in one group' % (per * 2 * _INTERNAL_K))
        snods = []
        for ch in chunks:
            ent = b''
            for nme in ch:
                addr, stab = members[nme]
                if stab:
                    ent += struct.pack('<QQII', off[nme], addr, 1, 0) + struct.pack('<QQ', *stab)
                else:
                    ent += struct.pack('<QQII', off[nme], addr, 0, 0) + bytes(16)
            snods.append(self.alloc(b'SNOD' + struct.pack('<BBH', 1, 0, len(ch)) + ent.ljust(per * 40, b'\0')))
        # one B-tree node (level 0): key 0 = empty name, key i+1 = largest name of child i
        body = struct.pack('<Q', 0)
        for ch, sn in zip(chunks, snods):
            body += struct.pack('<QQ', sn, off[ch[-1]] if ch else 0)
        full = 8 * (2 * _INTERNAL_K + 1) + 8 * 2 * _INTERNAL_K
        btree = self.alloc(b'TREE' + struct.pack('<BBH', 0, 0, len(chunks) if names else 0) + _UNDEF8 + _UNDEF8 + body.ljust(full, b'\0'))
        hdr = self.alloc(_h5_object_header([_h5_msg(0x11, struct.pack('<QQ', btree, heap))] + attr_msgs))
        return hdr, (btree, heap)

    def finish(self, root):
        hdr, (btree, heap) = root
        sb = (_H5_SIG + bytes([0, 0, 0, 0, 0, 8, 8, 0]) + struct.pack('<HHI', _LEAF_K, _INTERNAL_K, 0) +
              struct.pack('<Q', 0) + _UNDEF8 + struct.pack('<Q', len(self.buf)) + _UNDEF8 +
              struct.pack('<QQII', 0, hdr, 1, 0) + struct.pack('<QQ', btree, heap))
        assert len(sb) == 96
        self.buf[0:96] = sb
        return bytes(self.buf)


def write_keras_hdf5_weights(path, layers, keras_version='2.4.0', backend='tensorflow'):
    """``layers``: [(layer name, [(weight name like 'fc_dense0/kernel:0', float32 array), ...]), ...] in model.layers
    order (layers without weights carry an empty list) -> a file keras' ``load_weights`` reads by topology."""
    w = _H5Writer()
    top = {}
    for lname, weights in layers:
        members = {}
        if weights:
            # weight names contain the layer name and a '/': the datasets sit in an inner group of that name
            inner = {}
            for wname, arr in weights:
                head, _, leaf = wname.partition('/')
                if head != lname or not leaf or '/' in leaf:
                    raise KerasFileError(f'weight name {wname!r} is not "<layer>/<variable>" of layer {lname!r}')
                inner[leaf] = w.dataset(arr)
            members[lname] = w.group(inner, [])
        top[lname] = w.group(members, [_h5_attr_strings('weight_names', [n for n, _ in weights])])
    root = w.group(top, [_h5_attr_strings('layer_names', [n for n, _ in layers]), _h5_attr_scalar_string('backend', backend),
                         _h5_attr_scalar_string('keras_version', keras_version)])
    with open(path, 'wb') as fh:
        fh.write(w.finish(root))


def keras_layers_from_weights(weights, component='real', dropout=True):
    """Container-named tensors -> the ``layers`` list of write_keras_hdf5_weights for the reference's model
    (DNN.py:176-234): inputs, flatten, concatenate, then per hidden layer fc_dense<i> / batch_normalization[_k] /
    drop<i> (between hidden layers only, :222), fc_regressor.  BatchNormalization auto-numbers continue from the real to
    the imag model, which is built second in the same process."""
    n_hidden = 0
    while f'fc_dense{n_hidden}.kernel' in weights:
        n_hidden += 1
    use_bn = 'bn0.gamma' in weights
    k = 0 if component == 'real' else 1
    sfx = lambda base, i: base + (f'_{i}' if i else '')
    out = [(f'input_{1 + 2 * k}', []), (sfx('flatten', k), []), (f'input_{2 + 2 * k}', []), (sfx('concatenate', k), [])]
    f32 = lambda a: np.asarray(a, np.float32)
    for i in range(n_hidden):
        out.append((f'fc_dense{i}', [(f'fc_dense{i}/kernel:0', f32(weights[f'fc_dense{i}.kernel'])), (f'fc_dense{i}/bias:0', f32(weights[f'fc_dense{i}.bias']).ravel())]))
        if use_bn:
            bn = sfx('batch_normalization', k * n_hidden + i)
            out.append((bn, [(f'{bn}/{v}:0', f32(weights[f'bn{i}.{v}']).ravel()) for v in _BN_VARS]))
        if dropout and i < n_hidden - 1:
            out.append((f'drop{i}', []))
    out.append(('fc_regressor', [('fc_regressor/kernel:0', f32(weights['fc_regressor.kernel'])), ('fc_regressor/bias:0', f32(weights['fc_regressor.bias']).ravel())]))
    return out
