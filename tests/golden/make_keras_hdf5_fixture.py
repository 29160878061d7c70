#!/usr/bin/env python3
"""Writes tests/golden/keras_weights_{real,imag}.hdf5 with the GENUINE HDF5 library (libhdf5 1.10.6, the C library
h5py wraps; found in this image under /opt/conda/lib and driven through ctypes - neither h5py nor TensorFlow is
installed) in the layout keras 2.3 ``Model.save_weights`` / ``ModelCheckpoint`` produce for the reference's
network (massiveMIMO_CSI_prediction_DNN.py:176-234, :279-281, :319):

    /                      attrs: layer_names [n] S<len>, backend, keras_version
    /<layer>               attrs: weight_names [k] S<len>   (float64 [0] for layers without weights, as h5py
                                                              stores ``np.asarray([])``)
    /<layer>/<layer>/kernel:0 ...   contiguous little-endian float32 datasets

Layer list and order = ``model.layers`` of the reference's functional model with --useBN and dropout 0.15: the two
inputs, flatten, concatenate, fc_dense0, batch_normalization[_k], drop0, fc_dense1, batch_normalization_[k+1],
fc_regressor; the imag model is built second in the same process, so its BatchNormalization layers carry the next
auto-numbers (_2, _3) - which is why the reference can only match them by order.

The tensors written are also stored in keras_weights_expected.npz (container names) for the reader test.
Not run by the test-suite; needs /opt/conda/lib/libhdf5.so.103."""
import ctypes
import os
import sys

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
LIB = '/opt/conda/lib/libhdf5.so.103'

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64


class H5:
    def __init__(self):
        self.lib = lib = ctypes.CDLL(LIB)
        assert lib.H5open() >= 0
        for fn, res, args in (
                ('H5Fcreate', hid_t, [ctypes.c_char_p, ctypes.c_uint, hid_t, hid_t]), ('H5Fclose', ctypes.c_int, [hid_t]),
                ('H5Gcreate2', hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t]), ('H5Gclose', ctypes.c_int, [hid_t]),
                ('H5Screate_simple', hid_t, [ctypes.c_int, ctypes.POINTER(hsize_t), ctypes.POINTER(hsize_t)]),
                ('H5Screate', hid_t, [ctypes.c_int]), ('H5Sclose', ctypes.c_int, [hid_t]),
                ('H5Dcreate2', hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
                ('H5Dwrite', ctypes.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, ctypes.c_void_p]), ('H5Dclose', ctypes.c_int, [hid_t]),
                ('H5Acreate2', hid_t, [hid_t, ctypes.c_char_p, hid_t, hid_t, hid_t, hid_t]),
                ('H5Awrite', ctypes.c_int, [hid_t, hid_t, ctypes.c_void_p]), ('H5Aclose', ctypes.c_int, [hid_t]),
                ('H5Tcopy', hid_t, [hid_t]), ('H5Tset_size', ctypes.c_int, [hid_t, ctypes.c_size_t]),
                ('H5Tset_strpad', ctypes.c_int, [hid_t, ctypes.c_int]), ('H5Tclose', ctypes.c_int, [hid_t])):
            f = getattr(lib, fn)
            f.restype, f.argtypes = res, args
        g = lambda n: hid_t.in_dll(lib, n).value
        self.F32LE, self.F64LE, self.C_S1 = g('H5T_IEEE_F32LE_g'), g('H5T_IEEE_F64LE_g'), g('H5T_C_S1_g')

    def space(self, shape):
        if shape == ():
            return self.lib.H5Screate(0)                         # H5S_SCALAR
        dims = (hsize_t * len(shape))(*shape)
        return self.lib.H5Screate_simple(len(shape), dims, None)

    def fixed_str(self, n):
        t = self.lib.H5Tcopy(self.C_S1)
        self.lib.H5Tset_size(t, max(n, 1))
        self.lib.H5Tset_strpad(t, 1)                             # H5T_STR_NULLPAD: what h5py maps numpy 'S' to
        return t

    def attr_strings(self, loc, name, items):
        """list of bytes -> numpy 'S<max>' array attribute (h5py's conversion of a list of bytes)"""
        if not items:
            sp = self.space((0,))                                # np.asarray([]) is float64 of shape (0,)
            a = self.lib.H5Acreate2(loc, name, self.F64LE, sp, 0, 0)
            assert a >= 0
            self.lib.H5Aclose(a); self.lib.H5Sclose(sp)
            return
        arr = np.asarray(items, dtype='S')
        t, sp = self.fixed_str(arr.dtype.itemsize), self.space(arr.shape)
        a = self.lib.H5Acreate2(loc, name, t, sp, 0, 0)
        assert a >= 0 and self.lib.H5Awrite(a, t, arr.ctypes.data) >= 0
        self.lib.H5Aclose(a); self.lib.H5Sclose(sp); self.lib.H5Tclose(t)

    def attr_scalar_fixed(self, loc, name, value):
        arr = np.asarray(value, dtype='S')
        t, sp = self.fixed_str(arr.dtype.itemsize), self.space(())
        a = self.lib.H5Acreate2(loc, name, t, sp, 0, 0)
        assert a >= 0 and self.lib.H5Awrite(a, t, arr.ctypes.data) >= 0
        self.lib.H5Aclose(a); self.lib.H5Sclose(sp); self.lib.H5Tclose(t)

    def attr_scalar_vlen(self, loc, name, value):
        """variable-length string (what h5py >= 3 writes for a python bytes / str scalar)"""
        t = self.lib.H5Tcopy(self.C_S1)
        self.lib.H5Tset_size(t, ctypes.c_size_t(-1).value)       # H5T_VARIABLE
        sp = self.space(())
        buf = ctypes.c_char_p(value)
        a = self.lib.H5Acreate2(loc, name, t, sp, 0, 0)
        assert a >= 0 and self.lib.H5Awrite(a, t, ctypes.byref(buf)) >= 0
        self.lib.H5Aclose(a); self.lib.H5Sclose(sp); self.lib.H5Tclose(t)

    def dataset(self, loc, name, arr):
        arr = np.ascontiguousarray(arr, dtype='<f4')
        sp = self.space(arr.shape)
        d = self.lib.H5Dcreate2(loc, name, self.F32LE, sp, 0, 0, 0)
        assert d >= 0 and self.lib.H5Dwrite(d, self.F32LE, 0, 0, 0, arr.ctypes.data) >= 0
        self.lib.H5Dclose(d); self.lib.H5Sclose(sp)


def write_keras_weights(h5, path, layers):
    """layers: [(layer name, [(weight name, array)])] in model.layers order."""
    lib = h5.lib
    f = lib.H5Fcreate(path.encode(), 2, 0, 0)                    # H5F_ACC_TRUNC, default property lists = earliest format
    assert f >= 0
    h5.attr_strings(f, b'layer_names', [n.encode() for n, _ in layers])
    h5.attr_scalar_fixed(f, b'backend', b'tensorflow')
    h5.attr_scalar_vlen(f, b'keras_version', b'2.4.0')           # tf.keras of TF 2.3 reports 2.4.0
    for name, weights in layers:
        g = lib.H5Gcreate2(f, name.encode(), 0, 0, 0)
        assert g >= 0
        h5.attr_strings(g, b'weight_names', [w.encode() for w, _ in weights])
        if weights:
            inner = lib.H5Gcreate2(g, name.encode(), 0, 0, 0)    # 'fc_dense0/kernel:0' inside group 'fc_dense0'
            for w, arr in weights:
                assert w.startswith(name + '/')
                h5.dataset(inner, w.split('/', 1)[1].encode(), arr)
            lib.H5Gclose(inner)
        lib.H5Gclose(g)
    lib.H5Fclose(f)


def reference_layers(w, bn_first):
    """model.layers of DNN.py:176-234 (--useBN, --dropout 0.15, two hidden layers) with the tensors of the
    container-named dict w; BatchNormalization auto-numbering starts at bn_first."""
    def bn_name(k):
        return 'batch_normalization' + (f'_{k}' if k else '')
    out = [('input_%d' % (1 + 2 * (bn_first // 2)), []), ('flatten' + ('_%d' % (bn_first // 2) if bn_first else ''), []),
           ('input_%d' % (2 + 2 * (bn_first // 2)), []), ('concatenate' + ('_%d' % (bn_first // 2) if bn_first else ''), [])]
    i = 0
    while f'fc_dense{i}.kernel' in w:
        out.append((f'fc_dense{i}', [(f'fc_dense{i}/kernel:0', w[f'fc_dense{i}.kernel']), (f'fc_dense{i}/bias:0', w[f'fc_dense{i}.bias'])]))
        b = bn_name(bn_first + i)
        out.append((b, [(f'{b}/{v}:0', w[f'bn{i}.{v}']) for v in ('gamma', 'beta', 'moving_mean', 'moving_variance')]))
        if f'fc_dense{i + 1}.kernel' in w:
            out.append((f'drop{i}', []))
        i += 1
    out.append(('fc_regressor', [('fc_regressor/kernel:0', w['fc_regressor.kernel']), ('fc_regressor/bias:0', w['fc_regressor.bias'])]))
    return out


def main():
    from oracle import csi_oracle as o
    rng = np.random.default_rng(20240915)
    nt, hidden = 4, [16, 8]
    expected = {}
    h5 = H5()
    for d, bn_first in (('real', 0), ('imag', 2)):
        w = o.make_weights(rng, 320 * nt + nt, hidden, 234)
        write_keras_weights(h5, os.path.join(OUT, f'keras_weights_{d}.hdf5'), reference_layers(w, bn_first))
        for k, v in w.items():
            if isinstance(v, np.ndarray) and v.ndim >= 1:
                expected[f'{d}.{k}'] = v
    np.savez_compressed(os.path.join(OUT, 'keras_weights_expected.npz'), nt=nt, **expected)
    print('wrote keras_weights_{real,imag}.hdf5 + keras_weights_expected.npz')


if __name__ == '__main__':
    main()
