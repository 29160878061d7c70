#!/usr/bin/env python3
"""Generates the committed golden vectors under tests/golden/ by RUNNING THE REFERENCE'S OWN
pure-numpy code in the build container (where /root/reference is mounted).  It is not
run by the test-suite and cannot run on the GPU box (no /root/reference there); only the
.npz files it writes travel.

What is exercised, and how:
  * massiveMIMO_dataGenerator.DataGenerator (__getitem__ -> __data_generation,
    reorder_indexes, set_batchsize, __len__) on a small seeded dataset dict in the format
    create_massiveMIMO_CSIest_dnn_dataset.py:125 writes  -> ref_datagen_nt4.npz
  * inference.CSIPredictor.inference / preprocess_data / postprocess_data with the two
    keras models replaced by deterministic numpy stand-ins for ``.predict`` (the models are
    an INPUT of that code; its own arithmetic - dtype check, X.real/X.imag split,
    ``real + 1j*imag``, null re-insertion, ifftshift - is what gets recorded)
    -> ref_inference_rice.npz

TensorFlow is not installed here.  The two reference modules only need the NAMES
``tensorflow.keras`` / ``tensorflow.keras.utils.Sequence`` to import (inference.py:4,
massiveMIMO_dataGenerator.py:3); an empty import shim supplies them (Sequence = a plain
base class with no behaviour).  No TensorFlow functionality is emulated and none of the
recorded numbers pass through the shim.  The Dense/BN arithmetic itself therefore stays
unpinned (see oracle/csi_oracle.py header).
"""
import os
import sys
import types
import numpy as np

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def install_import_shim():
    tf = types.ModuleType('tensorflow')
    keras = types.ModuleType('tensorflow.keras')
    utils = types.ModuleType('tensorflow.keras.utils')

    class Sequence(object):          # base class only; no behaviour
        pass

    utils.Sequence = Sequence
    keras.utils = utils
    tf.keras = keras
    sys.modules['tensorflow'] = tf
    sys.modules['tensorflow.keras'] = keras
    sys.modules['tensorflow.keras.utils'] = utils


def make_dataset(rng, npkt, nr, nt, n_sub=234):
    """Small dataset dict in the layout of create_massiveMIMO_CSIest_dnn_dataset.py:26-126."""
    len_ltf = 320 * nt
    n = npkt * nr * nt
    X = np.zeros((n, 2), dtype=int)
    y_re = rng.standard_normal((n, n_sub))
    y_im = rng.standard_normal((n, n_sub))
    ltf = {}
    keys = []
    for p in range(npkt):
        for r in range(nr):
            while True:
                key = int(rng.integers(0, 2 ** 32))
                if key not in ltf:
                    break
            ltf[key] = {'real': rng.standard_normal(len_ltf), 'imag': rng.standard_normal(len_ltf)}
            keys.append(key)
            for t in range(nt):
                X[p * nr * nt + r * nt + t] = [key, t]
    # deliberately NON-symmetric pilot matrix so that a transposition error shows
    P_matlab = rng.integers(-3, 4, size=(nt, nt)).astype(np.float64)
    P_py = P_matlab.T.copy()         # what h5py hands to python (mk.py:37)
    sim = {'FFTLength': 256, 'CPLen': 64, 'numSym': nt, 'symOffset': 64, 'nTX': nt, 'nRX': nr,
           'lenLTF': len_ltf, 'nSubCarr': n_sub}
    ds = {'X': X, 'y': {'real': y_re, 'imag': y_im}, 'LTF': ltf, 'P': P_py, 'simParams': sim}
    return ds, keys, P_matlab


def golden_datagen():
    import massiveMIMO_dataGenerator as gen
    rng = np.random.default_rng(20240901)
    npkt, nr, nt = 3, 2, 4
    ds, keys, P_matlab = make_dataset(rng, npkt, nr, nt)
    n = npkt * nr * nt
    out = {
        'npkt': npkt, 'nr': nr, 'nt': nt,
        'ds_X': ds['X'], 'ds_keys': np.array(keys, dtype=np.int64),
        'ds_ltf_real': np.stack([ds['LTF'][k]['real'] for k in keys]),
        'ds_ltf_imag': np.stack([ds['LTF'][k]['imag'] for k in keys]),
        'ds_y_real': ds['y']['real'], 'ds_y_imag': ds['y']['imag'],
        'ds_P': ds['P'], 'P_matlab': P_matlab,
    }
    np.random.seed(7)                 # DataGenerator shuffles with the global numpy RNG
    for d in ('real', 'imag'):
        g = gen.DataGenerator(list(range(n)), ds, d, ds['simParams'], datasource='matlab_maMimo',
                              method='default', batch_size=5)        # shuffle=True (default)
        out[f'{d}_len_bs5'] = len(g)
        # the test branch of the reference: DNN.py:337,339
        g.reorder_indexes()
        g.set_batchsize(ds['simParams']['nTX'] * ds['simParams']['nRX'])
        out[f'{d}_len'] = len(g)
        xs, xp, ys = [], [], []
        for b in range(len(g)):
            X, y, rms_fact = g[b]
            assert rms_fact is None
            xs.append(X[0]); xp.append(X[1]); ys.append(y)
        out[f'{d}_Xsig'] = np.stack(xs)      # [nbatch, Nt*Nr, lenLTF, 1]
        out[f'{d}_Xp'] = np.stack(xp)        # [nbatch, Nt*Nr, Nt]
        out[f'{d}_y'] = np.stack(ys)         # [nbatch, Nt*Nr, 234]
    np.savez_compressed(os.path.join(OUT, 'ref_datagen_nt4.npz'), **out)
    print('wrote ref_datagen_nt4.npz', {k: np.asarray(v).shape for k, v in out.items()})


class _StandInModel:
    """Deterministic stand-in for a loaded keras model: float32 [bs, n_in] -> [bs, n_out]."""

    def __init__(self, A, b):
        self.A = A.astype(np.float32)
        self.b = b.astype(np.float32)
        self.calls = []

    def predict(self, X, batch_size=None):
        self.calls.append((np.asarray(X).dtype.str, int(batch_size)))
        return (np.asarray(X, dtype=np.float32) @ self.A + self.b).astype(np.float32)


def golden_inference():
    import inference as inf
    rng = np.random.default_rng(20240902)
    bs, n_in = 7, 160
    A_re, b_re = rng.standard_normal((n_in, 52)) / 8, rng.standard_normal(52)
    A_im, b_im = rng.standard_normal((n_in, 52)) / 8, rng.standard_normal(52)
    m_re, m_im = _StandInModel(A_re, b_re), _StandInModel(A_im, b_im)
    inf.CSIPredictor.load_model = lambda self: (m_re, m_im)      # the models are an input
    pred = inf.CSIPredictor('unused', experiment='RICE_RENEW', verbose=False)
    x = (rng.standard_normal((bs, n_in)) + 1j * rng.standard_normal((bs, n_in))).astype(np.complex128)
    y = pred.inference(x)
    ramp = (np.arange(1, 53)[None, :] + 1j * np.arange(101, 153)[None, :]).astype(np.complex64)
    post_ramp = pred.postprocess_data(ramp)
    codes = {}
    for name, fn in (('bad_dtype', lambda: pred.preprocess_data(x.astype(np.complex64))),
                     ('bad_width', lambda: pred.postprocess_data(np.zeros((2, 51), dtype=np.complex64)))):
        try:
            fn()
            codes[name] = 0
        except SystemExit as e:
            codes[name] = int(e.code)
    out = dict(A_re=A_re, b_re=b_re, A_im=A_im, b_im=b_im, x=x, y=y, y_dtype=str(y.dtype),
               model_calls_real=np.array(m_re.calls[0], dtype=object).astype(str),
               model_calls_imag=np.array(m_im.calls[0], dtype=object).astype(str),
               ramp=ramp, post_ramp=post_ramp, post_ramp_dtype=str(post_ramp.dtype),
               exit_bad_dtype=codes['bad_dtype'], exit_bad_width=codes['bad_width'])
    np.savez_compressed(os.path.join(OUT, 'ref_inference_rice.npz'), **out)
    print('wrote ref_inference_rice.npz', y.shape, y.dtype, codes, m_re.calls, m_im.calls)


if __name__ == '__main__':
    assert os.path.isdir(REF), 'this script only runs where the reference is mounted'
    install_import_shim()
    sys.path.insert(0, REF)
    golden_datagen()
    golden_inference()
