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
  * massiveMIMO_dataGenerator.DataGenerator with method='reshape' (massiveMIMO_dataGenerator.py:425-458,
    "THIS METHOD PERFORMS COMPLETE OFDM DEMODULATION"): the reference's own numpy statement of the OFDM
    demodulation convention - column-major split into numSym symbols of FFTLength + CPLen samples,
    cyclic-prefix removal index list, per-symbol FFT, fftshift - run on the same small dataset.  The
    batch rows it returns keep only the REAL part of one symbol's spectrum (a complex array assigned
    into a float array), so besides those rows the generator's own intermediate arrays (input2D,
    afterCPRemoval, afterFFT before and after the shift) are recorded through a line tracer on
    __data_generation: they are the reference's variables, read while the reference's code runs -
    nothing is recomputed here.                                              -> ref_ofdm_reshape_nt4.npz

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


def golden_ofdm_reshape():
    """Runs DataGenerator(method='reshape') (massiveMIMO_dataGenerator.py:425-458) on the Nt=4 dataset and
    records (a) the batches it returns and (b) its intermediate arrays, captured by a tracer."""
    import warnings
    import massiveMIMO_dataGenerator as gen
    rng = np.random.default_rng(20240901)                 # the same dataset as golden_datagen()
    npkt, nr, nt = 3, 2, 4
    ds, keys, P_matlab = make_dataset(rng, npkt, nr, nt)
    n = npkt * nr * nt
    prm = dict(ds['simParams'])
    # prm['ltf_freqdom'] is commented out in the reference's loadDataset (:40-42); the method appends it to
    # every row, so any 234-vector serves - a recognisable ramp
    prm['ltf_freqdom'] = np.arange(1, 235, dtype=np.float64)
    code = gen.DataGenerator._DataGenerator__data_generation.__code__
    first_line = code.co_firstlineno
    captured = {}                                          # d -> sample position -> dict of arrays

    def tracer_for(d):
        store = captured.setdefault(d, [])
        state = {'last': None, 'i_seen': None}

        def local_trace(frame, event, arg):
            if event not in ('line', 'return'):
                return local_trace
            loc = frame.f_locals
            # afterFFT is bound at :452 (the FFT) and re-bound at :453 (fftshift): a NEW object in that name is
            # the first binding of loop iteration i (before the shift) or the second (after it); between
            # iterations the name still holds the previous sample's array, which is not new
            a = loc.get('afterFFT')
            if a is None or a is state['last']:
                return local_trace
            state['last'] = a
            if loc['i'] != state['i_seen']:
                state['i_seen'] = loc['i']
                store.append({'i': int(loc['i']), 'sampleIx': int(loc['sampleIx']), 'input2D': loc['input2D'].copy(),
                              'afterCPRemoval': loc['afterCPRemoval'].copy(), 'noCP_ix': np.array(loc['noCP_ix']),
                              'fft_pre_shift': np.array(a).copy()})
            else:
                assert 'fft_post_shift' not in store[-1]
                store[-1]['fft_post_shift'] = np.array(a).copy()
            return local_trace

        def global_trace(frame, event, arg):
            if event == 'call' and frame.f_code is code:
                state['last'], state['i_seen'] = None, None
                return local_trace
            return None
        return global_trace

    out = {'npkt': npkt, 'nr': nr, 'nt': nt, 'ds_X': ds['X'], 'ds_P': ds['P'], 'P_matlab': P_matlab,
           'ltf_freqdom': prm['ltf_freqdom'], 'ref_first_line': first_line,
           'ds_ltf_real': np.stack([ds['LTF'][k]['real'] for k in keys]),
           'ds_ltf_imag': np.stack([ds['LTF'][k]['imag'] for k in keys])}
    for d in ('real', 'imag'):
        g = gen.DataGenerator(list(range(n)), ds, d, prm, datasource='matlab_maMimo', method='reshape',
                              batch_size=nt * nr, shuffle=False)
        rows = []
        sys.settrace(tracer_for(d))
        try:
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')            # ComplexWarning: the reference drops the imaginary part (:454)
                for b in range(len(g)):
                    X, y, _ = g[b]
                    rows.append(np.asarray(X))
        finally:
            sys.settrace(None)
        out[f'{d}_X'] = np.stack(rows)                     # [npkt, Nt*Nr, 256 + Nt + 234]
        recs = captured[d]
        assert len(recs) == n and all('fft_post_shift' in r for r in recs), (len(recs), [sorted(r) for r in recs[:2]])
        assert [r['sampleIx'] for r in recs] == list(range(n))
        # the Nt samples of an rx antenna share the preamble: keep iTx = 0 of every (packet, rx)
        sel = [r for r in recs if r['sampleIx'] % nt == 0]
        out[f'{d}_input2D'] = np.stack([r['input2D'] for r in sel])                # [npkt*nr, 320, Nt]
        out[f'{d}_afterCPRemoval'] = np.stack([r['afterCPRemoval'] for r in sel])  # [npkt*nr, 256, Nt]
        out[f'{d}_fft_pre_shift'] = np.stack([r['fft_pre_shift'] for r in sel])    # complex [npkt*nr, 256, Nt]
        out[f'{d}_fft_post_shift'] = np.stack([r['fft_post_shift'] for r in sel])
        out['noCP_ix'] = recs[0]['noCP_ix']
    np.savez_compressed(os.path.join(OUT, 'ref_ofdm_reshape_nt4.npz'), **out)
    print('wrote ref_ofdm_reshape_nt4.npz', {k: np.asarray(v).shape for k, v in out.items()})


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
    golden_ofdm_reshape()
