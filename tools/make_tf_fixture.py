#!/usr/bin/env python3
"""make_tf_fixture.py - run on a host WITH TensorFlow 2.x (any version that still has tf.keras; the reference used 2.3):
writes model files the way the reference does, for tests/golden/tf_written/.

It builds the FC graph of massiveMIMO_CSI_prediction_DNN.py:176-234 (test branch) at Nt = 4, hidden 16 x 8, BatchNormalization
on, 234 outputs - the real model first, then the imag model IN THE SAME PROCESS, so that the imag model's BatchNormalization
layers get the auto-numbered names the reference's files have - gives every variable a seeded non-trivial value, and saves

    <out>/<d>_weights-improvement.hdf5     Model.save_weights (DNN.py:319; Keras HDF5, read by Model.load_weights DNN.py:334)
    <out>/<d>_keras_model/                 Model.save       (DNN.py:411; SavedModel, read by keras.models.load_model inference.py:15-16)
    <out>/expected.npz                     <d>/<i> = i-th array of Model.get_weights() (model order), x_sig / x_p / <d>_y = a seeded
                                           batch and Model.predict on it

    python tools/make_tf_fixture.py tests/golden/tf_written

Nothing in the MI355X build image can run this (no TensorFlow there); tests/test_host.py::test_tensorflow_written_model_files
consumes the output when it exists."""
import os
import sys

import numpy as np


def build(len_ltf, ntx, nn, n_out):
    from tensorflow.keras.layers import BatchNormalization, Concatenate, Dense, Dropout, Flatten, Input
    from tensorflow.keras.models import Model
    seq_in = Input(shape=(len_ltf, 1))
    seq_p = Input(shape=(ntx,))
    x = Concatenate(axis=1)([Flatten()(seq_in), seq_p])
    for i, n in enumerate(nn):
        x = Dense(n, activation='relu', name='fc_dense' + str(i))(x)
        x = BatchNormalization()(x)
        if i < len(nn) - 1:
            x = Dropout(0.15, name='drop' + str(i))(x)
    return Model([seq_in, seq_p], Dense(n_out, activation='linear', name='fc_regressor')(x))


def main(out):
    import tensorflow as tf
    os.makedirs(out, exist_ok=True)
    nt, nn, n_out = 4, (16, 8), 234
    len_ltf = 320 * nt
    rng = np.random.default_rng(2025)
    x_sig = rng.standard_normal((6, len_ltf, 1)).astype(np.float32)
    x_p = rng.choice([-1.0, 1.0], (6, nt)).astype(np.float32)
    blob = {'x_sig': x_sig, 'x_p': x_p, 'tf_version': np.array(tf.__version__)}
    for d in ('real', 'imag'):
        m = build(len_ltf, nt, nn, n_out)
        ws = []
        for w in m.get_weights():
            v = rng.standard_normal(w.shape).astype(np.float32) * 0.1
            if w.ndim == 1 and w.shape[0] in nn:
                v = np.abs(v) + 0.5                     # gamma / moving_variance stay positive; harmless for bias / beta / mean
            ws.append(v)
        m.set_weights(ws)
        m.save_weights(os.path.join(out, d + '_weights-improvement.hdf5'))
        m.save(os.path.join(out, d + '_keras_model'))
        for i, w in enumerate(m.get_weights()):
            blob['%s/%d' % (d, i)] = w
        blob[d + '_names'] = np.array([w.name for w in m.weights])
        blob[d + '_y'] = m.predict([x_sig, x_p])
    np.savez_compressed(os.path.join(out, 'expected.npz'), **blob)
    print('wrote', sorted(os.listdir(out)))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'tests/golden/tf_written')
