#!/usr/bin/env python3
"""Keras -> weight-container exporter.  Runs on a host WITH TensorFlow/Keras (it is
not runnable in the MI355X build image, which has neither TensorFlow nor h5py, and is therefore
untested there).  It rebuilds nothing: it opens what the reference saved -

    <d>_keras_model/                 (SavedModel dir, massiveMIMO_CSI_prediction_DNN.py:411), or
    <d>_weights-improvement.hdf5     (save_weights file, :319; needs --nn/--useBN/--lenLTF/--nTX
                                      to rebuild the graph of :176-234 before load_weights)

- walks the layers in order (the reference itself matches weights by topology because the
BatchNormalization layer names are auto-numbered), and writes ``<out>/<d>_keras_model/
weights.safetensors`` + ``config.json`` with the tensor names the MI355X library expects:
fc_dense{i}.kernel/.bias, bn{i}.gamma/.beta/.moving_mean/.moving_variance, fc_regressor.kernel/.bias."""
import argparse
import json
import os

import numpy as np


def build_reference_graph(len_ltf, ntx, nn, use_bn, n_out, dropout=0.15):
    """The FC graph of massiveMIMO_CSI_prediction_DNN.py:176-234 (test branch: no AWGN layer)."""
    from tensorflow.keras.layers import BatchNormalization, Concatenate, Dense, Dropout, Flatten, Input
    from tensorflow.keras.models import Model
    seq_in = Input(shape=(len_ltf, 1))
    seq_p = Input(shape=(ntx,))
    Dropout(0.15, name='drop_test')(seq_in)
    x = Concatenate(axis=1)([Flatten()(seq_in), seq_p])
    for i, n in enumerate(nn):
        x = Dense(n, activation='relu', name='fc_dense' + str(i))(x)
        if use_bn:
            x = BatchNormalization()(x)
        if i < len(nn) - 1 and dropout != 0.0:
            x = Dropout(dropout, name='drop' + str(i))(x)
    out = Dense(n_out, activation='linear', name='fc_regressor')(x)
    return Model([seq_in, seq_p], out)


def tensors_of(model):
    out, dense_i, bn_i = {}, 0, 0
    for layer in model.layers:
        cls = layer.__class__.__name__
        w = layer.get_weights()
        if cls == 'Dense':
            name = 'fc_regressor' if layer.name == 'fc_regressor' else 'fc_dense%d' % dense_i
            if name != 'fc_regressor':
                dense_i += 1
            out[name + '.kernel'], out[name + '.bias'] = w[0], w[1]
        elif cls == 'BatchNormalization':
            g, b, m, v = w                      # gamma, beta, moving_mean, moving_variance
            out['bn%d.gamma' % bn_i], out['bn%d.beta' % bn_i] = g, b
            out['bn%d.moving_mean' % bn_i], out['bn%d.moving_variance' % bn_i] = m, v
            bn_i += 1
    return {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in out.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--src', required=True, help='folder with the reference outputs')
    ap.add_argument('--out', required=True)
    ap.add_argument('--nn', type=int, nargs='+', default=[1024, 1024])
    ap.add_argument('--useBN', action='store_true')
    ap.add_argument('--nTX', type=int, required=True)
    ap.add_argument('--nRX', type=int, required=True)
    ap.add_argument('--nSubCarr', type=int, default=234)
    ap.add_argument('--pilot', default='', help='optional .npy with dataset[\'P\'] (stored transposed as pilot rows)')
    args = ap.parse_args()
    from tensorflow import keras
    from safetensors.numpy import save_file
    len_ltf = 320 * args.nTX
    for d in ('real', 'imag'):
        saved = os.path.join(args.src, d + '_keras_model')
        if os.path.isdir(saved):
            model = keras.models.load_model(saved)
        else:
            model = build_reference_graph(len_ltf, args.nTX, args.nn, args.useBN, args.nSubCarr)
            model.load_weights(os.path.join(args.src, d + '_weights-improvement.hdf5'))
        t = tensors_of(model)
        assert t['fc_dense0.kernel'].shape[0] == len_ltf + args.nTX, t['fc_dense0.kernel'].shape
        if args.pilot:
            t['pilot'] = np.ascontiguousarray(np.load(args.pilot).T, dtype=np.float32)
        dst = os.path.join(args.out, d + '_keras_model')
        os.makedirs(dst, exist_ok=True)
        save_file(t, os.path.join(dst, 'weights.safetensors'))
        with open(os.path.join(dst, 'config.json'), 'w') as f:
            json.dump(dict(component=d, nt=args.nTX, nr=args.nRX, len_ltf=len_ltf, hidden=list(args.nn),
                           n_out=args.nSubCarr, use_bn=bool(args.useBN), bn_eps=1e-3, datasource='matlab_maMimo'), f, indent=1)
        print('wrote', dst)


if __name__ == '__main__':
    main()
