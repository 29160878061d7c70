"""Command-line twin of the reference's test run
(``massiveMIMO_CSI_prediction_DNN.py --test``, lines 330-411, driven by
full_pipeline_maMIMO_DNNEst.sh:47):

    python -m dl_channel_estimation_mamimo_amd.cli --test -x testDataset.b --modeldir MODEL \\
           -d OUT --nn 1024 1024 --useBN --datasource matlab_maMimo

loads the pickle dataset, the two component models, runs LS + DNN over every packet on the GPU,
prints the evaluate() figure (MSE of each model against the stored labels, DNN.py:343) and writes
the per-packet ``test_csi_predictions_{real,imag}_N.mat`` files BER_test_maMIMO_LTF.m consumes.
Only the flags that fix the model shape are honoured; training flags
are not part of this path."""
import argparse
import os
import sys
import time

import numpy as np


def build_parser():
    p = argparse.ArgumentParser(description='Test CSI prediction network on MI355X')
    p.add_argument('--test', action='store_true', help='(kept for command-line compatibility; this tool only tests)')
    p.add_argument('--model', default='FC', help='DNN model type; only FC is on this path')
    p.add_argument('-x', required=True, help='dataset pickle written by create_massiveMIMO_CSIest_dnn_dataset.py')
    p.add_argument('--datasource', default='matlab_maMimo')
    p.add_argument('-d', '--workdir', default='checkpoint', help='output folder for the per-packet .mat files')
    p.add_argument('--modeldir', default='', help='folder holding {real,imag}_keras_model/ or <d>_weights-improvement.safetensors')
    p.add_argument('--nn', default=[256, 128], type=int, nargs='+', help='neurons per hidden layer')
    p.add_argument('--useBN', action='store_true')
    p.add_argument('--dropout', default=0.15, type=float, help='ignored at inference (identity)')
    p.add_argument('--execTime', action='store_true', help='print per-kernel HIP-event times')
    p.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    p.add_argument('--device', default=0, type=int)
    return p


def _find_weights(modeldir, d):
    from .model import WEIGHT_FILE
    for cand in (os.path.join(modeldir, d + '_keras_model', WEIGHT_FILE),
                 os.path.join(modeldir, d + '_weights-improvement.safetensors'),
                 os.path.join(modeldir, d + '_weights-improvement.pt')):
        if os.path.exists(cand):
            return cand
    print('Given model directory holds no weights for the %s model. Aborting...' % d)
    sys.exit(0)


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.model != 'FC' or args.datasource != 'matlab_maMimo':
        print('Only --model FC --datasource matlab_maMimo is on the MI355X path. Aborting...')
        sys.exit(0)
    modeldir = args.modeldir or args.workdir
    if not os.path.isdir(args.workdir):
        print('Given directory does not exists. Aborting...')          # DNN.py:113-115
        sys.exit(0)
    from . import dataset as ds
    from .engine import CsiEngine
    from .model import CSIModel, load_weight_file

    packed = ds.packets_from_dataset(ds.load_dataset(args.x))
    nt, nr, npkt = packed['nt'], packed['nr'], packed['npkt']
    eng = CsiEngine(nt, nr, hidden=args.nn, n_out=packed['labels'].shape[-1], use_bn=args.useBN,
                    device=args.device, dtype=args.dtype)
    for d in ('real', 'imag'):
        print('Working on *', d, '* model')
        m = CSIModel(eng, d).load_weights(load_weight_file(_find_weights(modeldir, d)))
        m.summary()
    eng.set_pilot(packed['pilot'])
    if args.execTime:
        eng.profile_enable(True)
    t0 = time.perf_counter()
    out_re, out_im = eng.predict(packed['ltf'])
    h_ls = eng.ls_estimate(packed['ltf'])
    dt = time.perf_counter() - t0
    for d, out, lab in (('real', out_re, packed['labels'].real), ('imag', out_im, packed['labels'].imag)):
        print('%s model: loss (mse vs labels) = %.6e' % (d, float(np.mean((out - lab) ** 2))))          # evaluate(), DNN.py:343
    lab = packed['labels']
    num = np.linalg.norm((h_ls - lab).reshape(npkt, -1), axis=1)
    print('LS(GPU) vs stored LS labels: max packet rel. error %.3e' % float(np.max(num / np.linalg.norm(lab.reshape(npkt, -1), axis=1))))
    files = ds.export_predictions(args.workdir, packed, out_re, out_im)
    print('%d packets (%d pair-channels) in %.3f s incl. host transfers; wrote %d .mat files to %s'
          % (npkt, npkt * nr * nt, dt, len(files['real']) + len(files['imag']), args.workdir))
    if args.execTime:
        print('******** Check timings!! ********')
        for k, v in eng.profile().items():
            if v['launches']:
                print('  %-20s %4d launches  %9.3f ms' % (k, v['launches'], v['ms']))
    return 0


if __name__ == '__main__':
    sys.exit(main())
