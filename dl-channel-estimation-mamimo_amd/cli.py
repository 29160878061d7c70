"""Command-line twin of the reference's test run
(``massiveMIMO_CSI_prediction_DNN.py --test``, lines 330-411, driven by
full_pipeline_maMIMO_DNNEst.sh:47) and, with ``--train``, of its training run (lines 272-319,
pipe.sh:40: ``--train -x dataset.b --nn 1024 1024 -d MODEL --bs 256 --epochs 1000 --method default_SNR --useBN``),
which writes ``<d>_weights-improvement.safetensors`` into the work directory:

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
    p.add_argument('--test', action='store_true', help='test run (default)')
    p.add_argument('--train', action='store_true', help='train both component models on the dataset (on-box fine-tuning)')
    p.add_argument('--epochs', default=500, type=int)
    p.add_argument('--lr', default=0.0001, type=float)
    p.add_argument('--bs', default=256, type=int)
    p.add_argument('--method', default='default', help="'default_SNR': AWGN on the LTF input at a random SNR per batch")
    p.add_argument('--valTrainRatio', default=0.15, type=float)
    p.add_argument('--valSameTrain', action='store_true')
    p.add_argument('--onlyReal', action='store_true')
    p.add_argument('--onlyImag', action='store_true')
    p.add_argument('--init', default='', help='folder with weights to start from (default: Glorot-uniform initialisation)')
    p.add_argument('--seed', default=0, type=int)
    p.add_argument('--model', default='FC', help='DNN model type; only FC is on this path')
    p.add_argument('-x', required=True, help='dataset pickle written by create_massiveMIMO_CSIest_dnn_dataset.py')
    p.add_argument('--datasource', default='matlab_maMimo')
    p.add_argument('-d', '--workdir', default='checkpoint', help='output folder for the per-packet .mat files')
    p.add_argument('--modeldir', default='', help='folder holding {real,imag}_keras_model/ or <d>_weights-improvement.safetensors')
    p.add_argument('--nn', default=[256, 128], type=int, nargs='+', help='neurons per hidden layer')
    p.add_argument('--useBN', action='store_true')
    p.add_argument('--dropout', default=0.15, type=float, help='training only; identity at inference')
    p.add_argument('--execTime', action='store_true', help='print per-kernel HIP-event times')
    p.add_argument('--dtype', default='f32', choices=['f32', 'bf16'])
    p.add_argument('--device', default=0, type=int)
    return p


def _find_weights(modeldir, d):
    from .model import WEIGHT_FILE
    # the checkpoint of the last fit comes first, as in the reference's test branch (DNN.py:279-281,334 always loads
    # <d>_weights-improvement.hdf5): a <d>_keras_model/ folder in the same directory is what an EARLIER --test run
    # saved (DNN.py:411) and may be stale after a re-train
    for cand in (os.path.join(modeldir, d + '_weights-improvement.hdf5'),             # the reference's own checkpoint name
                 os.path.join(modeldir, d + '_weights-improvement.h5'),
                 os.path.join(modeldir, d + '_weights-improvement.safetensors'),
                 os.path.join(modeldir, d + '_weights-improvement.pt'),
                 os.path.join(modeldir, d + '_weights-improvement.npz'),
                 os.path.join(modeldir, d + '_keras_model', WEIGHT_FILE),
                 os.path.join(modeldir, d + '_keras_model')):                         # a TF SavedModel directory (DNN.py:411)
        if os.path.exists(cand):
            return cand
    print('Given model directory holds no weights for the %s model. Aborting...' % d)
    sys.exit(0)


def train_main(args):
    """--train: DNN.py:104-111 (work directory), :124-151 (split + generators), :272-319 (fit, save)."""
    from . import dataset as ds
    from . import trainer
    from .engine import CsiEngine
    from .model import load_weight_file, save_weight_file
    os.makedirs(args.workdir, exist_ok=True)
    data = ds.load_dataset(args.x)
    sim = data['simParams']
    nt, nr = int(sim['nTX']), int(sim['nRX'])
    n_out = int(np.asarray(data['y']['real']).shape[1])
    if args.valSameTrain:
        print('WARNING! Validation SAME AS Training!')
        train_ids = val_ids = list(range(int(np.asarray(data['X']).shape[0])))
    else:
        print('Validation separate from Training')
        train_ids, val_ids = ds.split_train_val(data, args.valTrainRatio)
    # under torchrun: one process per GPU, samples sharded by rank, gradients all-reduced over RCCL
    from . import dist
    rank, world, local = dist.env_rank_world()
    if world > 1:
        dist.init_process_group(os.environ.get('CSI_DIST_BACKEND', 'nccl'))       # before the engine: see _lib.load_library
        train_ids, val_ids = train_ids[rank::world], val_ids[rank::world]
        n = int(dist.all_reduce_min(len(train_ids) // args.bs))
        train_ids = train_ids[:n * args.bs]                                       # same number of steps on every rank
    eng = CsiEngine(nt, nr, hidden=args.nn, n_out=n_out, use_bn=args.useBN, device=(local % dist.local_device_count() if world > 1 else args.device))
    eng.set_pilot(np.asarray(data['P'], dtype=np.float64).T)          # the pilot columns of a sample are rows of P (resident dataset)
    dims = ['real'] if args.onlyReal else (['imag'] if args.onlyImag else ['real', 'imag'])
    for d in dims:
        print('Working on *', d, '* model')
        tr = ds.SampleGenerator(train_ids, data, d, batch_size=args.bs, shuffle=True, seed=args.seed)
        va = ds.SampleGenerator(val_ids, data, d, batch_size=args.bs, shuffle=True, seed=args.seed + 1)
        # decided by all ranks together: a rank leaving alone would strand the others in the gradient all-reduce
        enough = min(len(tr), len(va))
        if world > 1:
            enough = int(dist.all_reduce_min(enough))
        if enough == 0:
            print('Not enough samples for one batch of %d in the training / validation split%s. Aborting...'
                  % (args.bs, ' of at least one rank' if world > 1 else ''))
            sys.exit(2)
        init = load_weight_file(_find_weights(args.init, d)) if args.init else None
        if world > 1 and init is None:
            # identical initial tensors on every rank: rank 0 draws them (Glorot through a scratch trainer), all receive
            if rank == 0:
                eng.train_begin(d, lr=args.lr, dropout=args.dropout, seed=args.seed)
                init = eng.train_weights(d)
                eng.train_end(d, commit=False)
            init = dist.broadcast_weights(init, src=0)
        hist = trainer.fit(eng, d, tr, va, epochs=args.epochs, lr=args.lr, dropout=args.dropout, weights=init,
                           method=args.method, seed=args.seed, verbose=(rank == 0), data_parallel=(world > 1), resident=data)
        if rank == 0:
            path = os.path.join(args.modeldir or args.workdir, d + '_weights-improvement.safetensors')
            save_weight_file(path, hist['weights'])
            # ... and the file the reference itself writes here (DNN.py:319 save_weights): a Keras HDF5 checkpoint that
            # keras load_weights (DNN.py:334) takes back, so MI355X-trained weights enter the reference's own pipeline
            save_weight_file(os.path.join(args.modeldir or args.workdir, d + '_weights-improvement.hdf5'), hist['weights'], component=d)
            print('%s model: best val_loss %.6e after %d epochs; weights saved to %s' % (d, hist['best_val_loss'], len(hist['loss']), path))
    return 0


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.model != 'FC' or args.datasource != 'matlab_maMimo':
        print('Only --model FC --datasource matlab_maMimo is on the MI355X path. Aborting...')
        sys.exit(0)
    modeldir = args.modeldir or args.workdir
    if args.train:
        return train_main(args)
    if not os.path.isdir(args.workdir):
        print('Given directory does not exists. Aborting...')          # DNN.py:113-115
        sys.exit(0)
    from . import dataset as ds
    from .engine import CsiEngine
    from .model import CSIModel, load_weight_file

    data = ds.load_dataset(args.x)
    packed = ds.packets_from_dataset(data)
    if args.valSameTrain:
        print('WARNING! Validation SAME AS Training!')                 # DNN.py:130-133: every packet is tested
    else:
        # DNN.py:125-128 + loadDataset (massiveMIMO_dataGenerator.py:46-55): the test set is the LAST
        # floor(Npkt * valTrainRatio) packets, exported as files 1..n
        print('Validation separate from Training')
        _, val_ids = ds.split_train_val(data, args.valTrainRatio)
        n_val = len(val_ids) // (packed['nt'] * packed['nr'])
        first = packed['npkt'] - n_val
        packed = dict(packed, ltf=packed['ltf'][first:], labels=packed['labels'][first:], npkt=n_val)
        if n_val == 0:
            print('The validation split holds no packet (valTrainRatio %.3g of %d packets). Aborting...' % (args.valTrainRatio, first))
            sys.exit(0)
    nt, nr, npkt = packed['nt'], packed['nr'], packed['npkt']
    eng = CsiEngine(nt, nr, hidden=args.nn, n_out=packed['labels'].shape[-1], use_bn=args.useBN,
                    device=args.device, dtype=args.dtype)
    models = {}
    for d in ('real', 'imag'):
        print('Working on *', d, '* model')
        models[d] = CSIModel(eng, d).load_weights(load_weight_file(_find_weights(modeldir, d)))
        models[d].summary()
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
    # the number the MATLAB evaluation reports per estimator (NMSE_subk, BER_test_maMIMO_LTF.m:675-686), here against
    # the dataset's labels
    print('NMSE_subk of the DNN estimate vs labels = %.6e' % eng.nmse(lab, out_re + 1j * out_im))
    num = np.linalg.norm((h_ls - lab).reshape(npkt, -1), axis=1)
    print('LS(GPU) vs stored LS labels: max packet rel. error %.3e' % float(np.max(num / np.linalg.norm(lab.reshape(npkt, -1), axis=1))))
    files = ds.export_predictions(args.workdir, packed, out_re, out_im)
    for d in ('real', 'imag'):
        # DNN.py:411 CSI_predictor.save(<workdir>/<d>_keras_model): the folder inference.CSIPredictor loads
        models[d].save(os.path.join(args.workdir, d + '_keras_model'), pilot=packed['pilot'])
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
