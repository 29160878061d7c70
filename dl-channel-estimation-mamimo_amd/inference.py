"""Drop-in twin of the reference deployment wrapper ``inference.py::CSIPredictor``
(inference.py:6-68): same constructor arguments, ``load_model``, ``inference``,
``preprocess_data``, ``postprocess_data`` and the same error behaviour (message on stdout and
``sys.exit(-1)``), with the two keras models replaced by ``CSIModel`` objects that run on the
HIP library.  It adds the ``matlab_maMimo`` experiment, whose input is the packet tensor and
whose DNN runs on the packet-batched path with layer 0 shared across the Nt pairs."""
import json
import os
import sys
import numpy as np

from .engine import CsiEngine
from .model import CSIModel, load_weight_file, config_from_weights, WEIGHT_FILE, CONFIG_FILE


class CSIPredictor:

    def __init__(self, model_path, experiment='RICE_RENEW', verbose=False, device=0, pilot=None,
                 workspace_bytes=0, nr=None, pinned_results=False):
        self.path = model_path
        self.experiment = experiment
        self.verbose = verbose
        self.device = device
        self.workspace_bytes = workspace_bytes
        self.engine = None
        self._pilot = pilot
        self._nr = nr
        self._any_nr = False
        # matlab_maMimo: result arrays over recycled pinned buffers (fresh array objects as ever; the library then downloads
        # straight into them and assembles real + 1j*imag on the device - engine.PinnedPool, csi_estimate_c128)
        self.pinned_results = bool(pinned_results)
        self.model_real, self.model_imag = self.load_model()

    # inference.py:14-22
    def load_model(self):
        """``<model_path>/{real,imag}_keras_model`` - either the folder CSIModel.save writes (weights.safetensors +
        config.json) or the TF SavedModel directory the reference's test run leaves there (DNN.py:411), whose
        ``variables/`` bundle is read directly (keras_files.py).  A SavedModel carries neither the pilot matrix
        (pass ``pilot=`` or call set_pilot) nor the rx-antenna count: packets are independent per rx antenna, so
        without ``nr=`` any [nPkt, nRx, lenLTF] batch is accepted."""
        dirs = {d: os.path.join(self.path, d + '_keras_model') for d in ('real', 'imag')}
        weights, cfg = {}, None
        for d, p in dirs.items():
            if os.path.exists(os.path.join(p, WEIGHT_FILE)):
                weights[d] = load_weight_file(os.path.join(p, WEIGHT_FILE))
                with open(os.path.join(p, CONFIG_FILE)) as f:
                    c = json.load(f)
                cfg = cfg or c
            else:
                weights[d] = load_weight_file(p)            # SavedModel directory (raises if it is neither)
        if cfg is None:
            w = weights['real']
            d_in = int(w['fc_dense0.kernel'].shape[0])
            two_input = self.experiment == 'matlab_maMimo'
            if two_input and d_in % 321:
                print('[CSIPredictor] ERROR: the saved model has %d inputs, not 321*nTx (LTF samples + pilot row).' % d_in)
                sys.exit(-1)
            hidden, i = [], 0
            while f'fc_dense{i}.kernel' in w:
                hidden.append(int(w[f'fc_dense{i}.kernel'].shape[1]))
                i += 1
            cfg = dict(nt=d_in // 321 if two_input else 0, nr=self._nr or 1, len_ltf=d_in, hidden=hidden,
                       n_out=int(w['fc_regressor.kernel'].shape[1]), use_bn='bn0.gamma' in w, bn_eps=1e-3)
            self._any_nr = two_input and self._nr is None
        nt, nr = int(cfg['nt']), int(cfg.get('nr', 1))
        if nt > 0:
            shape = config_from_weights(weights['real'], nt)
        else:       # single-input model (DNN.py:180,234): no pilot input
            shape = dict(hidden=list(cfg['hidden']), n_out=int(cfg['n_out']), use_bn=bool(cfg['use_bn']))
        self.engine = CsiEngine(nt, nr, hidden=shape['hidden'], n_out=shape['n_out'], use_bn=shape['use_bn'],
                                bn_eps=float(cfg.get('bn_eps', 1e-3)), device=self.device,
                                workspace_bytes=self.workspace_bytes,
                                len_ltf=int(cfg['len_ltf']) if nt == 0 else None)
        models = {}
        for d in ('real', 'imag'):
            models[d] = CSIModel(self.engine, d).load_weights(weights[d])
        pilot = self._pilot if self._pilot is not None else weights['real'].get('pilot')
        if nt > 0 and pilot is not None:
            self.engine.set_pilot(pilot)
        if self.verbose:
            print('------- Real Model Summary -------')
            models['real'].summary()
            print('------- Imag Model Summary -------')
            models['imag'].summary()
        return models['real'], models['imag']

    def set_pilot(self, P):
        """P [nt,nt], row j = pilot sequence of tx j (dataset['P'][:, j])."""
        self.engine.set_pilot(P)

    # inference.py:24-32
    def inference(self, input_batch: np.ndarray):
        X = self.preprocess_data(input_batch)
        if self.experiment == 'matlab_maMimo':
            # X complex128 [npkt, nr, len_ltf]: both component models over all nt*nr pairs of each packet, dataset
            # sample order (mk.py:62) -> complex64 [npkt, nr, nt, n_out].  The X.real / X.imag split of :29-30 and
            # the ``real + 1j*imag`` of :31 happen inside the library's staging copies (csi_estimate_c128)
            if self._any_nr:        # engine built for one rx antenna per item: [nPkt, nRx, L] -> [nPkt*nRx, 1, L] and back
                npkt, nrx = X.shape[:2]
                out, _ = self.engine.estimate(X.reshape(npkt * nrx, 1, X.shape[2]), dnn=True, ls=False, pinned_results=self.pinned_results)
                out = out.reshape(npkt, nrx, *out.shape[2:])
            else:
                out, _ = self.engine.estimate(X, dnn=True, ls=False, pinned_results=self.pinned_results)
            return self.postprocess_data(out)
        else:
            bs = X.shape[0]   # assumes num. of samples in the first dimension
            output_real = self.model_real.predict(X.real, batch_size=bs)
            output_imag = self.model_imag.predict(X.imag, batch_size=bs)
        output_cmplx = output_real + 1j * output_imag   # create complex data
        return self.postprocess_data(output_cmplx)

    def ls_estimate(self, input_batch: np.ndarray):
        """LS pilot estimate of the same packets (helperMIMOChannelEstimate.m), complex64
        [npkt, nr, nt, 234]; matlab_maMimo only."""
        return self.estimate(input_batch, dnn=False)[1]

    def estimate(self, input_batch: np.ndarray, dnn=True, ls=True):
        """(DNN estimate, LS estimate) of the same packets with ONE upload of the preambles; matlab_maMimo only."""
        X = self.preprocess_data(input_batch)
        if self._any_nr:
            npkt, nrx = X.shape[:2]
            outs = self.engine.estimate(X.reshape(npkt * nrx, 1, X.shape[2]), dnn=dnn, ls=ls, pinned_results=self.pinned_results)
            return tuple(None if o is None else o.reshape(npkt, nrx, *o.shape[2:]) for o in outs)
        return self.engine.estimate(X, dnn=dnn, ls=ls, pinned_results=self.pinned_results)

    # inference.py:35-46
    def preprocess_data(self, input_batch):
        if self.experiment in ('RICE_RENEW', 'matlab_maMimo'):
            # we expect input_batch to be an array with type np.complex128
            if input_batch.dtype != np.complex128:
                print('[CSIPredictor] ERROR: Input batch must be of type np.complex128')
                sys.exit(-1)
            if self.experiment == 'matlab_maMimo':
                e = self.engine
                ok = input_batch.ndim == 3 and input_batch.shape[2] == e.len_ltf and (self._any_nr or input_batch.shape[1] == e.nr)
                if not ok:
                    print('[CSIPredictor] ERROR: Input batch must have shape [nPkt, %s, %d].' % ('nRx' if self._any_nr else e.nr, e.len_ltf))
                    sys.exit(-1)
            prep_data = input_batch
        else:
            print('[CSIPredictor] ERROR: Unknown experiment %r' % (self.experiment,))
            sys.exit(-1)
        return prep_data

    # inference.py:48-68
    def postprocess_data(self, output_batch):
        if self.experiment == 'RICE_RENEW':
            # re-insert null subcarriers and revert the FFT shift (FFT size 64 assumed)
            # bins 1..26 <- o[26:52], bins 38..63 <- o[0:26], everything else (DC, guards) zero: the
            # closed form of [0*6 | o[0:26] | 0 | o[26:52] | 0*5] followed by ifftshift(axis=1)
            if output_batch.shape[1] != 52:     # 52 = non-zero subcarriers (pilots+data)
                print('[CSIPredictor] ERROR: Output samples must have size 52 (assuming FFTLen = 64).')
                sys.exit(-1)
            postp_data = np.zeros((output_batch.shape[0], 64), dtype=np.result_type(output_batch.dtype, np.float64))
            postp_data[:, 1:27] = output_batch[:, 26:52]
            postp_data[:, 38:64] = output_batch[:, 0:26]
        else:
            postp_data = output_batch            # matlab_maMimo: the 234 data bins as they are
        return postp_data
