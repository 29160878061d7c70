"""Data formats on either side of the hot path.

IN  - the pickle dataset written by create_massiveMIMO_CSIest_dnn_dataset.py:125 and read by
      massiveMIMO_dataGenerator.py:22-55:
        {'X': int [N,2] (LTF key, iTx), 'y': {'real','imag'} [N,234], 'LTF': {key: {'real','imag'}
         [lenLTF]}, 'P': [Nt,Nt] (h5py-transposed MATLAB matrix), 'simParams': {...}}
      with N = Npkt*Nr*Nt and sample index s = p*Nr*Nt + iRx*Nt + iTx (mk.py:62).  The hot path
      wants it packed: preambles [Npkt,Nr,lenLTF] (each rx preamble ONCE - the dataset already
      de-duplicates them by key, mk.py:50-63), pilot rows [Nt,Nt], labels [Npkt,Nr,Nt,234].
OUT - the per-packet .mat files the MATLAB evaluation reads (massiveMIMO_CSI_prediction_DNN.py:
      401-409 writes them, BER_test_maMIMO_LTF.m:197-217 reads them):
        test_csi_predictions_<d>_<n>.mat  with struct all_pkts_csi_nn_out {x, y, true_y},
        rows (iRX-1)*nTX + iTX of packet n."""
import os
import pickle

import numpy as np


def load_dataset(path):
    """Unpickle a dataset file (massiveMIMO_dataGenerator.py:22-25)."""
    with open(path, 'rb') as f:
        return pickle.load(f)


def packets_from_dataset(ds):
    """Pack a dataset dict for the hot path.  Returns a dict with
        ltf      complex128 [Npkt, Nr, lenLTF]
        pilot    float64 [Nt, Nt], row t = ds['P'][:, t]  (what the DNN sees for tx t, gen.py:311)
        labels   complex128 [Npkt, Nr, Nt, nSubCarr]  (the LS estimates stored as training labels)
        nt, nr, npkt
    and verifies the structure the reference relies on (sample order, one LTF key per
    (packet, rx), iTx running 0..Nt-1)."""
    X = np.asarray(ds['X'])
    sim = ds['simParams']
    nt, nr = int(sim['nTX']), int(sim['nRX'])
    n = X.shape[0]
    if n % (nt * nr):
        raise ValueError('number of samples %d is not a multiple of nTX*nRX = %d' % (n, nt * nr))   # gen.py:47-50
    npkt = n // (nt * nr)
    keys = X[:, 0].reshape(npkt, nr, nt)
    itx = X[:, 1].reshape(npkt, nr, nt)
    if not (keys == keys[:, :, :1]).all():
        raise ValueError('samples of one (packet, rx) do not share one LTF key: not in dataset order (mk.py:62)')
    if not (itx == np.arange(nt)[None, None, :]).all():
        raise ValueError('iTx does not run 0..nTX-1 inside each (packet, rx): not in dataset order (mk.py:62)')
    first = ds['LTF'][int(keys[0, 0, 0])]
    len_ltf = int(np.asarray(first['real']).shape[0])
    ltf = np.empty((npkt, nr, len_ltf), dtype=np.complex128)
    for p in range(npkt):
        for r in range(nr):
            e = ds['LTF'][int(keys[p, r, 0])]
            ltf[p, r] = np.asarray(e['real']) + 1j * np.asarray(e['imag'])
    y = np.asarray(ds['y']['real']) + 1j * np.asarray(ds['y']['imag'])
    return dict(ltf=ltf, pilot=np.ascontiguousarray(np.asarray(ds['P'], dtype=np.float64).T),
                labels=y.reshape(npkt, nr, nt, -1), nt=nt, nr=nr, npkt=npkt)


def label_consistency(engine, packed):
    """the stored labels are the LS estimate of the very same noisy preamble
    (generate_maMIMO_LTF.m:326-354), so LS(ltf) must reproduce them.  Runs the LS kernel and
    returns the max norm-relative row error."""
    engine.set_pilot(packed['pilot'])
    h = engine.ls_estimate(packed['ltf'])
    ref = packed['labels']
    a = np.concatenate([h.real, h.imag], -1).reshape(-1, 2 * ref.shape[-1]).astype(np.float64)
    b = np.concatenate([ref.real, ref.imag], -1).reshape(a.shape)
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.linalg.norm(b, axis=1)))


def export_packet_mats(workdir, d, x_rows, y_pred, y_true, nt, nr):
    """Write test_csi_predictions_<d>_<n>.mat for n = 1..Npkt as DNN.py:401-409 does.
    x_rows [Npkt*Nr*Nt, lenLTF] (the LTF part of the DNN input), y_pred / y_true [Npkt*Nr*Nt, 234],
    all in dataset sample order.  Returns the list of files."""
    from scipy.io import savemat
    bs = nt * nr
    n = y_pred.shape[0]
    assert n % bs == 0 and x_rows.shape[0] == n and y_true.shape[0] == n
    os.makedirs(workdir, exist_ok=True)
    files = []
    for h in range(0, n, bs):
        mat_out = {'all_pkts_csi_nn_out': dict(x=x_rows[h:h + bs], y=y_pred[h:h + bs], true_y=y_true[h:h + bs])}
        f = os.path.join(workdir, 'test_csi_predictions_' + d + '_' + str(h // bs + 1) + '.mat')
        savemat(f, mat_out, do_compression=True)
        files.append(f)
    return files


def export_predictions(workdir, packed, out_real, out_imag):
    """Convenience wrapper: packed dataset + the two float32 output planes [Npkt,Nr,Nt,234] of
    CsiEngine.predict -> both families of .mat files."""
    npkt, nr, nt = packed['npkt'], packed['nr'], packed['nt']
    files = {}
    for d, out in (('real', out_real), ('imag', out_imag)):
        part = packed['ltf'].real if d == 'real' else packed['ltf'].imag
        x_rows = np.repeat(part[:, :, None, :], nt, axis=2).reshape(npkt * nr * nt, -1)
        lab = packed['labels'].real if d == 'real' else packed['labels'].imag
        files[d] = export_packet_mats(workdir, d, x_rows, np.asarray(out).reshape(npkt * nr * nt, -1),
                                      lab.reshape(npkt * nr * nt, -1), nt, nr)
    return files


# ------------------------------------------------------------------------------------------------
# Training-side twins: the sample split of loadDataset and the batch generator
# the reference hands to Model.fit.
def split_train_val(ds, val_train_ratio=0.15):
    """(train_ids, val_ids) as massiveMIMO_dataGenerator.py:46-55 + DNN.py:124-127 build them: the
    last floor(Npkt * ratio) PACKETS (whole packets, nTX*nRX samples each) are the validation set."""
    sim = ds['simParams']
    per_pkt = int(sim['nTX']) * int(sim['nRX'])
    n = int(np.asarray(ds['X']).shape[0])
    if n % per_pkt:
        print('Num. of packets is not an integer. Please double check --nTX and --nRX arguments to match the provided dataset. Aborting...')
        raise SystemExit(-1)                                                   # gen.py:47-50
    n_val = int(np.floor((n // per_pkt) * val_train_ratio)) * per_pkt
    return list(range(n - n_val)), list(range(n - n_val, n))


class SampleGenerator:
    """Twin of the reference's DataGenerator for datasource 'matlab_maMimo', method 'default' /
    'default_SNR' (massiveMIMO_dataGenerator.py:213-316): ``gen[b] -> ([Xsig [bs,lenLTF,1], Xp [bs,nTX]], y [bs,nSubCarr], None)``
    with Xsig = the rx preamble of the sample's LTF key (component ``d``), Xp = dataset['P'][:, iTx];
    floor(len(ids)/bs) batches per epoch; indexes reshuffled by on_epoch_end() when shuffle is set."""

    def __init__(self, list_ids, ds, d, batch_size=256, shuffle=True, seed=None):
        self.list_IDs = np.asarray(list(list_ids), dtype=np.int64)
        self.dataset = ds
        self.d = d
        self.batch_size = int(batch_size)
        self.shuffle = shuffle
        self._rng = np.random.default_rng(seed)
        X = np.asarray(ds['X'])
        self._P = np.ascontiguousarray(np.asarray(ds['P'], dtype=np.float32).T)      # row t = ds['P'][:, t]
        self._y = np.asarray(ds['y'][d], dtype=np.float32)
        self._itx = X[:, 1].astype(np.int64)
        # every rx preamble once, as one float32 table: a batch is then two fancy-indexing gathers instead
        # of bs dictionary look-ups and row copies (the reference's loop, massiveMIMO_dataGenerator.py:299-312)
        keys = sorted({int(k) for k in X[self.list_IDs, 0]}) if len(self.list_IDs) else []
        self._ltf = (np.stack([np.asarray(ds['LTF'][k][d], dtype=np.float32) for k in keys]) if keys
                     else np.zeros((0, 0), np.float32))
        row_of_key = {k: i for i, k in enumerate(keys)}
        self._ltf_row = np.full(X.shape[0], -1, dtype=np.int64)
        self._ltf_row[self.list_IDs] = [row_of_key[int(k)] for k in X[self.list_IDs, 0]]
        self.on_epoch_end()

    def __len__(self):
        return int(np.floor(len(self.list_IDs) / self.batch_size))

    def __getitem__(self, index):
        idx = self.indexes[index * self.batch_size:(index + 1) * self.batch_size]
        ids = self.list_IDs[idx]
        xsig = self._ltf[self._ltf_row[ids]][:, :, None]
        xp = self._P[self._itx[ids]]
        return [xsig, xp], self._y[ids], None

    def batch_ids(self, index):
        """Dataset sample indices of batch ``index`` (the resident-dataset training path sends only these)."""
        return self.list_IDs[self.indexes[index * self.batch_size:(index + 1) * self.batch_size]]

    def reorder_indexes(self):
        self.indexes = np.arange(len(self.list_IDs))

    def set_batchsize(self, bs):
        self.batch_size = int(bs)

    def get_batchsize(self):
        return self.batch_size

    def on_epoch_end(self):
        self.indexes = np.arange(len(self.list_IDs))
        if self.shuffle:
            self._rng.shuffle(self.indexes)


def resident_arrays(ds, d):
    """The dataset in the form csi_train_set_dataset takes: (ltf_table [n_keys, lenLTF] float32 - every rx
    preamble of component ``d`` once, ltf_row [N], itx [N], y [N, nSubCarr]) for ALL samples of the pickle,
    so that training and validation generators can both address it by sample index."""
    X = np.asarray(ds['X'])
    keys = sorted({int(k) for k in X[:, 0]})
    row_of_key = {k: i for i, k in enumerate(keys)}
    table = np.stack([np.asarray(ds['LTF'][k][d], dtype=np.float32) for k in keys])
    ltf_row = np.array([row_of_key[int(k)] for k in X[:, 0]], dtype=np.int32)
    return table, ltf_row, X[:, 1].astype(np.int32), np.asarray(ds['y'][d], dtype=np.float32)
