"""Weight container and the ``tf.keras.Model``-shaped object the reference script drives
(massiveMIMO_CSI_prediction_DNN.py:231-234 build, :334 load_weights, :346 predict, :411 save).

Weight container (the build's own format; the reference's Keras HDF5 checkpoints and SavedModel
directories are read as well, keras_files.py):
a ``.safetensors`` (or torch ``.pt``) file holding, per component model d in {real, imag},
  fc_dense{i}.kernel [in,out]  fc_dense{i}.bias [out]
  bn{i}.gamma / .beta / .moving_mean / .moving_variance [out]        (when --useBN)
  fc_regressor.kernel [in,n_out]  fc_regressor.bias [n_out]
plus a ``config.json`` next to it (nt, nr, hidden, n_out, use_bn, bn_eps).  Layer order = the
keras layer order, which is how the reference matches tensors (load_weights by topology)."""
import json
import os
import numpy as np

from .engine import CsiEngine, N_DATA, SYM_LEN
from ._lib import CsiError

WEIGHT_FILE = 'weights.safetensors'
CONFIG_FILE = 'config.json'


def save_weight_file(path, weights, component=None):
    """weights: dict name -> float32 ndarray.  Format by extension: .safetensors | .pt | .npz | .hdf5 / .h5 - the
    last is a Keras HDF5 weights file in the layout of the reference's own checkpoints (DNN.py:279-281,319), which
    keras ``load_weights`` reads by topology (keras_files.write_keras_hdf5_weights); ``component`` ('real' / 'imag',
    default: from the file name) only selects keras' auto-numbering of the layer names."""
    tensors = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in weights.items() if isinstance(v, np.ndarray)}
    if path.endswith(('.hdf5', '.h5')):
        from .keras_files import write_keras_hdf5_weights, keras_layers_from_weights
        if component is None:
            component = 'imag' if os.path.basename(path).startswith('imag') else 'real'
        write_keras_hdf5_weights(path, keras_layers_from_weights({k: v for k, v in tensors.items() if k != 'pilot'}, component))
        return
    if path.endswith('.npz'):
        np.savez(path, **tensors)
    elif path.endswith('.pt'):
        import torch
        torch.save({k: torch.from_numpy(v) for k, v in tensors.items()}, path)
    else:
        from safetensors.numpy import save_file
        save_file(tensors, path)


def normalize_keras_names(tensors):
    """Accept the variable names Keras itself uses and map them to the container's names:
        'fc_dense0/kernel:0', 'fc_regressor/bias:0'                       (model.weights[i].name)
        'fc_dense0/fc_dense0/kernel:0'                                    (HDF5 group/dataset paths)
        'batch_normalization_7/moving_mean:0'                             (auto-numbered BatchNormalization)
    BatchNormalization layers are matched by ORDER (their numeric suffixes sorted), which is how the
    reference itself pairs them (load_weights by topology, DNN.py:334): the n-th one becomes bn{n}.
    Names already in container form pass through."""
    import re
    out, bn = {}, {}
    for name, val in tensors.items():
        n = name.split(':')[0]
        parts = n.split('/')
        if len(parts) >= 2:
            layer, var = parts[-2], parts[-1]
        elif '.' in n:
            layer, var = n.rsplit('.', 1)
        else:
            out[name] = val
            continue
        m = re.fullmatch(r'batch_normalization(?:_(\d+))?', layer)
        if m:
            bn.setdefault(int(m.group(1) or 0), {})[var] = val
        else:
            out[f'{layer}.{var}'] = val
    for i, key in enumerate(sorted(bn)):
        for var, val in bn[key].items():
            out[f'bn{i}.{var}'] = val
    return out


def load_weight_file(path):
    """.safetensors | .pt | .npz (np.savez(path, **{v.name: v.numpy() for v in keras_model.weights}) on a
    Keras host needs nothing but numpy) | .hdf5 / .h5 - the reference's own checkpoint files
    (DNN.py:279-281,334) and whole-model .h5 files | a TF SavedModel directory (DNN.py:411, inference.py:15-16).
    The last two are read by keras_files.py without h5py / TensorFlow; Keras variable names are normalised."""
    if os.path.isdir(path):
        from .keras_files import read_savedmodel_variables
        return normalize_keras_names(read_savedmodel_variables(path))
    if path.endswith(('.hdf5', '.h5')):
        from .keras_files import read_keras_hdf5_weights
        return normalize_keras_names(read_keras_hdf5_weights(path))
    if path.endswith('.npz'):
        with np.load(path) as z:
            return normalize_keras_names({k: np.asarray(z[k], dtype=np.float32) for k in z.files})
    if path.endswith('.pt'):
        import torch
        return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in torch.load(path, map_location='cpu').items()}
    from safetensors.numpy import load_file
    return {k: np.asarray(v, dtype=np.float32) for k, v in load_file(path).items()}


def config_from_weights(weights, nt):
    """Derive hidden widths / n_out / use_bn from tensor shapes (the reference derives the model
    shape from --nn, --useBN and the dataset, massiveMIMO_dataGenerator.py:26-38)."""
    hidden = []
    i = 0
    while f'fc_dense{i}.kernel' in weights:
        hidden.append(int(weights[f'fc_dense{i}.kernel'].shape[1]))
        i += 1
    d_in = int(weights['fc_dense0.kernel'].shape[0])
    if d_in != SYM_LEN * nt + nt:
        raise CsiError(-1, f'fc_dense0.kernel has {d_in} rows, expected {SYM_LEN * nt + nt} for nt={nt}')
    return dict(hidden=hidden, n_out=int(weights['fc_regressor.kernel'].shape[1]), use_bn='bn0.gamma' in weights)


class CSIModel:
    """One component regressor ('real' or 'imag') with the keras call surface used by the
    reference: ``load_weights``, ``predict``, ``save``, ``summary``.

    ``predict`` has Keras semantics - arbitrary rows in, one output row per input row, no
    sharing of layer 0 between rows (csi_predict_samples).  The packet-batched fast path is
    ``CSIPredictor.inference`` / ``CsiEngine.predict``."""

    def __init__(self, engine: CsiEngine, component):
        assert component in ('real', 'imag')
        self.engine = engine
        self.d = component
        self._weights = None

    # DNN.py:334  CSI_predictor.load_weights(model_filepath)
    def load_weights(self, path_or_dict):
        w = load_weight_file(path_or_dict) if isinstance(path_or_dict, str) else dict(path_or_dict)
        self.engine.load_weights(self.d, w)
        self._weights = w
        return self

    # DNN.py:346 predict(x=generator) ; :434/:470 predict([Xsig, Xp], batch_size=...) ;
    # inference.py:29 predict(X.real, batch_size=bs)
    def predict(self, x, batch_size=None, verbose=0):
        if hasattr(x, '__getitem__') and hasattr(x, '__len__') and not isinstance(x, (list, tuple, np.ndarray)):
            # a keras Sequence: batches ([Xsig, Xp], y, rms_fact) in index order (gen.py:241-252)
            outs = [self.predict(x[b][0]) for b in range(len(x))]
            return np.concatenate(outs, axis=0) if outs else np.empty((0, self.engine.n_out), np.float32)
        if isinstance(x, (list, tuple)):
            xsig, xp = x
            xsig = np.asarray(xsig)
            flat = xsig.reshape(xsig.shape[0], -1)                      # Flatten, DNN.py:207
            x = np.concatenate([flat, np.asarray(xp)], axis=1)          # Concatenate(axis=1), :208
        return self.engine.predict_samples(self.d, np.asarray(x, dtype=np.float32))

    # DNN.py:411  CSI_predictor.save(<workdir>/<d>_keras_model)
    def save(self, model_dir, pilot=None):
        if self._weights is None:
            raise CsiError(-2, 'no weights loaded')
        os.makedirs(model_dir, exist_ok=True)
        w = dict(self._weights)
        if pilot is not None:
            w['pilot'] = np.asarray(pilot, dtype=np.float32)
        save_weight_file(os.path.join(model_dir, WEIGHT_FILE), w)
        e = self.engine
        with open(os.path.join(model_dir, CONFIG_FILE), 'w') as f:
            json.dump(dict(component=self.d, nt=e.nt, nr=e.nr, len_ltf=e.len_ltf, hidden=list(e.hidden),
                           n_out=e.n_out, use_bn=e.use_bn, bn_eps=1e-3, datasource='matlab_maMimo'), f, indent=1)

    def summary(self, print_fn=print):
        e = self.engine
        print_fn(f'Model: "{self.d}"  (FC regressor, massiveMIMO_CSI_prediction_DNN.py:176-234)')
        print_fn(f' input_1 (None, {e.len_ltf}, 1)   input_2 (None, {e.nt})   concatenate (None, {e.d_in})')
        fan, total = e.d_in, 0
        for i, h in enumerate(e.hidden):
            n = fan * h + h
            total += n
            print_fn(f' fc_dense{i} (Dense relu)        (None, {h})   params {n}')
            if e.use_bn:
                total += 4 * h
                print_fn(f' batch_normalization_{i}        (None, {h})   params {4 * h}')
            fan = h
        n = fan * e.n_out + e.n_out
        total += n
        print_fn(f' fc_regressor (Dense linear)   (None, {e.n_out})   params {n}')
        print_fn(f'Total params: {total}')
