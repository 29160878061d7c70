"""Twin of the reference's training loop (massiveMIMO_CSI_prediction_DNN.py --train, lines 272-319)
on top of the csi_train_* C-ABI: Adam + mse, AWGN on the LTF input at a random SNR per batch
(changeNoisePower, :86-102, SNR levels :303), EarlyStopping(patience 25, restore best weights) and
ReduceLROnPlateau(factor 0.1, patience 20, min_lr = lr*0.01) on val_loss (:285-286).

The data generators are anything indexable like the reference's DataGenerator: ``gen[b]`` returns
``(X, y, rms)`` with ``X = [ltf_rows [bs, lenLTF(,1)], pilot_rows [bs, nt]]`` (matlab_maMimo) or a
plain ``[bs, d_in]`` array; ``len(gen)`` is the number of batches; ``on_epoch_end()`` is called when
present (massiveMIMO_dataGenerator.py:264-316)."""
import numpy as np

from . import dist

SNR_LEVELS_MAMIMO = (30, 20, 10, 0, -10, -20)       # DNN.py:303


def rows_from_batch(X):
    """[ltf, P] (dataGenerator.py:314) or a plain array -> float32 rows [bs, d_in]."""
    if isinstance(X, (list, tuple)):
        ltf = np.asarray(X[0], np.float32)
        ltf = ltf.reshape(ltf.shape[0], -1)
        return np.concatenate([ltf, np.asarray(X[1], np.float32)], axis=1)
    X = np.asarray(X, np.float32)
    return X.reshape(X.shape[0], -1)


def average_signal_power(gen):
    """mean over the first mini-batch of rms(row)^2 of the LTF part (DNN.py:296-302)."""
    X, _, _ = gen[0]
    ltf = np.asarray(X[0] if isinstance(X, (list, tuple)) else X, np.float64)
    ltf = ltf.reshape(ltf.shape[0], -1)
    return float(np.mean(np.mean(ltf ** 2, axis=1)))


def noise_std_for(avg_sig_pow, snr_db):
    """stddev handed to the AWGN layer (DNN.py:97-100)."""
    return float(np.sqrt(avg_sig_pow / 10.0 ** (snr_db / 10.0)) / np.sqrt(2.0))


def evaluate(engine, model, gen, resident=False):
    """keras Model.evaluate: batch losses averaged with the batch sizes as weights."""
    tot, cnt = 0.0, 0
    for b in range(len(gen)):
        if resident:
            ids = gen.batch_ids(b)
            tot += engine.train_eval_indexed(model, ids) * len(ids)
            cnt += len(ids)
            continue
        X, y, _ = gen[b]
        rows = rows_from_batch(X)
        tot += engine.train_eval(model, rows, np.asarray(y, np.float32)) * rows.shape[0]
        cnt += rows.shape[0]
    return tot / max(cnt, 1)


def fit(engine, model, train_gen, val_gen, epochs=500, lr=1e-4, dropout=0.15, weights=None, method='default_SNR',
        snr_levels=SNR_LEVELS_MAMIMO, es_patience=25, rlr_patience=20, rlr_factor=0.1, rlr_min_delta=1e-4, min_lr=None, seed=0,
        verbose=True, commit=True, data_parallel=False, resident=None):
    """Trains one component model ('real' / 'imag') and returns the history dict
    {'loss': [...], 'val_loss': [...], 'lr': [...]}.  With commit the best weights (lowest val_loss,
    EarlyStopping restore_best_weights) become the engine's inference model.

    Callback semantics are keras 2.3's for the arguments the reference passes (DNN.py:285-286):
    EarlyStopping(patience 25) has min_delta 0 - an epoch improves when val_loss < best; ReduceLROnPlateau
    (factor 0.1, patience 20) keeps the keras DEFAULT min_delta = 1e-4 - an epoch improves only when
    val_loss < best - 1e-4, so at late-training loss levels the learning rate drops although the early
    stopping counter still sees improvements (``rlr_min_delta``).

    data_parallel (torch.distributed initialised, one process per GPU, every rank calling fit with its
    own shard of batches and the same seed / initial weights): each step is backward -> one flat
    gradient all-reduce (RCCL) -> Adam, so all ranks hold identical parameters; the validation loss is
    averaged over the ranks; BatchNormalization running statistics are averaged at the end.

    resident = the dataset dict (dataset.load_dataset): the training set is uploaded once
    (csi_train_set_dataset, engine.set_pilot must have been called) and every step sends only the batch's
    sample indices (``gen.batch_ids(b)``, dataset.SampleGenerator) - no per-step batch assembly or upload."""
    rng = np.random.default_rng(seed)
    min_lr = lr * 0.01 if min_lr is None else min_lr
    engine.train_begin(model, weights=weights, lr=lr, dropout=dropout, seed=seed)
    if resident is not None:
        from . import dataset as _ds
        engine.train_set_dataset(model, *_ds.resident_arrays(resident, model if isinstance(model, str) else ('real', 'imag')[model]))
    avg_pow = average_signal_power(train_gen) if method == 'default_SNR' else 0.0
    hist = {'loss': [], 'val_loss': [], 'lr': []}
    best, best_w, es_wait, rlr_best, rlr_wait, cur_lr = np.inf, None, 0, np.inf, 0, lr
    for ep in range(epochs):
        tot, cnt = 0.0, 0
        for b in range(len(train_gen)):
            std = noise_std_for(avg_pow, rng.choice(snr_levels)) if method == 'default_SNR' else 0.0
            if resident is not None:
                ids = train_gen.batch_ids(b)
                if data_parallel:
                    loss = engine.train_backward_indexed(model, ids, noise_std=std)
                    engine.synchronize()
                    dist.all_reduce_device(*engine.train_grads(model), average=True)
                    engine.train_apply(model)
                else:
                    loss = engine.train_step_indexed(model, ids, noise_std=std)
                tot += loss * len(ids)
                cnt += len(ids)
                continue
            X, y, _ = train_gen[b]
            rows = rows_from_batch(X)
            if data_parallel:
                loss = engine.train_backward(model, rows, np.asarray(y, np.float32), noise_std=std)
                engine.synchronize()
                dist.all_reduce_device(*engine.train_grads(model), average=True)
                engine.train_apply(model)
            else:
                loss = engine.train_step(model, rows, np.asarray(y, np.float32), noise_std=std)
            tot += loss * rows.shape[0]
            cnt += rows.shape[0]
        if hasattr(train_gen, 'on_epoch_end'):
            train_gen.on_epoch_end()
        val = evaluate(engine, model, val_gen, resident=resident is not None)
        if data_parallel:
            val = dist.all_reduce_sum(val) / dist.world_size()
        hist['loss'].append(tot / max(cnt, 1))
        hist['val_loss'].append(val)
        hist['lr'].append(cur_lr)
        if verbose:
            print(f'Epoch {ep + 1}/{epochs} - loss: {hist["loss"][-1]:.6g} - val_loss: {val:.6g} - lr: {cur_lr:.3g}')
        if val < best:
            best, best_w, es_wait = val, engine.train_weights(model), 0
            if data_parallel:
                bn = {k: v for k, v in best_w.items() if 'moving_' in k}
                best_w.update(dist.all_reduce_mean_arrays(bn))
        else:
            es_wait += 1
            if es_wait >= es_patience:
                if verbose:
                    print(f'Epoch {ep + 1}: early stopping')
                break
        # ReduceLROnPlateau keeps its own best / wait (keras resets wait to 0 after a reduction)
        if val < rlr_best - rlr_min_delta:
            rlr_best, rlr_wait = val, 0
        else:
            rlr_wait += 1
            if rlr_wait >= rlr_patience and cur_lr > min_lr:
                cur_lr = max(cur_lr * rlr_factor, min_lr)
                engine.train_set_lr(model, cur_lr)
                rlr_wait = 0
                if verbose:
                    print(f'Epoch {ep + 1}: ReduceLROnPlateau reducing learning rate to {cur_lr:.3g}')
    engine.train_end(model, commit=False)
    hist['best_val_loss'] = float(best)
    hist['weights'] = best_w
    if commit and best_w is not None:
        engine.load_weights(model, best_w)
    return hist
