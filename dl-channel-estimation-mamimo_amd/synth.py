"""Synthetic inputs for benchmarks and smoke runs.  The reference ships neither trained
weights nor datasets (README.md:17-18), so throughput is measured on random-initialised
weights of the shipped architecture and synthetic packets of the shipped shape."""
import numpy as np

SYM_LEN = 320


def hadamard(n):
    """Sylvester-Hadamard pilot mapping matrix; stand-in for the un-vendored helperGetP
    (helperMIMOChannelEstimate.m:13).  In the real pipeline P arrives with the dataset."""
    if n < 1 or n & (n - 1):
        raise ValueError('hadamard(n) needs a power of two')
    h = np.ones((1, 1), dtype=np.float32)
    while h.shape[0] < n:
        h = np.block([[h, h], [h, -h]])
    return h.astype(np.float32)


def make_weights(rng, nt, hidden=(1024, 1024), n_out=234, use_bn=True):
    """Random weights with the keras initialisers of the reference model
    (glorot_uniform kernels, DNN.py:213,227) and non-trivial BatchNormalization statistics."""
    w = {}
    fan_in = SYM_LEN * nt + nt
    for i, h in enumerate(hidden):
        lim = np.sqrt(6.0 / (fan_in + h))
        w[f'fc_dense{i}.kernel'] = rng.uniform(-lim, lim, (fan_in, h)).astype(np.float32)
        w[f'fc_dense{i}.bias'] = (0.01 * rng.standard_normal(h)).astype(np.float32)
        if use_bn:
            w[f'bn{i}.gamma'] = rng.uniform(0.5, 1.5, h).astype(np.float32)
            w[f'bn{i}.beta'] = (0.1 * rng.standard_normal(h)).astype(np.float32)
            w[f'bn{i}.moving_mean'] = (0.1 * rng.standard_normal(h)).astype(np.float32)
            w[f'bn{i}.moving_variance'] = rng.uniform(0.5, 1.5, h).astype(np.float32)
        fan_in = h
    lim = np.sqrt(6.0 / (fan_in + n_out))
    w['fc_regressor.kernel'] = rng.uniform(-lim, lim, (fan_in, n_out)).astype(np.float32)
    w['fc_regressor.bias'] = (0.01 * rng.standard_normal(n_out)).astype(np.float32)
    return w


def white_packets(rng, npkt, nr, nt):
    """i.i.d. CN(0,1) preambles, complex64 [npkt, nr, 320*nt] (host twin of csi_synth_white's
    distribution, not of its stream)."""
    shape = (npkt, nr, SYM_LEN * nt)
    return ((rng.standard_normal(shape) + 1j * rng.standard_normal(shape)) / np.sqrt(2.0)).astype(np.complex64)


# ---- structured sounding packets (the synthetic twin of generate_maMIMO_LTF.m:197-342) -----------
FFT_LEN, CP_LEN = 256, 64
SNR_LEVELS_DB = (-25, -20, -15, -10, -5, 0, 5, 10)      # setenv.sh:19-25 (SNRLev), 500 test packets per level
AMP_SCALE = np.sqrt(FFT_LEN - 14) / FFT_LEN             # generate_maMIMO_LTF.m:303-304: sqrt(FFTLength - #nulls) / FFTLength


def vht_ltf_sequence():
    """256-bin VHT-LTF frequency sequence of helperMIMOChannelEstimate.m:16-23, fftshift-ed order
    (index 0 = most negative frequency); 0 on the 7 + 1 + 6 null bins of generate_maMIMO_LTF.m:99."""
    left = [1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1, 1, 1, -1, -1, 1, 1, -1, 1, -1, 1, 1, 1, 1]
    right = [1, -1, -1, 1, 1, -1, 1, -1, 1, -1, -1, -1, -1, -1, 1, 1, -1, -1, 1, -1, 1, -1, 1, 1, 1, 1]
    mid_a = [-1, -1, -1, 1, 1, -1, 1, -1, 1, 1, -1]
    mid_b = [1, -1, 1, -1, 0, 1, -1, -1, 1]
    seg = left + [1] + right
    seq = [0] * 7 + seg + mid_a + seg + mid_b + seg + mid_a + seg + [0] * 6
    assert len(seq) == FFT_LEN
    return np.asarray(seq, dtype=np.float32)


def structured_packets(rng, npkt, nr, P, snr_db, n_taps=8):
    """Sounding packets as the reference's simulator hands them to the estimators: every (rx, tx) link
    an n_taps complex Gaussian impulse response, the Nt LTF symbols mapped by P, OFDM-modulated with
    the 64-sample cyclic prefix (generate_maMIMO_LTF.m:202,210), complex AWGN at `snr_db` relative to
    the received preamble power (:283-295; `snr_db` scalar or one value per packet), then the
    sub-carrier power scaling of :303-304 applied to signal AND noise, as there.  The signal level is
    the same at every SNR - only the noise moves - so a batch that mixes levels spans the amplitude
    range the reference's test set spans.  Returns complex64 [npkt, nr, 320*nt]."""
    P = np.asarray(P, dtype=np.float32)
    nt = P.shape[0]
    snr = np.broadcast_to(np.asarray(snr_db, dtype=np.float64), (npkt,))
    decay = (np.exp(-0.5 * np.arange(n_taps)) / np.sqrt(2.0)).astype(np.float32)
    cir = np.zeros((npkt * nr, nt, FFT_LEN), dtype=np.complex64)
    cir.real[..., :n_taps] = rng.standard_normal((npkt * nr, nt, n_taps), dtype=np.float32) * decay
    cir.imag[..., :n_taps] = rng.standard_normal((npkt * nr, nt, n_taps), dtype=np.float32) * decay
    h = np.fft.fft(cir, axis=-1)                                           # [pr, j, bin], un-shifted bin order
    # frequency-domain LTF symbols X[pr, s, k] = ltf[k] * sum_j H[pr, j, k] P[j, s]
    xf = np.matmul(np.ascontiguousarray(P.T).astype(np.complex64), h)      # [pr, s, bin]
    xf *= np.fft.ifftshift(vht_ltf_sequence())[None, None, :]
    xt = np.fft.ifft(xf, axis=-1)                                          # [pr, s, n]
    ltf = np.empty((npkt * nr, nt, SYM_LEN), dtype=np.complex64)
    ltf[..., :CP_LEN] = xt[..., -CP_LEN:]                                  # cyclic prefix
    ltf[..., CP_LEN:] = xt
    ltf = ltf.reshape(npkt, nr, nt * SYM_LEN)
    sig_pow = float(np.mean(ltf.real ** 2 + ltf.imag ** 2))
    nstd = np.sqrt(sig_pow / (10.0 ** (snr / 10.0)) / 2.0).astype(np.float32)       # per packet, per real component
    for part in (ltf.real, ltf.imag):
        part += rng.standard_normal(ltf.shape, dtype=np.float32) * nstd[:, None, None]
    ltf *= np.float32(AMP_SCALE)
    return ltf


def mixed_snr_jobs(seed, per_level=500, levels=SNR_LEVELS_DB, block=250):
    """BASELINE config 2's defining input: `per_level` test packets at EACH of the pipeline's SNR levels
    (setenv.sh:19-25, full_pipeline_maMIMO_DNNEst.sh:44-48), level after level (lowest SNR first), so
    that ONE launch sees the whole 35 dB spread.  The batch is cut into blocks of <= `block` packets with
    one independent random stream each (SeedSequence children of `seed`), so that blocks can be produced
    in any order / in parallel and any block can be regenerated alone (the parity checks do that).
    Returns [(first_packet, n_packets, snr_db, SeedSequence)]."""
    jobs, first = [], 0
    for lv in levels:
        left = per_level
        while left > 0:
            n = min(block, left)
            jobs.append([first, n, float(lv)])
            first += n
            left -= n
    for job, ss in zip(jobs, np.random.SeedSequence(seed).spawn(len(jobs))):
        job.append(ss)
    return [tuple(j) for j in jobs]


def mixed_snr_block(job, nr, P):
    """The packets of one job of mixed_snr_jobs: complex64 [n, nr, 320*nt]."""
    first, n, snr_db, ss = job
    return structured_packets(np.random.default_rng(ss), n, nr, P, snr_db)


def mixed_snr_batch(seed, nr, P, per_level=500, levels=SNR_LEVELS_DB, block=250, threads=2):
    """Generator over the blocks of the mixed-SNR batch, produced by a small thread pool (numpy's FFT and
    normal generator release the GIL): yields (first_packet, snr_db, complex64 [n, nr, 320*nt]) in order."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = mixed_snr_jobs(seed, per_level, levels, block)
    with ThreadPoolExecutor(max_workers=max(1, threads)) as pool:
        pending = []
        it = iter(jobs)
        for job in it:
            pending.append((job, pool.submit(mixed_snr_block, job, nr, P)))
            if len(pending) >= max(1, threads):
                j, f = pending.pop(0)
                yield j[0], j[2], f.result()
        for j, f in pending:
            yield j[0], j[2], f.result()
