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
