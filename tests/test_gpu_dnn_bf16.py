"""GPU tests of the bf16 DNN path (BASELINE configs[2]; -m gpu, through the C-ABI, checked against the numpy oracle's bf16-operand
emulation).  Round 6: the register-blocked band kernel (csrc/band4_kernel_gen.py, "csi_band4_bf16": 4 waves x 512 registers, every
weight fragment against two row groups) that replaces csi_band8_bf16 where the staged form applies (32 <= Nt <= 64)."""
import numpy as np
import pytest

from conftest import rel_rows

pytestmark = pytest.mark.gpu


def _weights(oracle, seed, nt, hidden, use_bn=True, n_out=234):
    rng = np.random.default_rng(seed)
    d_in = 320 * nt + nt
    return (oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn),
            oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn))


# (nt, nr, packets, hidden, n_out): ragged last bands (M % 128 != 0), rows of a band that straddle three pair rows (nt = 40, 48), one and
# several column steps, the smallest K1 the kernel serves (256), outputs narrower than one 32-column tile group
BAND4_CASES = [(64, 4, 8, (1024, 1024), 234), (32, 4, 5, (256, 512), 234), (48, 2, 7, (512, 256), 234), (40, 3, 11, (384, 768), 234),
               (64, 4, 37, (1024, 1024), 234), (32, 1, 3, (256, 256), 52), (64, 2, 3, (512, 512), 200)]


@pytest.mark.parametrize('nt,nr,npkt,hidden,n_out', BAND4_CASES)
def test_register_blocked_band_kernel_bf16(pkg, oracle, nt, nr, npkt, hidden, n_out):
    """csi_band4_bf16 against (a) csi_band8_bf16 on the same operand buffers - same bf16 roundings of h1 / h2, fp32 sums in another order
    (the h2 fragments alternate between the halves), so equal to fp32 rounding - (b) the oracle's bf16-operand emulation
    (massiveMIMO_CSI_prediction_DNN.py:211-227 with h1, h2 and the weights rounded to bf16 once), (c) itself, run to run."""
    rng = np.random.default_rng(100 + nt + npkt)
    w_re, w_im = _weights(oracle, 60 + nt, nt, hidden, n_out=n_out)
    P = oracle.hadamard(nt) if nt & (nt - 1) == 0 else rng.choice([-1.0, 1.0], (nt, nt))
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden, n_out=n_out, dtype='bf16')
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    e.set_option('force_tile', 256)                           # the large-grid kernels regardless of the batch size
    assert e.get_option('band4') == 1, 'the register-blocked form is the default'
    n0 = e.get_option('band_launches')
    b_re, b_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 2 and e.get_option('band4_available') == 1
    e.set_option('band4', 0)
    a_re, a_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 4
    assert np.isfinite(b_re).all() and np.isfinite(b_im).all()
    assert rel_rows(b_re, a_re) < 2e-6 and rel_rows(b_im, a_im) < 2e-6, (rel_rows(b_re, a_re), rel_rows(b_im, a_im))
    r_re, r_im = oracle.predict_packets_bf16(ltf.astype(np.complex64), P, w_re, w_im)
    assert rel_rows(b_re, r_re) < 4e-3 and rel_rows(b_im, r_im) < 4e-3
    e.set_option('band4', 1)
    c_re, c_im = e.predict(ltf)
    assert np.array_equal(c_re, b_re) and np.array_equal(c_im, b_im)
    e.close()


def test_register_blocked_band_kernel_in_a_graph_and_at_full_size(pkg, oracle):
    """configs[2] at its size (Nt = 64, Nr = 4, 5000 packets = 10 000 bands, 40 rounds of workgroups) through csi_estimate_device as ONE
    hipGraph: replays bit-identical with the eager step, sampled packets against the oracle's bf16 emulation, band4 = 1 and 0 agree."""
    nt, nr, hidden, npkt = 64, 4, (1024, 1024), 5000
    w_re, w_im = _weights(oracle, 7, nt, hidden)
    P = pkg.synth.hadamard(nt)
    e = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(5, 0, npkt, d_re, d_im)
    o = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]
    n0 = e.get_option('band_launches')
    e.estimate_device(d_re, d_im, npkt, *o)
    e.synchronize()
    assert e.get_option('band_launches') == n0 + 2
    sel = [0, 1, 2499, 4998, 4999]
    take = lambda d: np.concatenate([d.download(p, 1) for p in sel])
    g_re, g_im = take(o[0]), take(o[1])
    ltf = take(d_re) + 1j * take(d_im)
    r_re, r_im = oracle.predict_packets_bf16(ltf.astype(np.complex64), P, w_re, w_im)
    assert rel_rows(g_re, r_re) < 4e-3 and rel_rows(g_im, r_im) < 4e-3
    e.set_option('use_graph', 1)
    for _ in range(4):
        e.estimate_device(d_re, d_im, npkt, *o)
    e.synchronize()
    assert e.get_option('graph_replays') >= 1
    assert np.array_equal(take(o[0]), g_re) and np.array_equal(take(o[1]), g_im)
    e.set_option('use_graph', 0)
    e.set_option('band4', 0)
    e.estimate_device(d_re, d_im, npkt, *o)
    e.synchronize()
    assert rel_rows(take(o[0]), g_re) < 2e-6 and rel_rows(take(o[1]), g_im) < 2e-6
    e.close()
