"""ctypes face of oracle/csi_oracle_c.c - the second, plain-C statement of the oracle.  TEST INFRASTRUCTURE ONLY.

Same function names and array conventions as oracle/csi_oracle.py (which see for the reference citations), so that a test can run
the two statements side by side; everything is float64 / complex128.  ``build()`` compiles the C file with gcc into
oracle/libcsi_oracle_c.so (git-ignored, travels to the GPU box with the snapshot like the product's own .so).  Only tests/,
__graft_entry__ and bench.py's cpu_baseline leg may import this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SOURCE = os.path.join(_HERE, 'csi_oracle_c.c')
LIBRARY = os.path.join(_HERE, 'libcsi_oracle_c.so')
N_DATA = 234
BN_EPS = 1e-3
_lib = None


def build(force=False):
    """gcc -O3 -std=c99 -shared -fPIC csi_oracle_c.c -lm -> libcsi_oracle_c.so (rebuilt when the source is newer)."""
    if force or not os.path.exists(LIBRARY) or os.path.getmtime(LIBRARY) < os.path.getmtime(SOURCE):
        tmp = LIBRARY + '.%d.tmp' % os.getpid()
        subprocess.check_call(['gcc', '-O3', '-std=c99', '-Wall', '-Werror', '-shared', '-fPIC', SOURCE, '-o', tmp, '-lm'])
        os.replace(tmp, LIBRARY)
    return LIBRARY


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIBRARY):
            build()
        lib = ctypes.CDLL(LIBRARY)
        dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
        dpp = ctypes.POINTER(dp)
        lib.oc_data_bins.argtypes, lib.oc_data_bins.restype = [ip], ctypes.c_int
        lib.oc_vht_ltf_256.argtypes, lib.oc_vht_ltf_256.restype = [dp], None
        lib.oc_ofdm_demod.argtypes, lib.oc_ofdm_demod.restype = [dp, dp, ctypes.c_long, ctypes.c_int, dp, dp], None
        lib.oc_ls_from_rxsym.argtypes, lib.oc_ls_from_rxsym.restype = [dp, dp, ctypes.c_long, ctypes.c_int, dp, dp, dp, dp], None
        lib.oc_ls_estimate.argtypes, lib.oc_ls_estimate.restype = [dp, dp, ctypes.c_long, ctypes.c_int, dp, dp, dp, dp], ctypes.c_int
        lib.oc_fc_forward.argtypes = [dp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ip, dpp, dpp, dpp, ctypes.c_double, dp, dp, ctypes.c_int, dp]
        lib.oc_fc_forward.restype = ctypes.c_int
        lib.oc_samples_from_packets.argtypes = [dp, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp]
        lib.oc_samples_from_packets.restype = None
        lib.oc_lmmse_ce.argtypes = [dp, dp, ctypes.c_int, ctypes.c_int, ctypes.c_int, dp, dp, ctypes.c_int, ctypes.c_double, dp, dp]
        lib.oc_lmmse_ce.restype = ctypes.c_int
        lib.oc_nmse_subk.argtypes, lib.oc_nmse_subk.restype = [dp, dp, dp, dp, ctypes.c_long, ctypes.c_int], ctypes.c_double
        _lib = lib
    return _lib


def _d(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def data_carrier_indices():
    out = np.zeros(256, dtype=np.int32)
    n = load().oc_data_bins(out.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    return out[:n].astype(np.int64)


def vht_ltf_256():
    out = np.empty(256)
    load().oc_vht_ltf_256(_p(out))
    return out


def ofdm_demod(ltf, nt):
    """ltf complex [..., 320 nt] -> rxsym complex [..., 234, nt]"""
    ltf = np.asarray(ltf)
    lead = ltf.shape[:-1]
    assert ltf.shape[-1] == 320 * nt
    re, im = _d(ltf.real).reshape(-1, 320 * nt), _d(ltf.imag).reshape(-1, 320 * nt)
    o_re, o_im = np.empty((re.shape[0], N_DATA, nt)), np.empty((re.shape[0], N_DATA, nt))
    load().oc_ofdm_demod(_p(re), _p(im), re.shape[0], nt, _p(o_re), _p(o_im))
    return (o_re + 1j * o_im).reshape(lead + (N_DATA, nt))


def ls_from_rxsym(rxsym, P):
    """rxsym complex [..., 234, nt], P [nt, nt] (row j = tx j) -> H complex [..., 234, nt] (bins x tx), as csi_oracle.ls_from_rxsym"""
    rxsym, P = np.asarray(rxsym), np.asarray(P)
    nt = P.shape[0]
    lead = rxsym.shape[:-2]
    re, im = _d(rxsym.real).reshape(-1, N_DATA, nt), _d(rxsym.imag).reshape(-1, N_DATA, nt)
    p_re = _d(P.real)
    p_im = _d(P.imag) if np.iscomplexobj(P) else None
    h_re, h_im = np.empty((re.shape[0], nt, N_DATA)), np.empty((re.shape[0], nt, N_DATA))
    load().oc_ls_from_rxsym(_p(re), _p(im), re.shape[0], nt, _p(p_re), _p(p_im) if p_im is not None else None, _p(h_re), _p(h_im))
    return np.swapaxes(h_re + 1j * h_im, -1, -2).reshape(lead + (N_DATA, nt))


def ls_estimate(ltf, P):
    """ltf complex [..., 320 nt] -> H complex [..., nt, 234]"""
    ltf, P = np.asarray(ltf), np.asarray(P)
    nt = P.shape[0]
    lead = ltf.shape[:-1]
    re, im = _d(ltf.real).reshape(-1, 320 * nt), _d(ltf.imag).reshape(-1, 320 * nt)
    p_re = _d(P.real)
    p_im = _d(P.imag) if np.iscomplexobj(P) else None
    h_re, h_im = np.empty((re.shape[0], nt, N_DATA)), np.empty((re.shape[0], nt, N_DATA))
    rc = load().oc_ls_estimate(_p(re), _p(im), re.shape[0], nt, _p(p_re), _p(p_im) if p_im is not None else None, _p(h_re), _p(h_im))
    if rc:
        raise MemoryError('oc_ls_estimate')
    return (h_re + 1j * h_im).reshape(lead + (nt, N_DATA))


def fc_forward(x, w):
    """One component model on flattened + concatenated rows x [B, d_in]; ``w`` as in csi_oracle.fc_forward (keras names)."""
    x = _d(x)
    n_hidden = 0
    while f'fc_dense{n_hidden}.kernel' in w:
        n_hidden += 1
    use_bn = 'bn0.gamma' in w
    kernels = [_d(w[f'fc_dense{i}.kernel']) for i in range(n_hidden)]
    biases = [_d(w[f'fc_dense{i}.bias']) for i in range(n_hidden)]
    bn = [_d(w[f'bn{i}.{k}']) for i in range(n_hidden) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')] if use_bn else []
    widths = np.array([k.shape[1] for k in kernels], dtype=np.int32)
    w_reg, b_reg = _d(w['fc_regressor.kernel']), _d(w['fc_regressor.bias'])
    dp = ctypes.POINTER(ctypes.c_double)
    arr = lambda xs: (dp * max(len(xs), 1))(*[_p(a) for a in xs])
    y = np.empty((x.shape[0], w_reg.shape[1]))
    rc = load().oc_fc_forward(_p(x), x.shape[0], x.shape[1], n_hidden, widths.ctypes.data_as(ctypes.POINTER(ctypes.c_int)),
                              arr(kernels), arr(biases), arr(bn) if use_bn else None, float(w.get('bn_eps', BN_EPS)),
                              _p(w_reg), _p(b_reg), w_reg.shape[1], _p(y))
    if rc:
        raise MemoryError('oc_fc_forward')
    return y


def samples_from_packets(ltf, P, d):
    ltf = np.asarray(ltf)
    npkt, nr, len_ltf = ltf.shape
    P = _d(P)
    nt = P.shape[0]
    part = _d(ltf.real if d == 'real' else ltf.imag)
    x = np.empty((npkt * nr * nt, len_ltf + nt))
    load().oc_samples_from_packets(_p(part), npkt, nr, nt, len_ltf, _p(P), _p(x))
    return x


def predict_packets(ltf, P, w_real, w_imag):
    """The reference's test loop (DNN.py:339-346) through the literal network: (out_real, out_imag) [Npkt, Nr, Nt, 234]"""
    ltf = np.asarray(ltf)
    npkt, nr, _ = ltf.shape
    nt = np.asarray(P).shape[0]
    outs = []
    for d, w in (('real', w_real), ('imag', w_imag)):
        y = fc_forward(samples_from_packets(ltf, P, d), w)
        outs.append(y.reshape(npkt, nr, nt, -1))
    return outs[0], outs[1]


def lmmse_ce(h_tilde, nfft, np_, nps, h, snr_db):
    """LMMSE_ce.m:23-39 for one link, as csi_oracle.lmmse_ce"""
    lib = load()
    ht = np.asarray(h_tilde, dtype=np.complex128).reshape(-1)
    h = np.asarray(h, dtype=np.complex128).reshape(-1)
    t_re, t_im, h_re, h_im = _d(ht.real), _d(ht.imag), _d(h.real), _d(h.imag)
    o_re, o_im = np.empty(nfft), np.empty(nfft)
    rc = lib.oc_lmmse_ce(_p(t_re), _p(t_im), nfft, np_, nps, _p(h_re), _p(h_im), h.size, float(snr_db), _p(o_re), _p(o_im))
    if rc:
        raise RuntimeError('oc_lmmse_ce: %d' % rc)
    return o_re + 1j * o_im


def lmmse_estimate(h_ls, h, snr_db):
    """helperMIMOChannelEstimate.m:33-39 with isMMSE, as csi_oracle.lmmse_estimate"""
    h_ls = np.asarray(h_ls)
    npkt, nr, nt, n = h_ls.shape
    out = np.empty(h_ls.shape, dtype=np.complex128)
    for p in range(npkt):
        for i in range(nr):
            for j in range(nt):
                out[p, i, j] = lmmse_ce(h_ls[p, i, j], n, n, 1, h[p], snr_db[p, i])
    return out


def nmse_subk(h_ref, h_est):
    """NMSE_subk (BER_test_maMIMO_LTF.m:675-686), as csi_oracle.nmse_subk"""
    lib = load()
    r = np.asarray(h_ref, dtype=np.complex128)
    e = np.asarray(h_est, dtype=np.complex128)
    n_bins = r.shape[-1]
    r_re, r_im, e_re, e_im = _d(r.real), _d(r.imag), _d(e.real), _d(e.imag)
    return float(lib.oc_nmse_subk(_p(r_re), _p(r_im), _p(e_re), _p(e_im), r.size // n_bins, n_bins))
