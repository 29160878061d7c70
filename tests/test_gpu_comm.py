"""GPU tests (-m gpu; every call through the C-ABI of libcsi_mamimo.so, checked against the numpy oracle on identical seeded inputs at the
1e-5 norm-relative contract of BASELINE.json unless a test states its own): weight transport: csi_clone_weights (the receiver side of csi_broadcast_weights), RCCL self-broadcast at world size 1."""


import numpy as np
import pytest

from conftest import rel_rows

pytestmark = pytest.mark.gpu


TOL = 1e-5


def _weights(oracle, seed, nt, hidden, use_bn=True, n_out=234):
    rng = np.random.default_rng(seed)
    d_in = 320 * nt + nt
    return (oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn),
            oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn))


P_VHT4 = np.array([[1, -1, 1, 1], [1, 1, -1, 1], [1, 1, 1, -1], [-1, 1, 1, 1]], np.float64)


def vht_pilot(oracle, nt):
    """kron(H_{nt/4}, P_VHT4): Hadamard, NOT in the Sylvester order."""
    return np.kron(oracle.hadamard(nt // 4), P_VHT4)


# (tag, nt, nr, npkt, hidden, use_bn, dtype, pilot, options)
CLONE_CASES = [
    ('shipped_band', 32, 4, 48, (1024, 1024), True, 'f32', 'hadamard', {'f32_engine': 1}),
    ('shipped_fp32_mfma', 32, 2, 3, (1024, 1024), True, 'f32', 'hadamard', {'f32_engine': 0}),
    ('bf16_band', 32, 4, 40, (256, 256), True, 'bf16', 'hadamard', {}),
    ('generic_p_16', 16, 2, 24, (128, 256), True, 'f32', 'generic', {'f32_engine': 1}),
    ('vht_pilot_64', 64, 2, 10, (128, 128), True, 'f32', 'vht', {'f32_engine': 1}),
    ('nt128', 128, 2, 6, (64, 64), True, 'f32', 'hadamard', {'f32_engine': 1}),
    ('one_hidden_no_bn', 8, 2, 30, (128,), False, 'f32', 'hadamard', {'f32_engine': 1}),
    ('three_hidden', 16, 2, 20, (128, 64, 128), True, 'f32', 'generic', {'f32_engine': 1}),
]


def _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn=True, n_out=234, **kw):
    e = pkg.CsiEngine(nt, nr, hidden=hidden, n_out=n_out, use_bn=use_bn, **kw)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    return e


# ------------------------------------------------------------------------------------ RCCL inside the C-ABI
def test_rccl_self_broadcast_world1(pkg, oracle):
    """csi_get_unique_id / csi_comm_init / csi_broadcast_weights with one rank: the communicator comes up on RCCL, the
    broadcast walks every device buffer of both models and P (root = the only rank), and the context answers as before.
    (Two ranks on ONE GPU are refused by RCCL; the N > 1 path runs in the driver's scaling bench.)"""
    from dl_channel_estimation_mamimo_amd.engine import get_unique_id
    rng = np.random.default_rng(4)
    nt, nr, npkt, hidden = 8, 2, 6, (64, 48)
    w_re, w_im = _weights(oracle, 21, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=10.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    uid = get_unique_id()
    assert len(uid) == 128 and any(uid)
    e.comm_init(0, 1, uid)
    assert e.get_option('comm_world') == 1 and e.get_option('comm_rank') == 0
    with pytest.raises(pkg.CsiError):
        e.broadcast_weights(3)                                 # no such root
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    before = e.predict(ltf)
    moved = e.broadcast_weights(0)
    n_params = sum(int(np.prod(v.shape)) for v in w_re.values() if isinstance(v, np.ndarray)) * 2
    assert moved > 4 * n_params and e.get_option('comm_blobs') >= 2 * (3 * 3 + 2) + 2       # fp32 + split forms of every matrix, vectors, P
    after = e.predict(ltf)
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(after[0], r_re) < TOL and rel_rows(after[1], r_im) < TOL
    e.comm_destroy()
    assert e.get_option('comm_world') == 0


@pytest.mark.parametrize('tag,nt,nr,npkt,hidden,use_bn,dtype,pilot,opts', CLONE_CASES, ids=[c[0] for c in CLONE_CASES])
def test_clone_weights_receiver_is_bit_identical(pkg, oracle, tag, nt, nr, npkt, hidden, use_bn, dtype, pilot, opts):
    rng = np.random.default_rng(400 + nt + len(hidden))
    w_re, w_im = _weights(oracle, 40 + nt, nt, hidden, use_bn=use_bn)
    P = {'hadamard': lambda: oracle.hadamard(nt), 'vht': lambda: vht_pilot(oracle, nt),
         'generic': lambda: rng.integers(-2, 3, (nt, nt)).astype(np.float64) + 3.0 * np.eye(nt)}[pilot]()
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    kw = dict(hidden=hidden, use_bn=use_bn, dtype=dtype)
    root = pkg.CsiEngine(nt, nr, **kw)
    root.load_weights('real', w_re)
    root.load_weights('imag', w_im)
    root.set_pilot(P)
    recv = pkg.CsiEngine(nt, nr, **kw)
    # a receiver that already holds OTHER weights and another pilot: everything must be replaced
    junk_re, junk_im = _weights(oracle, 999, nt, hidden, use_bn=use_bn)
    recv.load_weights('real', junk_re)
    recv.load_weights('imag', junk_im)
    recv.set_pilot(np.eye(nt))
    recv.clone_weights_from(root)
    for e in (root, recv):
        for k, v in opts.items():
            e.set_option(k, v)
    for name in ('ls_mode', 'ls_pilot_pieces', 'ls_pilot_fast'):
        assert root.get_option(name) == recv.get_option(name), name
    a_re, a_im = root.predict(ltf)
    b_re, b_im = recv.predict(ltf)
    assert np.array_equal(a_re, b_re) and np.array_equal(a_im, b_im), tag
    assert root.get_option('band_launches') == recv.get_option('band_launches')
    assert root.get_option('hs_launches') == recv.get_option('hs_launches')
    ha, hb = root.ls_estimate(ltf), recv.ls_estimate(ltf)
    assert np.array_equal(ha, hb), tag
    # the literal (un-shared) network reads the fp32 / bf16 matrices and the full layer 0
    x = rng.standard_normal((5, 320 * nt + nt)).astype(np.float32)
    assert np.array_equal(root.predict_samples('imag', x), recv.predict_samples('imag', x))
    # ... and the receiver is right, not only equal
    k = min(npkt, 2)
    r_re, r_im = oracle.predict_packets(ltf[:k], P, w_re, w_im, np.float64, pkt_batch=k)
    tol = TOL if dtype == 'f32' else 2e-2
    assert rel_rows(b_re[:k], r_re) < tol and rel_rows(b_im[:k], r_im) < tol
    ref = oracle.ls_estimate(ltf[:k], P)
    assert rel_rows(np.concatenate([hb[:k].real, hb[:k].imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    # a changed pilot on the receiver alone rebuilds ITS tables (they are its own allocations, not aliases of the root's)
    s_re, _ = root.predict(ltf[:2])
    c0_re, _ = recv.predict(ltf[:2])
    assert np.array_equal(s_re, c0_re)
    recv.set_pilot(P[::-1].copy())
    c_re, _ = recv.predict(ltf[:2])
    s2_re, _ = root.predict(ltf[:2])
    assert np.array_equal(s2_re, s_re) and not np.array_equal(c_re, s_re)


def test_clone_weights_partial_and_refusals(pkg, oracle):
    """Only one component model / no pilot on the source; mismatched csi_config refused with text, receiver left empty."""
    nt, nr, hidden = 8, 2, (64, 64)
    w_re, w_im = _weights(oracle, 7, nt, hidden)
    src = pkg.CsiEngine(nt, nr, hidden=hidden)
    src.load_weights('real', w_re)                       # imag missing, no pilot
    dst = pkg.CsiEngine(nt, nr, hidden=hidden)
    dst.load_weights('imag', w_im)
    dst.set_pilot(oracle.hadamard(nt))
    dst.clone_weights_from(src)
    ltf = np.zeros((1, nr, 320 * nt), np.complex64)
    with pytest.raises(pkg.CsiError) as ei:              # the source had no pilot: neither has the clone now
        dst.predict(ltf)
    assert ei.value.code == -2 and 'csi_set_pilot' in str(ei.value)
    dst.set_pilot(oracle.hadamard(nt))
    with pytest.raises(pkg.CsiError) as ei:
        dst.predict(ltf)
    assert ei.value.code == -2 and 'imag' in str(ei.value)
    x = np.ones((2, 320 * nt + nt), np.float32)
    assert np.array_equal(dst.predict_samples('real', x), src.predict_samples('real', x))
    # refusals
    src.load_weights('imag', w_im)
    src.set_pilot(oracle.hadamard(nt))
    for kw, word in ((dict(hidden=(64, 32)), 'hidden layer 1'), (dict(hidden=(64,)), 'hidden layers'),
                     (dict(hidden=hidden, dtype='bf16'), 'dtype'), (dict(hidden=hidden, use_bn=False), 'use_bn')):
        other = pkg.CsiEngine(nt, nr, **kw)
        hd = kw['hidden']
        o_re, _ = _weights(oracle, 8, nt, hd, use_bn=kw.get('use_bn', True))
        other.load_weights('real', o_re)
        other.set_pilot(oracle.hadamard(nt))
        with pytest.raises(pkg.CsiError) as ei:
            other.clone_weights_from(src)
        assert ei.value.code == -1 and word in str(ei.value), (kw, str(ei.value))
        with pytest.raises(pkg.CsiError) as ei2:         # refused -> empty, never half a model
            other.predict_samples('real', np.ones((1, 320 * nt + nt), np.float32))
        assert ei2.value.code == -2
    with pytest.raises(pkg.CsiError):
        src.clone_weights_from(src)
    other = pkg.CsiEngine(16, nr, hidden=hidden)
    with pytest.raises(pkg.CsiError) as ei:
        other.clone_weights_from(src)
    assert 'nt' in str(ei.value)
    # the source is untouched by all of it
    ltf = (np.random.default_rng(0).standard_normal((2, nr, 320 * nt)) + 0j).astype(np.complex64)
    o_re, o_im = src.predict(ltf)
    r_re, r_im = oracle.predict_packets(ltf, oracle.hadamard(nt), w_re, w_im, np.float64, pkt_batch=2)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL


def test_rccl_self_broadcast_world1_with_status_word(pkg, oracle):
    """World of one rank through RCCL: record broadcast, the ranks' status all-reduce, grouped blob broadcast (root == self)."""
    nt, nr, hidden = 16, 2, (128, 128)
    w_re, w_im = _weights(oracle, 11, nt, hidden)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(vht_pilot(oracle, nt))
    ltf = (np.random.default_rng(1).standard_normal((3, nr, 320 * nt)) + 0j).astype(np.complex64)
    a_re, a_im = e.predict(ltf)
    e.comm_init(0, 1, pkg.engine.get_unique_id())
    moved = e.broadcast_weights(0)
    assert moved > 0 and e.get_option('comm_blobs') > 10
    b_re, b_im = e.predict(ltf)
    assert np.array_equal(a_re, b_re) and np.array_equal(a_im, b_im)
    assert e.get_option('ls_pilot_fast') == 2
    e.comm_destroy()


def test_a_receiver_counts_a_pinned_model_and_success_leaves_no_error_text(pkg, oracle):
    """csi_load_weights used to leave an explanatory text in csi_last_error while returning CSI_OK when it pinned a model
    to the fp32 MFMA kernels, and a context that RECEIVED such a model (csi_clone_weights = the receiver side of csi_broadcast_weights)
    read 0 for "hs_weight_pins" / "hs_weight_err_e12".  Now: no text after a successful load, and the receiver's counters equal the
    sender's."""
    nt, nr, hidden = 32, 2, (64, 64)
    w_re, w_im = _weights(oracle, 5, nt, hidden)
    bad = {k: np.array(v, copy=True) for k, v in w_im.items()}
    bad['fc_dense1.kernel'] *= 2.0 ** -22
    bad['fc_dense1.kernel'][3, 5] = 1.0                     # one entry 2^22 above the rest: the split copies are not fp32-grade
    e = _engine(pkg, nt, nr, hidden, w_re, bad, oracle.hadamard(nt))
    assert e.get_option('hs_weight_pins') == 1 and e.get_option('hs_weight_err_e12') > 1e6
    assert (e._lib.csi_last_error(e._ctx) or b'') == b'', 'a successful csi_load_weights leaves no error text'
    r = pkg.CsiEngine(nt, nr, hidden=hidden)
    r.clone_weights_from(e)
    assert r.get_option('hs_weight_pins') == 1 and r.get_option('hs_weight_err_e12') == e.get_option('hs_weight_err_e12')
    rng = np.random.default_rng(1)
    ltf = oracle.make_structured_packets(rng, 40, nr, oracle.hadamard(nt), snr_db=5.0)[0].astype(np.complex64)
    a, b = e.predict(ltf), r.predict(ltf)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
