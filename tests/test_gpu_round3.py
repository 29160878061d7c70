"""GPU tests added in round 3 (-m gpu; everything through the C-ABI, checked against the numpy oracle):

  * one HOT packet / one hot rx row inside a config-2 sized batch (split-f16 engine: the input scale comes from a strided
    sample, so a loud item may or may not be seen) - every sampled packet incl. the hot one and its neighbours inside
    the contract, the range guard / the wrapper's recovery telling which engine served the call;
  * BASELINE configs[0] as it is named: CSIPredictor(..., experiment='matlab_maMimo').inference on 500 structured packets
    at 0 dB, Nt=32, Nr=4, shipped model;
  * the seeded, bounded runs of the fuzzers (tests/fuzz_ls.py, tests/fuzz_shapes.py) that used to live outside pytest.
"""
import os
import sys

import numpy as np
import pytest

from conftest import rel_rows

pytestmark = pytest.mark.gpu
TOL = 1e-5
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def _weights(oracle, seed, nt, hidden, use_bn=True, n_out=234):
    rng = np.random.default_rng(seed)
    d_in = 320 * nt + nt
    return (oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn),
            oracle.make_weights(rng, d_in, list(hidden), n_out, use_bn=use_bn))


@pytest.fixture(scope='module')
def config2(pkg, oracle):
    """Engine with the shipped model at Nt=32, Nr=4 and 4000 white packets resident in HBM (config-2 size)."""
    nt, nr, hidden, npkt = 32, 4, (1024, 1024), 4000
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = pkg.synth.hadamard(nt)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(77, 0, npkt, d_re, d_im)
    outs = tuple(e.empty((npkt, nr, nt, 234)) for _ in range(4))
    e.synchronize()
    return dict(e=e, nt=nt, nr=nr, npkt=npkt, w_re=w_re, w_im=w_im, P=P, d_re=d_re, d_im=d_im, outs=outs)


def _sampled_blocks(npkt, nr, len_ltf):
    """Which 1-KiB blocks (256 floats) of a preamble plane hs_absmax_sample_kernel reads (csi_dnn_hs.hpp: every step-th)."""
    nblk = (npkt * nr * len_ltf // 4 + 63) // 64
    return max(1, nblk // 4096), nblk


# (gain, what is scaled, seen by the magnitude sample?)
HOT_CASES = [(2.0 ** 12, 'packet', True), (2.0 ** 20, 'packet', True), (2.0 ** 12, 'row', False), (2.0 ** 20, 'row', False)]


@pytest.mark.parametrize('gain,what,sampled', HOT_CASES)
def test_hot_packet_in_a_config2_batch(pkg, oracle, config2, gain, what, sampled):
    c = config2
    e, nt, nr, npkt = c['e'], c['nt'], c['nr'], c['npkt']
    step, nblk = _sampled_blocks(npkt, nr, e.len_ltf)
    blocks_per_row = e.len_ltf // 256
    hot_p, hot_r = 1777, None
    if what == 'row':
        # an rx row none of whose 1-KiB blocks is read by the sample
        for p in range(1500, 2500):
            for r in range(nr):
                b0 = (p * nr + r) * blocks_per_row
                if all(b % step for b in range(b0, b0 + blocks_per_row)):
                    hot_p, hot_r = p, r
                    break
            if hot_r is not None:
                break
        assert hot_r is not None, 'no unsampled row (the sample stride changed?)'
    else:
        b0 = hot_p * nr * blocks_per_row
        assert any(b % step == 0 for b in range(b0, b0 + nr * blocks_per_row))
    keep_re, keep_im = c['d_re'].download(hot_p, 1), c['d_im'].download(hot_p, 1)
    hot_re, hot_im = keep_re.copy(), keep_im.copy()
    rows = slice(None) if hot_r is None else slice(hot_r, hot_r + 1)
    hot_re[0, rows] *= gain
    hot_im[0, rows] *= gain
    try:
        c['d_re'].upload(hot_re, first=hot_p)
        c['d_im'].upload(hot_im, first=hot_p)
        pick = [0, hot_p - 1, hot_p, hot_p + 1, npkt - 1]
        ltf = np.concatenate([c['d_re'].download(p, 1) + 1j * c['d_im'].download(p, 1) for p in pick])
        r_re, r_im = oracle.predict_packets_shared(ltf, c['P'], c['w_re'], c['w_im'])
        o_re, o_im, h_re, h_im = c['outs']
        e.set_option('f32_engine', -1)

        def verify(tag):
            g_re = np.concatenate([o_re.download(p, 1) for p in pick])
            g_im = np.concatenate([o_im.download(p, 1) for p in pick])
            assert np.isfinite(g_re).all() and np.isfinite(g_im).all(), tag
            for i, p in enumerate(pick):                  # per packet: the quiet ones must not hide behind the hot one
                assert rel_rows(g_re[i], r_re[i]) < TOL and rel_rows(g_im[i], r_im[i]) < TOL, (tag, p)

        # 1. device-pointer call, the wrapper recovering from CSI_ERR_RANGE
        n0, f0 = e.get_option('hs_launches'), getattr(e, 'range_recoveries', 0)
        served = e.predict_device(c['d_re'], c['d_im'], npkt, o_re, o_im, checked=True)
        assert e.get_option('hs_launches') > n0                  # the split engine did take the call first
        assert served in ('split', 'fp32') and (served == 'fp32') == (getattr(e, 'range_recoveries', 0) == f0 + 1)
        if gain >= 2.0 ** 20 or not sampled:
            # 2^20: the rest of the batch falls into the f16 denormals of a scale chosen for the hot packet (low-side
            # guard), or - not seen by the sample - the hot row overflows f16 (high-side guard): the fp32 kernels must serve it
            assert served == 'fp32', (gain, what)
        assert e.get_option('f32_engine') == -1
        verify('predict_device checked -> ' + served)

        # 2. LS + DNN as one hipGraph-replayed call: eager, capture, replay - each one recovered the same way
        e.set_option('use_graph', 1)
        for it in range(3):
            o_re.upload(np.zeros((1, nr, nt, 234), np.float32), first=hot_p)
            s2 = e.estimate_device(c['d_re'], c['d_im'], npkt, o_re, o_im, h_re, h_im, checked=True)
            assert s2 == served, (it, s2, served)
            verify('estimate_device + graph, call %d -> %s' % (it, s2))
        e.set_option('use_graph', 0)
        ref = oracle.ls_estimate(ltf, c['P'])
        h = np.concatenate([h_re.download(p, 1) + 1j * h_im.download(p, 1) for p in pick])
        assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL

        # 3. host-buffer entry point on the packets around the hot one (its own automatic repeat; `hs_range_fallbacks` counts it)
        fb0 = e.get_option('hs_range_fallbacks')
        sl = slice(hot_p - 40, hot_p + 40)
        x_re = np.concatenate([c['d_re'].download(p, 1) for p in range(sl.start, sl.stop)])
        x_im = np.concatenate([c['d_im'].download(p, 1) for p in range(sl.start, sl.stop)])
        e.set_option('f32_engine', 1)
        s_re, s_im = e.predict(x_re, x_im)
        e.set_option('f32_engine', -1)
        fb = e.get_option('hs_range_fallbacks') - fb0
        assert fb in (0, 1)
        assert rel_rows(s_re[39:42], r_re[1:4]) < TOL and rel_rows(s_im[39:42], r_im[1:4]) < TOL, fb
        print('hot %s x2^%d (sampled=%s): device call served by %s, host call fallbacks %d'
              % (what, int(np.log2(gain)), sampled, served, fb))
    finally:
        c['d_re'].upload(keep_re, first=hot_p)
        c['d_im'].upload(keep_im, first=hot_p)
        e.set_option('use_graph', 0)
        e.set_option('f32_engine', -1)


def test_range_guard_retry_does_not_replay_split_graphs(pkg, oracle):
    """Round-2 advice: with use_graph the host pipeline calls the device entry points with recurring buffers / chunk sizes,
    so the range-guard repeat of csi_predict / csi_estimate_c128 could replay a hipGraph captured with the split-f16 kernels
    and hand the overflowed outputs back as CSI_OK.  Same call three times (eager, capture, replay) on data that trips the
    guard: every one must come back inside the contract, counted as a fallback."""
    rng = np.random.default_rng(3)
    nt, nr, npkt, hidden = 8, 2, 20, (64, 64)
    w_re, w_im = _weights(oracle, 17, nt, hidden)
    P = oracle.hadamard(nt)
    base = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=10.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    e.set_option('f32_engine', 1)
    e.set_option('use_graph', 1)
    huge = (3.0e4 * base).astype(np.complex128)
    r_re, r_im = oracle.predict_packets(huge.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    for it in range(4):
        fb = e.get_option('hs_range_fallbacks')
        s_re, s_im = e.predict(huge.astype(np.complex64))
        assert e.get_option('hs_range_fallbacks') == fb + 1, it
        assert np.isfinite(s_re).all() and rel_rows(s_re, r_re) < TOL and rel_rows(s_im, r_im) < TOL, it
        dnn, _ = e.estimate(huge, dnn=True, ls=False)
        assert e.get_option('hs_range_fallbacks') == fb + 2, it
        assert rel_rows(dnn.real, r_re) < TOL and rel_rows(dnn.imag, r_im) < TOL, it
    assert e.get_option('f32_engine') == 1 and e.get_option('use_graph') == 1


def test_config1_csipredictor_500_packets_0db(pkg, oracle, tmp_path):
    """BASELINE configs[0]: Nt=32, Nr=4, TEST_Npkt=500, SNR 0 dB through the reference's deployment surface -
    CSIPredictor(model_path, experiment='matlab_maMimo').inference (inference.py:24-32) on the shipped architecture, the
    model folders written the way the reference's test run leaves them (DNN.py:411), LS through estimate()."""
    nt, nr, npkt, hidden = 32, 4, 500, (1024, 1024)
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = pkg.synth.hadamard(nt)
    e0 = pkg.CsiEngine(nt, nr, hidden=hidden)
    for d, w in (('real', w_re), ('imag', w_im)):
        pkg.CSIModel(e0, d).load_weights(w).save(str(tmp_path / f'{d}_keras_model'), pilot=P)
    e0.close()
    pred = pkg.CSIPredictor(str(tmp_path), experiment='matlab_maMimo')
    ltf = np.concatenate([blk for _, _, blk in pkg.synth.mixed_snr_batch(2024, nr, P, per_level=npkt, levels=(0.0,))]).astype(np.complex128)
    assert ltf.shape == (npkt, nr, 320 * nt)
    csi = pred.inference(ltf)
    assert csi.shape == (npkt, nr, nt, 234) and csi.dtype == np.complex64 and np.isfinite(csi.view(np.float32)).all()
    pick = [0, 1, 249, 498, 499]
    r_re, r_im = oracle.predict_packets_shared(ltf[pick].astype(np.complex64), P, w_re, w_im)
    ref = oracle.recombine(r_re, r_im)
    assert rel_rows(np.concatenate([csi[pick].real, csi[pick].imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    # literal Model.predict of one packet's 128 samples (DNN.py:339-346) agrees with the packet path
    x = oracle.samples_from_packets(ltf[249:250].astype(np.complex64), P.astype(np.float32), 'real')
    lit = pred.model_real.predict(x)
    assert rel_rows(lit.reshape(nr, nt, 234), r_re[2]) < TOL
    dnn, h_ls = pred.estimate(ltf)
    assert np.array_equal(dnn, csi)
    ref_ls = oracle.ls_estimate(ltf[pick].astype(np.complex64), P)
    assert rel_rows(np.concatenate([h_ls[pick].real, h_ls[pick].imag], -1), np.concatenate([ref_ls.real, ref_ls.imag], -1)) < TOL
    # the figure the pipeline reports for this configuration: NMSE_subk (BER_test_maMIMO_LTF.m:675-686) of the DNN vs the LS labels
    assert abs(oracle.nmse_subk(h_ls[pick], csi[pick]) - oracle.nmse_subk(ref_ls, ref)) < 1e-4 * oracle.nmse_subk(ref_ls, ref)
    with pytest.raises(SystemExit) as ex:
        pred.inference(ltf.astype(np.complex64))
    assert ex.value.code == -1


@pytest.mark.parametrize('seed,cases', [(5, 10), (11, 10)])
def test_fuzz_ls_cases(seed, cases):
    """tests/fuzz_ls.py inside the suite: random antenna / rx / packet counts, Hadamard or generic pilots, every LS kernel
    that can serve the shape."""
    import fuzz_ls
    rng = np.random.default_rng(seed)
    log = []
    fails = sum(fuzz_ls.run_case(rng, c, log.append) for c in range(cases))
    assert fails == 0, '\n'.join(l for l in log if l.startswith('FAIL'))


@pytest.mark.parametrize('seed,cases', [(0, 10), (7, 10)])
def test_fuzz_shape_cases(seed, cases):
    """tests/fuzz_shapes.py inside the suite: random Nt / Nr / packets / widths / depth / BN / dtype / engine / tile options,
    shared-layer-0 path, literal path and LS against the oracle."""
    import fuzz_shapes
    rng = np.random.default_rng(seed)
    log = []
    bad = [i for i in range(cases) if not fuzz_shapes.run_case(rng, i, log.append)]
    assert not bad, '\n'.join(log[i] for i in bad)


# ------------------------------------------------------------------------------------ fused band kernel (assembly)
BAND_CASES = [
    (32, 4, 24, (1024, 1024)),      # the shipped network; 3072 rows = 24 bands
    (8, 2, 70, (128, 256)),         # K1 = 128: no trip of the stage-1 loop; one column step; ragged last band (1120 rows)
    (4, 1, 131, (192, 512)),        # K1 = 192: one loop trip; two column steps; 524 rows
    (12, 2, 21, (256, 768)),        # Nt that does not divide the band: rows of one band span several (packet, rx) items
    # 16 <= Nt <= 128: the form that streams the L0 / pilot-table values through LDS (Nt = 32 above as well)
    (16, 2, 41, (128, 256)),        # smallest Nt of that form: 8-9 L0 rows per band, a 1-KiB table slab; ragged last band (1312 rows)
    (24, 2, 21, (256, 256)),        # L0 rows change inside a wave
    (48, 3, 9, (192, 512)),
    (100, 1, 5, (128, 256)),        # table slab of 6400 bytes: the last DMA chunk reaches into the next slab
    (128, 2, 3, (256, 256)),        # largest: 8-KiB slabs, one chunk per wave
]


@pytest.mark.parametrize('nt,nr,npkt,hidden', BAND_CASES)
def test_band_kernel_matches_oracle_and_separate_kernels(pkg, oracle, nt, nr, npkt, hidden):
    """First per-pair layer + regressor as ONE kernel (band_kernel_gen.py, option hs_band): against the fp64 oracle, against
    the two kernels it replaces, run-to-run identical, and really launched."""
    rng = np.random.default_rng(nt * 1000 + npkt)
    w_re, w_im = _weights(oracle, 5 + nt, nt, hidden)
    P = oracle.hadamard(nt) if nt & (nt - 1) == 0 else rng.integers(-2, 3, (nt, nt)).astype(np.float64)
    ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=5.0)[0] if nt & (nt - 1) == 0 else \
        (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    e.set_option('f32_engine', 1)
    e.set_option('band_split', 0)                             # (the column-split launch of small calls sums in another order: its own test, round 5)
    assert e.get_option('hs_band') == 1                       # the default
    n0 = e.get_option('band_launches')
    b_re, b_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 2, 'the band kernel did not serve the call'
    assert e.get_option('hs_range_fallbacks') == 0
    k = min(npkt, 6)
    sel = np.r_[0:k // 2, npkt - (k - k // 2):npkt]
    r_re, r_im = oracle.predict_packets(ltf[sel].astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=len(sel))
    assert rel_rows(b_re[sel], r_re) < TOL and rel_rows(b_im[sel], r_im) < TOL
    b2_re, _ = e.predict(ltf)
    assert np.array_equal(b_re, b2_re)
    e.set_option('hs_band', 3)                                # the form with per-lane global loads of L0 / T: the same arithmetic
    p_re, p_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 6
    assert np.array_equal(b_re, p_re) and np.array_equal(b_im, p_im)
    e.set_option('hs_band', 0)
    s_re, s_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 6
    assert rel_rows(b_re, s_re) < 5e-6 and rel_rows(b_im, s_im) < 5e-6


def test_band_kernel_range_guard_and_graph(pkg, oracle):
    """The band kernel carries the split engine's range guard (both words) and is capturable: data that overflows the hidden
    activations makes csi_predict repeat on the fp32 MFMA kernels; a hipGraph of the device call replays bit-identically."""
    rng = np.random.default_rng(9)
    nt, nr, npkt, hidden = 8, 2, 40, (128, 256)
    w_re, w_im = _weights(oracle, 3, nt, hidden)
    P = oracle.hadamard(nt)
    base = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=10.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    e.set_option('f32_engine', 1)
    for gain, fallbacks in ((1.0, 0), (3.0e4, 1)):
        ltf = (gain * base).astype(np.complex64)
        fb, bl = e.get_option('hs_range_fallbacks'), e.get_option('band_launches')
        o_re, o_im = e.predict(ltf)
        assert e.get_option('band_launches') == bl + 2 and e.get_option('hs_range_fallbacks') == fb + fallbacks, gain
        r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
        assert np.isfinite(o_re).all() and rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL, gain
    d_re, d_im = e.to_device(np.ascontiguousarray(base.real, np.float32)), e.to_device(np.ascontiguousarray(base.imag, np.float32))
    o1, o2 = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.predict_device(d_re, d_im, npkt, o1, o2)
    e.synchronize()
    eager = o1.download().copy()
    e.set_option('use_graph', 1)
    for _ in range(4):
        o1.upload(np.zeros((npkt, nr, nt, 234), np.float32))
        e.predict_device(d_re, d_im, npkt, o1, o2)
        e.synchronize()
        assert np.array_equal(o1.download(), eager)
    assert e.get_option('graph_replays') >= 1


def test_full_size_config2_band_vs_separate(pkg, oracle, config2):
    """BASELINE configs[1] size (512 000 pair rows per component model): the band kernel against the separate kernels on
    every 97th packet, and against the oracle on a handful."""
    c = config2
    e, npkt = c['e'], c['npkt']
    o_re, o_im, s_re, s_im = c['outs']
    e.set_option('f32_engine', -1)
    e.set_option('hs_band', 1)
    bl = e.get_option('band_launches')
    assert e.predict_device(c['d_re'], c['d_im'], npkt, o_re, o_im, checked=True) == 'split'
    assert e.get_option('band_launches') == bl + 2
    e.set_option('hs_band', 0)
    assert e.predict_device(c['d_re'], c['d_im'], npkt, s_re, s_im, checked=True) == 'split'
    e.set_option('hs_band', 1)
    pick = list(range(0, npkt, 97)) + [npkt - 1]
    a = np.concatenate([o_re.download(p, 1) for p in pick])
    b = np.concatenate([s_re.download(p, 1) for p in pick])
    assert np.isfinite(a).all() and rel_rows(a, b) < 5e-6
    few = [0, 1234, npkt - 1]
    ltf = np.concatenate([c['d_re'].download(p, 1) + 1j * c['d_im'].download(p, 1) for p in few])
    r_re, r_im = oracle.predict_packets_shared(ltf, c['P'], c['w_re'], c['w_im'])
    assert rel_rows(np.concatenate([o_re.download(p, 1) for p in few]), r_re) < TOL
    assert rel_rows(np.concatenate([o_im.download(p, 1) for p in few]), r_im) < TOL


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


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(8, 2, 37, (256, 256)), (64, 2, 5, (384, 512)), (4, 2, 70, (1024, 256)),
                                               # 32 <= nt <= 64: the form with the L0 / T values streamed through LDS (Nt = 48: L0 rows change inside a wave)
                                               (32, 4, 9, (256, 256)), (48, 2, 7, (512, 256)), (64, 4, 33, (1024, 1024)), (40, 3, 5, (256, 512))])
def test_band_kernel_bf16_mode(pkg, oracle, nt, nr, npkt, hidden):
    """BASELINE configs[2] arithmetic (bf16 operands, fp32 accumulation): the bf16 form of the band kernel against the
    oracle's bf16-operand emulation (same rounding points: h1 and h2 rounded to bf16 once) and against the separate bf16
    kernels it replaces."""
    rng = np.random.default_rng(nt + npkt)
    w_re, w_im = _weights(oracle, 40 + nt, nt, hidden)
    P = oracle.hadamard(nt) if nt & (nt - 1) == 0 else rng.choice([-1.0, 1.0], (nt, nt))
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)[0]
    e = pkg.CsiEngine(nt, nr, hidden=hidden, dtype='bf16')
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    e.set_option('force_tile', 256)                           # the large-grid kernels regardless of the batch size
    n0 = e.get_option('band_launches')
    d_re, d_im = e.predict(ltf)
    staged = 32 <= nt <= 64
    # default: the band kernel where its staged form applies (3.5 against 3.4 + 1.0 ms at configs[2]); elsewhere the separate kernels
    assert e.get_option('band_launches') == n0 + (2 if staged else 0)
    n0 = e.get_option('band_launches')
    e.set_option('hs_band', 2)
    b_re, b_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 2, 'the bf16 band kernel did not serve the call'
    if staged:
        assert np.array_equal(b_re, d_re) and np.array_equal(b_im, d_im)
    r_re, r_im = oracle.predict_packets_bf16(ltf.astype(np.complex64), P, w_re, w_im)
    assert np.isfinite(b_re).all()
    assert rel_rows(b_re, r_re) < 4e-3 and rel_rows(b_im, r_im) < 4e-3          # accumulation-order re-roundings of the bf16 activations only
    assert np.array_equal(b_re, e.predict(ltf)[0])
    e.set_option('hs_band', 0)
    s_re, s_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 4
    assert rel_rows(b_re, s_re) < 4e-3 and rel_rows(b_im, s_im) < 4e-3
