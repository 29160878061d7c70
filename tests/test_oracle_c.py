"""CPU tests (-m "not gpu"): the plain-C statement of the oracle (oracle/csi_oracle_c.c, built with gcc) against the vectors the
reference's own numpy code produced (tests/golden/ref_*.npz), against the committed fixture of the numpy statement
(tests/golden/oracle_nt4.npz) and against the numpy statement itself on seeded problems.  The two statements share no code - the C one
has its own DFT, its own carrier / LTF tables and plain dot products - so agreement to fp64 rounding is evidence about both."""
import os

import numpy as np
import pytest


@pytest.fixture(scope='module')
def oc():
    from oracle import csi_oracle_c
    csi_oracle_c.build()
    return csi_oracle_c


def _w(g, tag):
    return {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + '.')}


def test_c_tables_equal_the_numpy_statement(oc, oracle):
    np.testing.assert_array_equal(oc.data_carrier_indices(), oracle.data_carrier_indices())
    np.testing.assert_array_equal(oc.vht_ltf_256(), oracle.vht_ltf_256())


def test_c_ofdm_demod_matches_reference_reshape_method(oc, oracle, golden_dir):
    """Row a-1 of the C statement pinned to the spectra recorded from the reference's own code
    (massiveMIMO_dataGenerator.py:425-453, executed by tests/golden/make_golden.py)."""
    g = np.load(os.path.join(golden_dir, 'ref_ofdm_reshape_nt4.npz'))
    nt = int(g['nt'])
    spec = g['real_fft_pre_shift'] + 1j * g['imag_fft_pre_shift']                 # [pr, 256 bins (un-shifted), nt symbols]
    want = np.fft.fftshift(spec, axes=1)[:, oracle.data_carrier_indices() - 1, :]
    rx = oc.ofdm_demod(g['ds_ltf_real'] + 1j * g['ds_ltf_imag'], nt)
    np.testing.assert_allclose(rx, want, rtol=0, atol=1e-11 * np.abs(want).max())
    # and the despread of those reference-computed spectra equals the C statement's whole time-domain LS path
    h_from_ref = np.swapaxes(oc.ls_from_rxsym(want, g['P_matlab']), -1, -2)
    h = oc.ls_estimate(g['ds_ltf_real'] + 1j * g['ds_ltf_imag'], g['P_matlab'])
    np.testing.assert_allclose(h, h_from_ref, rtol=0, atol=1e-11 * np.abs(h_from_ref).max())


def test_c_sample_rows_match_reference_datagenerator(oc, golden_dir):
    """Row a-3: [Xsig | Xp] rows as the reference's DataGenerator produced them (massiveMIMO_dataGenerator.py:299-316)."""
    g = np.load(os.path.join(golden_dir, 'ref_datagen_nt4.npz'))
    nt, nr, npkt = int(g['nt']), int(g['nr']), int(g['npkt'])
    ltf = (g['ds_ltf_real'] + 1j * g['ds_ltf_imag']).reshape(npkt, nr, 320 * nt)
    for d in ('real', 'imag'):
        ref = np.concatenate([g[f'{d}_Xsig'][..., 0], g[f'{d}_Xp']], axis=-1).reshape(npkt * nr * nt, -1)
        np.testing.assert_array_equal(oc.samples_from_packets(ltf, g['P_matlab'], d), ref)
    # the fixture's P is deliberately non-symmetric: its transpose must NOT reproduce the reference's rows
    assert not np.array_equal(oc.samples_from_packets(ltf, g['P_matlab'].T, 'imag'), ref)


def test_c_statement_reproduces_the_committed_oracle_fixture(oc, golden_dir):
    """tests/golden/oracle_nt4.npz holds what the numpy statement produced for a seeded Nt = 4 problem (the fixture the GPU suite
    checks the HIP path against without running any oracle): the C statement reproduces it."""
    g = np.load(os.path.join(golden_dir, 'oracle_nt4.npz'))
    ltf = g['ltf'].astype(np.complex128)
    h = oc.ls_estimate(ltf, g['P'])
    np.testing.assert_allclose(h, g['ls'], rtol=0, atol=1e-12 * np.abs(g['ls']).max())
    o_re, o_im = oc.predict_packets(ltf, g['P'], _w(g, 'w_re'), _w(g, 'w_im'))
    for got, want in ((o_re, g['dnn_real']), (o_im, g['dnn_imag'])):
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max())
    np.testing.assert_allclose(o_re + 1j * o_im, g['csi'], rtol=0, atol=1e-12 * np.abs(g['csi']).max())


@pytest.mark.parametrize('nt,nr,kind', [(4, 2, 'complex'), (8, 3, 'hadamard'), (32, 2, 'signed'), (6, 1, 'unitary'), (24, 2, 'generic'), (100, 1, 'generic')])
def test_c_ls_equals_numpy_ls_and_the_known_channel(oc, oracle, nt, nr, kind):
    """helperMIMOChannelEstimate.m:24-36 in the two statements, real and complex P (the conjugate transpose of :24 matters only for
    the complex one); with P P^H = Nt I the estimate is the channel the packet was synthesised from."""
    rng = np.random.default_rng(100 + nt)
    if kind == 'hadamard':
        P = oracle.hadamard(nt)
    elif kind == 'signed':
        P = oracle.hadamard(nt)[rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[:, None]
    elif kind == 'generic':
        P = rng.integers(-3, 4, (nt, nt)).astype(np.float64)                      # not orthogonal: LS.m:24-36 is still rx * P' ./ denom
    else:
        q, _ = np.linalg.qr(rng.standard_normal((nt, nt)) + 1j * rng.standard_normal((nt, nt)))
        P = q * np.sqrt(nt) if kind == 'complex' else (np.linalg.qr(rng.standard_normal((nt, nt)))[0] * np.sqrt(nt))
    if np.iscomplexobj(P):
        # synthesise the received symbols directly: rx = H P on the data bins (the identity behind the known-answer test)
        H = rng.standard_normal((3, nr, 234, nt)) + 1j * rng.standard_normal((3, nr, 234, nt))
        ltf_seq = oracle.vht_ltf_256()[oracle.data_carrier_indices() - 1]
        rx = (H * ltf_seq[:, None]) @ P                                           # [.., 234, symbol]
        got = oc.ls_from_rxsym(rx, P)
        np.testing.assert_allclose(got, oracle.ls_from_rxsym(rx, P), rtol=0, atol=1e-12 * np.abs(H).max())
        np.testing.assert_allclose(got, H, rtol=0, atol=1e-12 * np.abs(H).max())
        return
    ltf, H = oracle.make_structured_packets(rng, 3, nr, P)
    got = oc.ls_estimate(ltf, P)
    want = oracle.ls_estimate(ltf, P)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-11 * np.abs(want).max())
    if kind != 'generic':
        np.testing.assert_allclose(got, H, rtol=0, atol=1e-9 * np.abs(H).max())


@pytest.mark.parametrize('hidden,use_bn', [((64,), True), ((48, 96), True), ((32, 64, 16), False), ((1024, 1024), True)])
def test_c_network_equals_numpy_network(oc, oracle, hidden, use_bn):
    """massiveMIMO_CSI_prediction_DNN.py:207-227 in the two statements: Dense + relu, BatchNormalization AFTER the relu (keras
    inference form, eps 1e-3), linear regressor - depth 1 to 3, with and without BatchNormalization, and the shipped widths."""
    rng = np.random.default_rng(len(hidden) * 7 + hidden[0])
    nt, nr, npkt = 4, 2, 2
    P = oracle.hadamard(nt)[::-1].copy()
    ltf, _ = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=3.0)
    w_re = oracle.make_weights(rng, 320 * nt + nt, hidden, 234, use_bn=use_bn)
    w_im = oracle.make_weights(rng, 320 * nt + nt, hidden, 234, use_bn=use_bn)
    want = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    got = oc.predict_packets(ltf, P, w_re, w_im)
    for a, b in zip(got, want):
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12 * np.abs(b).max())
    # a BatchNormalization placed BEFORE the relu would not pass: the check has teeth
    if use_bn:
        x = oracle.samples_from_packets(ltf, P, 'real')
        h = x @ w_re['fc_dense0.kernel'].astype(np.float64) + w_re['fc_dense0.bias']
        wrong = np.maximum(oracle.bn_inference(h, *(w_re[f'bn0.{k}'].astype(np.float64) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance'))), 0)
        right = oracle.bn_inference(np.maximum(h, 0), *(w_re[f'bn0.{k}'].astype(np.float64) for k in ('gamma', 'beta', 'moving_mean', 'moving_variance')))
        assert np.abs(wrong - right).max() > 1e-3 * np.abs(right).max()


def test_c_network_against_torch_functional(oc, oracle):
    """The C statement against torch.nn.functional.linear / batch_norm(training=False, eps=1e-3) in float64 - a third, unrelated
    implementation of the published layer definitions."""
    torch = pytest.importorskip('torch')
    F = torch.nn.functional
    rng = np.random.default_rng(9)
    w = oracle.make_weights(rng, 50, (40, 24), 234, use_bn=True)
    x = rng.standard_normal((17, 50))
    h = torch.from_numpy(x)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    for i in range(2):
        h = F.relu(F.linear(h, t(w[f'fc_dense{i}.kernel']).T, t(w[f'fc_dense{i}.bias'])))
        h = F.batch_norm(h, t(w[f'bn{i}.moving_mean']), t(w[f'bn{i}.moving_variance']), t(w[f'bn{i}.gamma']), t(w[f'bn{i}.beta']),
                         training=False, eps=1e-3)
    want = F.linear(h, t(w['fc_regressor.kernel']).T, t(w['fc_regressor.bias'])).numpy()
    got = oc.fc_forward(x, w)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max())


@pytest.mark.parametrize('snr_db', [-5.0, 0.0, 10.0, 25.0])
def test_c_lmmse_equals_numpy_lmmse(oc, oracle, snr_db):
    """LMMSE_ce.m:23-39 in the two statements (the numpy one forms inv(Rpp) as :39 does, the C one eliminates): one link at the
    reference's sizes Nfft = Np = 234, Nps = 1 (helperMIMOChannelEstimate.m:37-39), and the per-link loop over a small packet."""
    rng = np.random.default_rng(int(snr_db) + 50)
    h_tilde = rng.standard_normal(234) + 1j * rng.standard_normal(234)
    h = np.abs(rng.standard_normal(6)) * 3.0                                       # "h_tau": a few positive delays
    want = oracle.lmmse_ce(h_tilde, 234, 234, 1, h, snr_db)
    got = oc.lmmse_ce(h_tilde, 234, 234, 1, h, snr_db)
    assert np.linalg.norm(got - want) < 1e-9 * np.linalg.norm(want)
    h_ls = rng.standard_normal((2, 2, 1, 234)) + 1j * rng.standard_normal((2, 2, 1, 234))
    hs = np.abs(rng.standard_normal((2, 5))) * 2.0
    snr = np.full((2, 2), snr_db) + rng.standard_normal((2, 2))
    a, b = oc.lmmse_estimate(h_ls, hs, snr), oracle.lmmse_estimate(h_ls, hs, snr)
    assert np.linalg.norm(a - b) < 1e-9 * np.linalg.norm(b)


def test_c_nmse_equals_numpy_nmse(oc, oracle):
    rng = np.random.default_rng(3)
    ref = rng.standard_normal((5, 2, 4, 234)) + 1j * rng.standard_normal((5, 2, 4, 234))
    est = ref + 0.1 * (rng.standard_normal(ref.shape) + 1j * rng.standard_normal(ref.shape))
    assert abs(oc.nmse_subk(ref, est) - oracle.nmse_subk(ref, est)) < 1e-14
    assert oc.nmse_subk(ref, ref) == 0.0 and abs(oc.nmse_subk(ref, 0.9 * ref) - 0.01) < 1e-12


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(128, 2, 1, (64, 64)), (12, 3, 2, (40,)), (64, 1, 2, (32, 48, 16))])
def test_c_literal_network_equals_numpy_shared_layer0_form(oc, oracle, nt, nr, npkt, hidden):
    """The algebra the HIP path rests on (the first layer's product with the preamble is shared by the Nt pairs of an rx
    antenna, the pilot rows become a constant table) is what csi_oracle.predict_packets_shared computes - the checker of the Nt = 128
    GPU tests.  Here it is held against the C statement's LITERAL network (every sample row through the whole first layer)."""
    rng = np.random.default_rng(nt + len(hidden))
    P = rng.choice([-1.0, 1.0], (nt, nt))
    ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
    w_re = oracle.make_weights(rng, 321 * nt, list(hidden), 234)
    w_im = oracle.make_weights(rng, 321 * nt, list(hidden), 234)
    got = oc.predict_packets(ltf, P, w_re, w_im)
    want = oracle.predict_packets_shared(ltf, P, w_re, w_im, np.float64)
    for a, b in zip(got, want):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-11 * np.abs(b).max())
