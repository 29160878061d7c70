"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors that were produced
by running the reference's own numpy code (tests/golden/make_golden.py), and against exact
identities for the rows the reference cannot pin (LS, Dense/BN)."""
import os

import numpy as np
import pytest


def _dataset_from_golden(g):
    keys = g['ds_keys'].tolist()
    ltf = {k: {'real': g['ds_ltf_real'][i], 'imag': g['ds_ltf_imag'][i]} for i, k in enumerate(keys)}
    nt, nr = int(g['nt']), int(g['nr'])
    sim = {'nTX': nt, 'nRX': nr, 'lenLTF': 320 * nt, 'nSubCarr': 234}
    return {'X': g['ds_X'], 'y': {'real': g['ds_y_real'], 'imag': g['ds_y_imag']}, 'LTF': ltf, 'P': g['ds_P'],
            'simParams': sim}


def test_sample_assembly_matches_reference_datagenerator(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, 'ref_datagen_nt4.npz'))
    ds = _dataset_from_golden(g)
    nt, nr, npkt = int(g['nt']), int(g['nr']), int(g['npkt'])
    bs = nt * nr
    for d in ('real', 'imag'):
        assert int(g[f'{d}_len']) == npkt            # one batch per packet (DNN.py:339)
        assert int(g[f'{d}_len_bs5']) == (npkt * bs) // 5
        for b in range(npkt):
            ids = list(range(b * bs, (b + 1) * bs))
            xsig, xp, y = oracle.assemble_batch(ds, d, ids)
            np.testing.assert_array_equal(xsig, g[f'{d}_Xsig'][b])
            np.testing.assert_array_equal(xp, g[f'{d}_Xp'][b])
            np.testing.assert_array_equal(y, g[f'{d}_y'][b])


def test_packed_layout_equals_reference_sample_order(oracle, golden_dir):
    """[Npkt,Nr,lenLTF] planes + P rows reproduce the reference's [Xsig | Xp] rows in dataset
    order s = p*Nr*Nt + r*Nt + t, including the P orientation (P_py[:, t] = MATLAB P(t,:))."""
    g = np.load(os.path.join(golden_dir, 'ref_datagen_nt4.npz'))
    nt, nr, npkt = int(g['nt']), int(g['nr']), int(g['npkt'])
    ltf = (g['ds_ltf_real'] + 1j * g['ds_ltf_imag']).reshape(npkt, nr, 320 * nt)
    P_rows = oracle.pilot_rows_from_dataset_P(g['ds_P'])
    np.testing.assert_array_equal(P_rows, g['P_matlab'])
    for d in ('real', 'imag'):
        x = oracle.samples_from_packets(ltf, P_rows, d)
        ref = np.concatenate([g[f'{d}_Xsig'][..., 0], g[f'{d}_Xp']], axis=-1).reshape(npkt * nr * nt, -1)
        np.testing.assert_array_equal(x, ref)
    # transposed P must NOT match (the fixture's P is deliberately non-symmetric)
    assert not np.array_equal(oracle.samples_from_packets(ltf, P_rows.T, 'real'), ref)


def test_inference_recombine_and_postprocess_match_reference(oracle, golden_dir):
    g = np.load(os.path.join(golden_dir, 'ref_inference_rice.npz'), allow_pickle=True)
    x = g['x']
    out_re = (x.real.astype(np.float32) @ g['A_re'].astype(np.float32) + g['b_re'].astype(np.float32)).astype(np.float32)
    out_im = (x.imag.astype(np.float32) @ g['A_im'].astype(np.float32) + g['b_im'].astype(np.float32)).astype(np.float32)
    y = oracle.postprocess_rice_renew(oracle.recombine(out_re, out_im))
    assert str(y.dtype) == str(g['y_dtype'])
    np.testing.assert_allclose(y, g['y'], rtol=0, atol=1e-5)       # float32 matmul order only
    np.testing.assert_array_equal(y == 0, g['y'] == 0)              # null pattern exact
    post = oracle.postprocess_rice_renew(g['ramp'])
    np.testing.assert_array_equal(post, g['post_ramp'])
    assert str(post.dtype) == str(g['post_ramp_dtype'])
    assert int(g['exit_bad_dtype']) == -1 and int(g['exit_bad_width']) == -1
    with pytest.raises(ValueError):
        oracle.preprocess_rice_renew(x.astype(np.complex64))
    with pytest.raises(ValueError):
        oracle.postprocess_rice_renew(np.zeros((2, 51), dtype=np.complex64))
    # the reference hands float64 planes and batch_size=bs to the models
    assert list(g['model_calls_real']) == ['<f8', str(x.shape[0])]


def _reference_spectra(g):
    """Complex per-symbol spectra [npkt*nr, 256, nt] as the REFERENCE computed them (un-shifted FFT bin order):
    its 'reshape' method transforms the real and the imaginary plane of a preamble separately
    (massiveMIMO_dataGenerator.py:436,452 with d = 'real' / 'imag'); the transform of the complex preamble is
    their sum F(re) + j F(im) (linearity - the only step taken here)."""
    return g['real_fft_pre_shift'] + 1j * g['imag_fft_pre_shift']


def test_ofdm_demod_matches_reference_reshape_method(oracle, golden_dir):
    """Pins row a-1 (OFDM demodulation convention) to the reference's own numpy statement of it,
    massiveMIMO_dataGenerator.py:425-453, executed by tests/golden/make_golden.py: column-major split into
    Nt symbols of 320 samples (:437-439), CP removal = samples 64..319 of each symbol (:442-443), un-scaled
    256-point FFT along the symbol (:452), DC moved to the middle of the frequency axis (:453)."""
    g = np.load(os.path.join(golden_dir, 'ref_ofdm_reshape_nt4.npz'))
    nt, nr, npkt = int(g['nt']), int(g['nr']), int(g['npkt'])
    # what the reference did, as recorded from its own variables
    assert g['noCP_ix'].tolist() == list(range(64, 320))
    for d in ('real', 'imag'):
        ltf = g[f'ds_ltf_{d}']
        np.testing.assert_array_equal(g[f'{d}_input2D'], ltf.reshape(npkt * nr, nt, 320).transpose(0, 2, 1))     # order='F' split
        np.testing.assert_array_equal(g[f'{d}_afterCPRemoval'], g[f'{d}_input2D'][:, 64:320, :])
    spec = _reference_spectra(g)                                         # [pr, 256 bins, nt symbols]
    ltf = g['ds_ltf_real'] + 1j * g['ds_ltf_imag']                       # [pr, 1280]
    # the oracle: same window, same transform, DC in the middle, then the 234 data bins
    rx = oracle.ofdm_demod(ltf, nt)                                      # [pr, 234, nt]
    want = np.fft.fftshift(spec, axes=1)[:, oracle.data_carrier_indices() - 1, :]
    np.testing.assert_allclose(rx, want, rtol=0, atol=1e-11 * np.abs(want).max())
    # the known defect of the reference's line :453 (np.fft.fftshift without axes also rotates the SYMBOL
    # axis by Nt/2): its recorded output equals the frequency shift the oracle applies plus that rotation
    for d in ('real', 'imag'):
        pre, post = g[f'{d}_fft_pre_shift'], g[f'{d}_fft_post_shift']
        np.testing.assert_array_equal(post, np.roll(np.fft.fftshift(pre, axes=1), nt // 2, axis=2))
        # ... and the rows it hands to the network keep the real part of column iTx of that array (:454-455)
        X = g[f'{d}_X'].reshape(npkt * nr, nt, -1)
        for t in range(nt):
            np.testing.assert_array_equal(X[:, t, :256], post[:, :, t].real)
            np.testing.assert_array_equal(X[:, t, 256:256 + nt], np.broadcast_to(g['ds_P'][:, t], (npkt * nr, nt)))
        np.testing.assert_array_equal(X[:, :, 256 + nt:], np.broadcast_to(g['ltf_freqdom'], (npkt * nr, nt, 234)))


def test_ls_estimate_on_reference_spectra(oracle, golden_dir):
    """LS (a-2) fed with the reference-computed spectra equals the oracle's full time-domain LS path on the
    same preambles - the OFDM half of ls_estimate is the pinned one."""
    g = np.load(os.path.join(golden_dir, 'ref_ofdm_reshape_nt4.npz'))
    nt = int(g['nt'])
    rx = np.fft.fftshift(_reference_spectra(g), axes=1)[:, oracle.data_carrier_indices() - 1, :]
    h_from_ref = np.swapaxes(oracle.ls_from_rxsym(rx, g['P_matlab']), -1, -2)          # [pr, nt, 234]
    h = oracle.ls_estimate(g['ds_ltf_real'] + 1j * g['ds_ltf_imag'], g['P_matlab'])
    np.testing.assert_allclose(h, h_from_ref, rtol=0, atol=1e-11 * np.abs(h_from_ref).max())


def test_postprocess_index_map(oracle):
    o = np.arange(1, 53)[None, :].astype(np.complex128)
    y = oracle.postprocess_rice_renew(o).real.astype(int)[0]
    assert y[0] == 0 and list(y[1:27]) == list(range(27, 53)) and not y[27:38].any() and list(y[38:]) == list(range(1, 27))


def test_ofdm_constants(oracle):
    idx = oracle.data_carrier_indices()
    assert idx.size == 234 and idx[0] == 8 and idx[-1] == 250 and 129 not in idx and 26 not in idx
    ltf = oracle.vht_ltf_256()
    assert ltf.size == 256 and not ltf[:7].any() and not ltf[-6:].any() and ltf[128] == 0
    assert set(np.unique(ltf[idx - 1]).tolist()) == {-1.0, 1.0}


@pytest.mark.parametrize('nt', [4, 8, 32])
def test_ls_known_answer(oracle, nt):
    """LS(x) == H exactly (to fp64 rounding) when the packet is synthesised from H and
    P P^H = Nt I (follows from helperMIMOChannelEstimate.m:24-36)."""
    rng = np.random.default_rng(nt)
    P = oracle.hadamard(nt)
    perm = rng.permutation(nt)
    P = P[perm] * rng.choice([-1.0, 1.0], nt)[:, None]              # non-symmetric, still orthogonal
    ltf, H = oracle.make_structured_packets(rng, 2, 3, P)
    np.testing.assert_allclose(oracle.ls_estimate(ltf, P), H, rtol=0, atol=1e-12)
    if nt > 4:
        assert np.abs(oracle.ls_estimate(ltf, P.T) - H).max() > 1e-3  # orientation matters


def test_ls_is_linear_and_ignores_cp(oracle):
    rng = np.random.default_rng(5)
    nt = 8
    P = oracle.hadamard(nt)
    a = rng.standard_normal((2, 1, 320 * nt)) + 1j * rng.standard_normal((2, 1, 320 * nt))
    b = rng.standard_normal((2, 1, 320 * nt)) + 1j * rng.standard_normal((2, 1, 320 * nt))
    lhs = oracle.ls_estimate(2.0 * a - 3j * b, P)
    np.testing.assert_allclose(lhs, 2.0 * oracle.ls_estimate(a, P) - 3j * oracle.ls_estimate(b, P), atol=1e-10)
    a2 = a.copy().reshape(2, 1, nt, 320)
    a2[..., :64] = 0                                                 # the CP samples are not used
    np.testing.assert_allclose(oracle.ls_estimate(a2.reshape(a.shape), P), oracle.ls_estimate(a, P), atol=1e-12)


def test_fc_forward_against_torch_layers(oracle):
    """Dense/BN semantics cross-checked against an independent implementation
    (torch.nn.functional.linear / batch_norm in eval mode, eps=1e-3)."""
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(11)
    nt = 4
    w = oracle.make_weights(rng, 320 * nt + nt, [48, 40], 234, use_bn=True, dtype=np.float64)
    x = rng.standard_normal((9, 320 * nt + nt))
    y = oracle.fc_forward(x, w, np.float64)
    h = torch.from_numpy(x)
    for i in range(2):
        h = F.relu(F.linear(h, torch.from_numpy(w[f'fc_dense{i}.kernel'].T.copy()), torch.from_numpy(w[f'fc_dense{i}.bias'])))
        h = F.batch_norm(h, torch.from_numpy(w[f'bn{i}.moving_mean']), torch.from_numpy(w[f'bn{i}.moving_variance']),
                         torch.from_numpy(w[f'bn{i}.gamma']), torch.from_numpy(w[f'bn{i}.beta']), training=False, eps=1e-3)
    yt = F.linear(h, torch.from_numpy(w['fc_regressor.kernel'].T.copy()), torch.from_numpy(w['fc_regressor.bias'])).numpy()
    np.testing.assert_allclose(y, yt, rtol=1e-12, atol=1e-12)


def test_fc_forward_identities(oracle):
    rng = np.random.default_rng(3)
    nt = 4
    d_in = 320 * nt + nt
    w = oracle.make_weights(rng, d_in, [16], 234, use_bn=False, dtype=np.float64)
    x = rng.standard_normal((5, d_in))
    w0 = {k: (np.zeros_like(v) if k.endswith('kernel') else v) for k, v in w.items()}
    # zero kernels: output = regressor bias
    np.testing.assert_allclose(oracle.fc_forward(x, w0), np.broadcast_to(w['fc_regressor.bias'], (5, 234)))
    # fp32 evaluation stays within the fp32 contract of the fp64 one
    w32 = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) else v) for k, v in w.items()}
    assert oracle.row_rel_err(oracle.fc_forward(x, w32, np.float32), oracle.fc_forward(x, w32, np.float64)) < 1e-5


def test_predict_packets_layout(oracle):
    """Row s of the literal predict equals H[p, r, t] with s = p*Nr*Nt + r*Nt + t."""
    rng = np.random.default_rng(8)
    nt, nr, npkt = 4, 2, 3
    P = rng.integers(-2, 3, (nt, nt)).astype(np.float64)
    w_re = oracle.make_weights(rng, 320 * nt + nt, [24, 24], 234)
    w_im = oracle.make_weights(rng, 320 * nt + nt, [24, 24], 234)
    ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
    o_re, o_im = oracle.predict_packets(ltf, P, w_re, w_im)
    x = oracle.samples_from_packets(ltf, P, 'imag')
    y = oracle.fc_forward(x, w_im)
    p, r, t = 2, 1, 3
    np.testing.assert_allclose(o_im[p, r, t], y[p * nr * nt + r * nt + t], rtol=1e-12)
    assert o_re.shape == (npkt, nr, nt, 234)


def test_shared_layer0_oracle_equals_literal(oracle):
    rng = np.random.default_rng(21)
    nt, nr, npkt = 8, 2, 3
    P = rng.integers(-2, 3, (nt, nt)).astype(np.float64)
    w_re = oracle.make_weights(rng, 320 * nt + nt, [24, 16, 20], 234)
    w_im = oracle.make_weights(rng, 320 * nt + nt, [24], 234, use_bn=False)
    ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
    a = oracle.predict_packets(ltf, P, w_re, w_im)
    b = oracle.predict_packets_shared(ltf, P, w_re, w_im)
    np.testing.assert_allclose(a[0], b[0], rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-11, atol=1e-11)


def test_nmse(oracle):
    rng = np.random.default_rng(2)
    h = rng.standard_normal((2, 2, 4, 234)) + 1j * rng.standard_normal((2, 2, 4, 234))
    assert oracle.nmse_subk(h, h) == 0.0
    assert abs(oracle.nmse_subk(h, 0.9 * h) - 0.01) < 1e-12


def _oracle_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, 'oracle_nt4.npz'))
    w = {tag: {k.split('.', 1)[1]: g[k] for k in g.files if k.startswith(f'w_{tag}.')} for tag in ('re', 'im')}
    return g, w['re'], w['im']


def test_oracle_reproduces_its_committed_fixture(oracle, golden_dir):
    """The oracle's own outputs for a seeded Nt=4 problem are committed (tests/golden/oracle_nt4.npz,
    written by make_oracle_fixture.py): an edit that changes what the oracle computes shows up here."""
    g, w_re, w_im = _oracle_fixture(golden_dir)
    ls = oracle.ls_estimate(g['ltf'], g['P'])
    np.testing.assert_allclose(ls, g['ls'], rtol=0, atol=1e-12 * np.abs(g['ls']).max())
    o_re, o_im = oracle.predict_packets(g['ltf'], g['P'], w_re, w_im, np.float64, pkt_batch=3)
    np.testing.assert_allclose(o_re, g['dnn_real'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(o_im, g['dnn_imag'], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(oracle.recombine(o_re, o_im), g['csi'])


def test_split_f16_operands_are_fp32_grade(oracle):
    """The GPU's split-f16 engine carries fp32 operands as hi + lo float16 halves and drops lo*lo
    (gemm_hs.hip.h).  Its representation error, isolated here with fp64 accumulation, must sit far
    below the 1e-5 contract - at the level of fp32 rounding itself - over a wide range of magnitudes."""
    rng = np.random.default_rng(0)
    x = rng.standard_normal(4096).astype(np.float32)
    for shift in (-6, 0, 4, 8):
        hi, lo = oracle.split_f16(x, shift)
        err = np.abs((hi + lo) * 2.0 ** -shift - x.astype(np.float64))
        big = np.abs(x) * 2.0 ** shift >= 2.0 ** -3            # lo is a normal float16 there
        assert np.all(err[big] <= 2.0 ** -22 * np.abs(x[big]))
        assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(x), 2.0 ** -25 * 2.0 ** -shift))
    nt = 4
    w = oracle.make_weights(rng, 321 * nt, [64, 48], 234)
    for gain in (1e-2, 1.0, 50.0):
        xs = (gain * rng.standard_normal((24, 321 * nt))).astype(np.float32)
        ref = oracle.fc_forward(xs, w, np.float64)
        got = oracle.fc_forward_split_f16(xs, w)
        f32 = oracle.fc_forward(xs, w, np.float32)
        e_split = oracle.row_rel_err(got, ref)
        assert e_split < 5e-7, (gain, e_split)
        assert e_split < 3 * max(oracle.row_rel_err(f32, ref), 1e-7)       # no worse than plain fp32 evaluation


def test_oracle_computes_in_double_whatever_it_is_handed(oracle):
    """The checker's results must not depend on the precision of the arrays a test hands it: numpy >= 2 transforms complex64 input in
    single precision (the LS oracle did just that until the C statement, oracle/csi_oracle_c.c, disagreed with it at 1e-7)."""
    rng = np.random.default_rng(77)
    nt, nr = 8, 2
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, 3, nr, P, snr_db=0.0)[0].astype(np.complex64)
    wide = ltf.astype(np.complex128)
    a, b = oracle.ls_estimate(ltf, P), oracle.ls_estimate(wide, P)
    assert a.dtype == np.complex128 and np.array_equal(a, b)
    assert oracle.ofdm_demod(ltf, nt).dtype == np.complex128
    w_re = oracle.make_weights(rng, 321 * nt, [32, 16], 234)
    w_im = oracle.make_weights(rng, 321 * nt, [32, 16], 234)
    for fn in (oracle.predict_packets, oracle.predict_packets_shared):
        x, y = fn(ltf, P, w_re, w_im, np.float64), fn(wide, P, w_re, w_im, np.float64)
        assert x[0].dtype == np.float64 and np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    h = a.astype(np.complex64)
    est = (h * np.complex64(0.97)).astype(np.complex64)
    assert oracle.nmse_subk(h, est) == oracle.nmse_subk(h.astype(np.complex128), est.astype(np.complex128))
