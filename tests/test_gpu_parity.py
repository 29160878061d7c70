"""GPU parity tests (-m gpu): every call goes through the C-ABI of libcsi_mamimo.so and is
checked against the numpy oracle on identical seeded inputs.

Tolerance: the contract is 1e-5 norm-relative per output row in fp32 (BASELINE.json north_star;
SURVEY.md section 7 'hard parts' defines the norm-relative form), measured against the fp64
evaluation of the oracle.  TOL below is that number; nothing is loosened per test."""
import os

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


def _pilot(rng, nt, orthogonal=True):
    from oracle import csi_oracle as o
    if orthogonal:
        P = o.hadamard(nt)
        return (P[rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[:, None]).astype(np.float64)
    return rng.integers(-3, 4, (nt, nt)).astype(np.float64)


def _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn=True, n_out=234, **kw):
    e = pkg.CsiEngine(nt, nr, hidden=hidden, n_out=n_out, use_bn=use_bn, **kw)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    return e


# ------------------------------------------------------------------------------------ LS
@pytest.mark.parametrize('nt,nr,npkt', [(4, 2, 3), (8, 1, 5), (16, 3, 2), (32, 4, 4), (64, 4, 2), (128, 2, 2)])
def test_ls_matches_oracle_and_known_channel(pkg, oracle, nt, nr, npkt):
    rng = np.random.default_rng(100 + nt)
    P = _pilot(rng, nt)
    ltf, H = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=None)
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    e.set_pilot(P)
    h = e.ls_estimate(ltf)
    assert h.shape == (npkt, nr, nt, 234) and h.dtype == np.complex64
    ref = oracle.ls_estimate(ltf, P)
    assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    # known-answer: noiseless structured packet -> LS == H
    assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([H.real, H.imag], -1)) < TOL


def test_ls_kernel_on_reference_ofdm_fixture(pkg, oracle, golden_dir):
    """The LS kernels against spectra the REFERENCE computed: tests/golden/ref_ofdm_reshape_nt4.npz holds the
    per-symbol FFTs recorded from massiveMIMO_dataGenerator.py:425-453 (method 'reshape') on the Nt=4 fixture
    preambles.  Despreading those spectra (helperMIMOChannelEstimate.m:24-36, non-symmetric P) must give what
    the HIP kernels compute from the time-domain preambles - symbol split, CP window, FFT and bin order of
    the kernels are thereby checked against reference-executed output, for every LS kernel that serves Nt=4."""
    g = np.load(os.path.join(golden_dir, 'ref_ofdm_reshape_nt4.npz'))
    nt, nr, npkt = int(g['nt']), int(g['nr']), int(g['npkt'])
    spec = g['real_fft_pre_shift'] + 1j * g['imag_fft_pre_shift']                       # [pr, 256, nt], FFT bin order
    rx = np.fft.fftshift(spec, axes=1)[:, oracle.data_carrier_indices() - 1, :]
    want = np.swapaxes(oracle.ls_from_rxsym(rx, g['P_matlab']), -1, -2).reshape(npkt, nr, nt, 234)
    want2 = np.concatenate([want.real, want.imag], -1)
    ltf = (g['ds_ltf_real'] + 1j * g['ds_ltf_imag']).reshape(npkt, nr, 320 * nt)
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    e.set_pilot(g['P_matlab'])
    for kernel in (0, 1, 3):                                                            # automatic, FFT-first, despread-first
        e.set_option('ls_kernel', kernel)
        h = e.ls_estimate(ltf)
        assert rel_rows(np.concatenate([h.real, h.imag], -1), want2) < TOL, kernel


def test_ls_noisy_generic_pilot_and_linearity(pkg, oracle):
    rng = np.random.default_rng(7)
    nt, nr, npkt = 8, 2, 6
    P = _pilot(rng, nt, orthogonal=False)            # generic real P: no Hadamard assumption
    a = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=0.0)[0]
    b = (rng.standard_normal(a.shape) + 1j * rng.standard_normal(a.shape))
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    e.set_pilot(P)
    ha, hb = e.ls_estimate(a), e.ls_estimate(b)
    ref = oracle.ls_estimate(a, P)
    assert rel_rows(np.concatenate([ha.real, ha.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    hab = e.ls_estimate(2.0 * a - 0.5 * b)
    lin = 2.0 * ha.astype(np.complex128) - 0.5 * hb.astype(np.complex128)
    assert rel_rows(np.concatenate([hab.real, hab.imag], -1), np.concatenate([lin.real, lin.imag], -1)) < 5e-6


@pytest.mark.parametrize('nt', [12, 72, 96])
def test_ls_non_power_of_two_nt_generic_pilot(pkg, oracle, nt):
    """Nt that is not a power of two (partial 32-row MFMA tiles, and for Nt > 64 the
    despread-first kernel with a partial last chunk), generic real P."""
    rng = np.random.default_rng(nt)
    P = rng.integers(-2, 3, (nt, nt)).astype(np.float64)
    ltf = rng.standard_normal((2, 2, 320 * nt)) + 1j * rng.standard_normal((2, 2, 320 * nt))
    e = pkg.CsiEngine(nt, 2, hidden=(8,))
    e.set_pilot(P)
    h = e.ls_estimate(ltf)
    ref = oracle.ls_estimate(ltf.astype(np.complex64), P)
    assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL


@pytest.mark.parametrize('kernel', [1, 2, 3, 6])
def test_ls_all_kernels_agree(pkg, oracle, kernel):
    """The generic LS kernels (1 FFT-first, 2 chunked FFT-first, 3 despread-first, 6 LDS-DMA ring) against the oracle on
    every Nt each of them serves, incl. partial MFMA tiles and partial symbol chunks (Nt = 40, 72, 100)
    and many more items than resident workgroups (persistent loops)."""
    rng = np.random.default_rng(kernel + 3)
    cases = {1: ((8, 2, 3), (32, 3, 3), (64, 2, 3), (40, 1, 2)),
             2: ((40, 2, 3), (64, 2, 3), (72, 1, 2), (96, 2, 2), (100, 1, 2), (128, 2, 2), (64, 4, 300)),
             3: ((8, 2, 3), (64, 2, 3), (72, 1, 2), (128, 2, 2), (160, 1, 1)),
             # 6 = the chunked kernel's successor on the LDS-DMA ring (generic P): every antenna-tile count, partial last chunks
             6: ((16, 2, 5), (24, 2, 3), (32, 3, 300), (40, 2, 3), (64, 2, 3), (72, 1, 2), (96, 2, 2), (100, 1, 2), (128, 2, 2),
                 (64, 4, 300), (128, 3, 100))}[kernel]
    for nt, nr, npkt in cases:
        P = _pilot(rng, nt) if nt & (nt - 1) == 0 else rng.integers(-2, 3, (nt, nt)).astype(np.float64)
        if npkt > 10:
            ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
        else:
            ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
        e = pkg.CsiEngine(nt, nr, hidden=(8,))
        e.set_option('ls_kernel', kernel)
        e.set_pilot(P)
        h = e.ls_estimate(ltf)
        sel = slice(None) if npkt <= 10 else np.r_[0:2, npkt - 2:npkt]
        ref = oracle.ls_estimate(np.asarray(ltf)[sel].astype(np.complex64), P)
        assert rel_rows(np.concatenate([h[sel].real, h[sel].imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL, (nt, nr, npkt)
        if npkt > 10:       # persistent walk: every item written, and identical on a second run
            assert np.isfinite(h.view(np.float32)).all() and np.abs(h).sum(axis=(1, 2, 3)).min() > 0
            assert np.array_equal(h, e.ls_estimate(ltf))


@pytest.mark.parametrize('pieces', [1, 2, 3])
def test_ls_bf16_split_despread(pkg, oracle, pieces):
    """ls_kernel 7: the generic-P ring kernel with the despread on v_mfma_f32_32x32x16_bf16, fp32 values cut exactly into
    three bf16 pieces.  Pilot matrices that need 1 (+-1 entries), 2 (16 significand bits) and 3 (arbitrary floats) pieces,
    every antenna-tile count, partial last chunks (Nt = 24, 40, 72, 100), persistent walks, extreme amplitudes (bf16 has
    fp32's exponent range: no scaling is involved), and the automatic choice for a non-Hadamard pilot."""
    rng = np.random.default_rng(70 + pieces)
    cases = ((16, 2, 5), (24, 2, 3), (32, 3, 300), (40, 2, 3), (64, 2, 3), (72, 1, 2), (96, 2, 2), (100, 1, 2), (128, 2, 2), (64, 4, 300), (128, 3, 100),
             (32, 4, 1000), (24, 4, 700), (96, 4, 200))
    for nt, nr, npkt in cases:
        if pieces == 1:
            P = rng.choice([-1.0, 1.0], (nt, nt))
        else:
            P = np.linalg.qr(rng.standard_normal((nt, nt)))[0].astype(np.float32) * np.float32(np.sqrt(nt))
            if pieces == 2:
                P = (P.view(np.uint32) & np.uint32(0xffffff00)).view(np.float32)
            P = P.astype(np.float64)
        if npkt > 10:
            ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
        else:
            ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
            ltf[0] *= 1e-18                      # far below / above anything an f16 scheme could hold
            ltf[-1] *= 1e15
        e = pkg.CsiEngine(nt, nr, hidden=(8,))
        e.set_pilot(P)
        assert e.get_option('ls_pilot_pieces') == pieces
        assert e.get_option('ls_mode') == (6 if nt <= 32 else 7), (nt, pieces)   # the automatic choice (round 4: the bf16-split kernel from Nt = 33)
        if nt <= 32 and npkt * nr > 256:
            # the one-antenna-tile form of kernel 7 with TWO workgroups per CU is not selected any more (rare wrong first items of a CU's
            # second workgroup on one box of the pool, open: DESIGN.md 4.2) - what the library runs for this shape is checked instead
            h = e.ls_estimate(ltf)
            ref = oracle.ls_estimate(np.asarray(ltf).astype(np.complex64), P)
            assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL, (nt, nr, npkt)
            for _ in range(4):
                assert np.array_equal(h, e.ls_estimate(ltf))
            continue
        e.set_option('ls_kernel', 7)
        h = e.ls_estimate(ltf)
        ref = oracle.ls_estimate(np.asarray(ltf).astype(np.complex64), P)          # EVERY item against the oracle
        assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL, (nt, nr, npkt)
        e.set_option('ls_kernel', 6)             # the fp32 matrix-core despread: same answer to rounding
        h6 = e.ls_estimate(ltf)
        assert not np.array_equal(h, h6)
        assert rel_rows(np.concatenate([h.real, h.imag], -1).reshape(-1, 468), np.concatenate([h6.real, h6.imag], -1).reshape(-1, 468)) < 2e-6
        e.set_option('ls_kernel', 7)
        e.set_option('ls_v2', 1)                 # the other ring depth of the same kernel: the same arithmetic per (bin, antenna)
        assert np.array_equal(h, e.ls_estimate(ltf))
        e.set_option('ls_v2', 0)
        if npkt > 10:                            # persistent walk, repeated: a race shows up as a run that differs
            for _ in range(4):
                assert np.array_equal(h, e.ls_estimate(ltf))


@pytest.mark.parametrize('nt,nr,npkt', [(16, 2, 5), (32, 3, 300), (64, 2, 7), (128, 2, 3), (16, 4, 400), (64, 4, 300), (128, 2, 200)])
def test_ls_walsh_hadamard_despread(pkg, oracle, nt, nr, npkt):
    """With the Sylvester Hadamard pilot matrix the LS despread is a fast Walsh-Hadamard transform
    (chosen automatically): same answer as the oracle and as the MFMA despread (ls_kernel 2) up to the
    summation order; a P that is not exactly that matrix must not take it."""
    rng = np.random.default_rng(nt + npkt)
    P = oracle.hadamard(nt)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    e.set_pilot(P)
    h = e.ls_estimate(ltf)                       # automatic -> Walsh-Hadamard kernel
    sel = slice(None) if npkt <= 10 else np.r_[0:2, npkt - 2:npkt]
    ref = oracle.ls_estimate(ltf[sel], P)
    assert rel_rows(np.concatenate([h[sel].real, h[sel].imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    e.set_option('ls_kernel', 2)
    h2 = e.ls_estimate(ltf)                      # MFMA despread
    assert not np.array_equal(h, h2)             # really another kernel (summation order differs) ...
    assert rel_rows(np.concatenate([h.real, h.imag], -1).reshape(-1, 468), np.concatenate([h2.real, h2.imag], -1).reshape(-1, 468)) < 2e-6
    e.set_option('ls_kernel', 0)
    assert np.array_equal(h, e.ls_estimate(ltf))                 # deterministic
    # the two generations of the kernel (4: register prefetch, 5 = automatic: LDS-DMA ring) and both shapes of the
    # second (16- and 8-symbol chunks group the additions differently) agree to rounding; every item is written
    for opt, val in (('ls_kernel', 4), ('ls_v2', 1)):
        e.set_option('ls_kernel', 0)
        e.set_option(opt, val)
        h4 = e.ls_estimate(ltf)
        assert rel_rows(np.concatenate([h.real, h.imag], -1).reshape(-1, 468), np.concatenate([h4.real, h4.imag], -1).reshape(-1, 468)) < 1e-6, (opt, val)
    e.set_option('ls_v2', 0)
    e.set_option('ls_kernel', 0)
    # a pilot matrix that differs from the Sylvester matrix in one sign: generic path, still right
    P2 = P.copy()
    P2[3, 5] = -P2[3, 5]
    e2 = pkg.CsiEngine(nt, nr, hidden=(8,))
    e2.set_pilot(P2)
    e2.set_option('ls_kernel', 4)                # forcing it is refused silently (falls back)
    g = e2.ls_estimate(ltf[:2])
    ref2 = oracle.ls_estimate(ltf[:2], P2)
    assert rel_rows(np.concatenate([g.real, g.imag], -1), np.concatenate([ref2.real, ref2.imag], -1)) < TOL


@pytest.mark.parametrize('nt,nr,npkt', [(4, 2, 2), (32, 2, 1), (40, 1, 1)])
def test_lmmse_matches_reference_formula(pkg, oracle, nt, nr, npkt):
    """LMMSE smoothing (LMMSE_ce.m, one 234x234 inverse per link in the reference; one Levinson solve
    per (packet, rx) here) against the literal restatement of the reference formula in fp64."""
    rng = np.random.default_rng(900 + nt)
    P = oracle.hadamard(nt) if nt & (nt - 1) == 0 else rng.integers(-2, 3, (nt, nt)).astype(np.float64)
    ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    e.set_pilot(P)
    h_ls = e.ls_estimate(ltf)
    hvec = np.sort(np.abs(rng.standard_normal((npkt, 100)))).astype(np.float32) * 1e-7     # like the delays h_tau
    snr_db = rng.choice([-10.0, 0.0, 10.0, 25.0], size=(npkt, nr)).astype(np.float32)
    got = e.lmmse_estimate(h_ls, hvec, snr_db)
    assert got.shape == h_ls.shape and got.dtype == np.complex64
    ref = oracle.lmmse_estimate(h_ls, hvec.astype(np.float64), snr_db.astype(np.float64))
    assert rel_rows(np.concatenate([got.real, got.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    # smoothing must actually change the estimate, and shrink it
    assert np.linalg.norm(got - h_ls) > 1e-2 * np.linalg.norm(h_ls)
    assert np.linalg.norm(got) < np.linalg.norm(h_ls)


def test_ls_empty_batch(pkg):
    e = pkg.CsiEngine(4, 2, hidden=(8,))
    e.set_pilot(np.eye(4))
    h = e.ls_estimate(np.zeros((0, 2, 1280), dtype=np.complex128))
    assert h.shape == (0, 2, 4, 234)


# ------------------------------------------------------------------------------------ DNN
CASES = [
    # nt, nr, npkt, hidden, use_bn
    (4, 2, 3, (64, 64), True),          # the small fixture shape of SURVEY.md 8c
    (4, 2, 37, (64, 64), True),         # ragged: M2 = 296 rows, not a multiple of the 128 tile
    (8, 3, 5, (100, 36), True),         # widths that are not multiples of the 32 / 128 tiles
    (8, 2, 4, (64,), True),             # single hidden layer (regressor fed by the pair prologue)
    (4, 1, 6, (32, 48, 40), True),      # three hidden layers (ping-pong buffers)
    (8, 2, 4, (64, 64), False),         # --useBN off
    (32, 4, 2, (1024, 1024), True),     # the shipped model (pipe.sh:40,47), 2 packets
    (64, 2, 3, (64, 32), True),         # BASELINE configs 3/4 antenna count
    (128, 2, 2, (64, 64), True),        # BASELINE config 5 antenna count (two T pieces per wave)
    (12, 2, 7, (40,), True),            # Nt not a power of two
    (8, 2, 3, (512, 320), True),        # small batch, K >= 256: split-K latency path of every layer
    (8, 2, 1, (256,), False),           # one packet, single hidden layer, no BN, split-K regressor
]


@pytest.mark.parametrize('nt,nr,npkt,hidden,use_bn', CASES)
def test_predict_matches_fp64_oracle(pkg, oracle, nt, nr, npkt, hidden, use_bn):
    rng = np.random.default_rng(nt * 1000 + npkt)
    w_re, w_im = _weights(oracle, 1234 + nt, nt, hidden, use_bn)
    pow2 = (nt & (nt - 1)) == 0
    P = _pilot(rng, nt, orthogonal=(pow2 and nt != 8))
    if pow2:
        ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=5.0)[0]
    else:
        ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn)
    o_re, o_im = e.predict(ltf)
    assert o_re.shape == (npkt, nr, nt, 234) and o_re.dtype == np.float32
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(o_re, r_re) < TOL
    assert rel_rows(o_im, r_im) < TOL
    # accuracy metric of the reference evaluation (BER_test_maMIMO_LTF.m:675-686)
    assert oracle.nmse_subk(r_re + 1j * r_im, o_re + 1j * o_im) < 1e-10


@pytest.mark.parametrize('tile', ['128', '256'])
@pytest.mark.parametrize('nt,nr,npkt,hidden', [(4, 2, 70, (64, 48)), (32, 2, 9, (96, 64)), (128, 1, 3, (64, 64)),
                                               (12, 3, 11, (40, 24)), (8, 2, 5, (64,)), (8, 1, 40, (72, 136, 200))])
def test_both_pair_tile_kernels(pkg, oracle, monkeypatch, tile, nt, nr, npkt, hidden):
    """Every GEMM has a 128-row and a 256-row tile kernel (chosen by grid size); force each one on
    ragged row counts, every (T pieces, L pieces) template variant of the pair kernel, and the
    plain kernels of layer 0 / hidden layers / regressor."""
    monkeypatch.setenv('CSI_FORCE_PAIR_TILE', tile)
    rng = np.random.default_rng(nt + npkt)
    w_re, w_im = _weights(oracle, 77 + nt, nt, hidden)
    P = _pilot(rng, nt, orthogonal=False)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    o_re, o_im = e.predict(ltf)
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL


@pytest.mark.parametrize('tile', [128, 256])
@pytest.mark.parametrize('nt,nr,npkt,hidden', [(4, 3, 47, (200, 72)), (8, 2, 70, (1100, 40, 300))])
def test_xcd_super_tile_order_is_a_bijection(pkg, oracle, tile, nt, nr, npkt, hidden):
    """Plain GEMMs can walk their output tiles in an XCD-aware super-tile order (ragged edges map
    outside the matrix and exit).  Forced here on ragged tile counts (1, 2, 3 and 9 column tiles;
    row-tile counts that are not multiples of the super-tile height): every output must still be
    produced exactly once, i.e. match the oracle."""
    rng = np.random.default_rng(nt * npkt)
    w_re, w_im = _weights(oracle, 300 + nt, nt, hidden)
    P = _pilot(rng, nt, orthogonal=False)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('force_tile', tile)
    e.set_option('xcd_order', 1)
    o_re, o_im = e.predict(ltf)
    r_re, r_im = oracle.predict_packets_shared(ltf, P, w_re, w_im)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL
    x = oracle.samples_from_packets(ltf[:9], P.astype(np.float32), 'imag')
    assert rel_rows(e.predict_samples('imag', x), oracle.fc_forward(x, w_im, np.float64)) < TOL
    e.set_option('xcd_order', 0)
    l_re, l_im = e.predict(ltf)
    np.testing.assert_array_equal(l_re, o_re)       # tile order must not change a single bit
    np.testing.assert_array_equal(l_im, o_im)


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(4, 2, 5, (64, 64)), (32, 4, 1, (1024, 1024))])
def test_literal_predict_equals_shared_layer0_path(pkg, oracle, nt, nr, npkt, hidden):
    """Key structural identity: the packet path (layer 0 once per rx antenna + pilot table) and
    the literal Keras predict over [Xsig | Xp] rows agree to rounding, and both match fp64."""
    rng = np.random.default_rng(55 + nt)
    w_re, w_im = _weights(oracle, 99, nt, hidden)
    P = _pilot(rng, nt, orthogonal=False)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    o_re, o_im = e.predict(ltf)
    for d, w, fast in (('real', w_re, o_re), ('imag', w_im, o_im)):
        x = oracle.samples_from_packets(ltf, P.astype(np.float32), d)
        y = e.predict_samples(d, x)
        ref = oracle.fc_forward(x, w, np.float64)
        assert rel_rows(y, ref) < TOL
        assert rel_rows(fast.reshape(y.shape), ref) < TOL
        assert rel_rows(fast.reshape(y.shape), y) < TOL


def test_keras_model_surface_with_reference_batches(pkg, oracle, golden_dir):
    """CSIModel.predict driven exactly like DNN.py:339-346: a Sequence yielding
    ([Xsig (B,lenLTF,1), Xp (B,Nt)], y, None) with B = nTX*nRX, batches taken from the golden
    vectors the reference's DataGenerator produced."""
    g = np.load(os.path.join(golden_dir, 'ref_datagen_nt4.npz'))
    nt, nr, npkt = int(g['nt']), int(g['nr']), int(g['npkt'])
    w_re, w_im = _weights(oracle, 4321, nt, (64, 64))
    P_rows = g['P_matlab']
    e = _engine(pkg, nt, nr, (64, 64), w_re, w_im, P_rows)

    class Seq:                                   # stands in for the reference DataGenerator
        def __init__(self, d):
            self.d = d

        def __len__(self):
            return npkt

        def __getitem__(self, b):
            return [g[f'{self.d}_Xsig'][b], g[f'{self.d}_Xp'][b]], g[f'{self.d}_y'][b], None

    ltf = (g['ds_ltf_real'] + 1j * g['ds_ltf_imag']).reshape(npkt, nr, 320 * nt)
    fast_re, fast_im = e.predict(ltf)
    for d, w, fast in (('real', w_re, fast_re), ('imag', w_im, fast_im)):
        model = pkg.CSIModel(e, d).load_weights(w)
        csi_out = model.predict(Seq(d))
        assert csi_out.shape == (npkt * nt * nr, 234) and csi_out.dtype == np.float32
        x = np.concatenate([g[f'{d}_Xsig'][..., 0], g[f'{d}_Xp']], axis=-1).reshape(npkt * nr * nt, -1)
        ref = oracle.fc_forward(x, w, np.float64)
        assert rel_rows(csi_out, ref) < TOL
        assert rel_rows(fast.reshape(csi_out.shape), ref) < TOL      # packet path, same samples


def test_predict_chunking_is_invisible(pkg, oracle):
    """A tiny workspace forces many packet chunks; results must be bitwise those of one chunk
    whenever the split-K factor is the same, and within tolerance always."""
    rng = np.random.default_rng(77)
    nt, nr, npkt, hidden = 4, 2, 23, (64, 64)
    w_re, w_im = _weights(oracle, 5, nt, hidden)
    P = _pilot(rng, nt)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    big = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    small = _engine(pkg, nt, nr, hidden, w_re, w_im, P, workspace_bytes=5 * (2 * 64 * 4 * 9 + 2 * 4 * 64 * 4))
    a_re, a_im = big.predict(ltf)
    b_re, b_im = small.predict(ltf)
    assert rel_rows(a_re, b_re) < 1e-6 and rel_rows(a_im, b_im) < 1e-6
    r_re, _ = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(b_re, r_re) < TOL


def test_predict_is_deterministic(pkg, oracle):
    rng = np.random.default_rng(78)
    nt, nr, npkt, hidden = 8, 2, 9, (64, 64)
    w_re, w_im = _weights(oracle, 6, nt, hidden)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, _pilot(rng, nt))
    ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
    a = e.predict(ltf)
    b = e.predict(ltf)
    np.testing.assert_array_equal(a[0], b[0])
    np.testing.assert_array_equal(a[1], b[1])


def test_zero_kernels_give_bias(pkg, oracle):
    nt, nr, hidden = 4, 2, (32,)
    w_re, w_im = _weights(oracle, 9, nt, hidden, use_bn=False)
    for w in (w_re, w_im):
        w['fc_regressor.kernel'] = np.zeros_like(w['fc_regressor.kernel'])
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, np.eye(nt), use_bn=False)
    ltf = pkg.synth.white_packets(np.random.default_rng(1), 3, nr, nt)
    o_re, o_im = e.predict(ltf)
    np.testing.assert_array_equal(o_re, np.broadcast_to(w_re['fc_regressor.bias'], o_re.shape))
    np.testing.assert_array_equal(o_im, np.broadcast_to(w_im['fc_regressor.bias'], o_im.shape))


def test_empty_and_error_paths(pkg, oracle):
    nt, nr, hidden = 4, 2, (32, 32)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    ltf = pkg.synth.white_packets(np.random.default_rng(1), 2, nr, nt)
    with pytest.raises(pkg.CsiError) as ex:         # nothing loaded yet
        e.predict(ltf)
    assert ex.value.code == -2
    w_re, w_im = _weights(oracle, 9, nt, hidden)
    e.load_weights('real', w_re)
    e.set_pilot(np.eye(nt))
    with pytest.raises(pkg.CsiError) as ex:         # imag model missing
        e.predict(ltf)
    assert ex.value.code == -2
    e.load_weights('imag', w_im)
    o_re, o_im = e.predict(np.zeros((0, nr, 320 * nt), dtype=np.complex64))
    assert o_re.shape == (0, nr, nt, 234)
    bad = dict(w_re)
    bad['fc_dense1.kernel'] = bad['fc_dense1.kernel'][:, :16]
    with pytest.raises(pkg.CsiError) as ex:
        e.load_weights('real', bad)
    assert ex.value.code == -1
    with pytest.raises(pkg.CsiError):
        e.predict(np.zeros((1, nr, 7), dtype=np.complex64))


def test_hip_path_matches_committed_oracle_fixture(pkg, golden_dir):
    """LS and DNN of the committed Nt=4 fixture (tests/golden/oracle_nt4.npz) without running the oracle."""
    import os
    g = np.load(os.path.join(golden_dir, 'oracle_nt4.npz'))
    w = {tag: {k.split('.', 1)[1]: g[k] for k in g.files if k.startswith(f'w_{tag}.')} for tag in ('re', 'im')}
    e = pkg.CsiEngine(int(g['nt']), int(g['nr']), hidden=(64, 64))
    e.load_weights('real', w['re'])
    e.load_weights('imag', w['im'])
    e.set_pilot(g['P'])
    o_re, o_im = e.predict(g['ltf'])
    assert rel_rows(o_re, g['dnn_real']) < TOL and rel_rows(o_im, g['dnn_imag']) < TOL
    h = e.ls_estimate(g['ltf'])
    assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([g['ls'].real, g['ls'].imag], -1)) < TOL


# ------------------------------------------------------------------------------------ accuracy metric
def test_nmse_metric_matches_oracle(pkg, oracle):
    """NMSE_subk (BER_test_maMIMO_LTF.m:675-686) on the device: host-buffer and device-pointer entry
    points against the oracle; per-link ratios; a ragged link count; determinism."""
    rng = np.random.default_rng(21)
    nt, nr, npkt = 8, 2, 37
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    ref = (rng.standard_normal((npkt, nr, nt, 234)) + 1j * rng.standard_normal((npkt, nr, nt, 234))).astype(np.complex64)
    est = (ref + 0.05 * (rng.standard_normal(ref.shape) + 1j * rng.standard_normal(ref.shape))).astype(np.complex64)
    want = oracle.nmse_subk(ref, est)
    got = e.nmse(ref, est)
    assert abs(got - want) <= 1e-6 * want
    assert e.nmse(ref, est) == got
    assert e.nmse(ref, ref) == 0.0
    # other bin counts (RICE_RENEW has 52 outputs) and a single link
    assert abs(e.nmse(ref[0, 0, 0, :52], est[0, 0, 0, :52]) - oracle.nmse_subk(ref[0, 0, 0, :52], est[0, 0, 0, :52])) <= 1e-6 * want
    d = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]
    for a, h in zip(d, (ref.real, ref.imag, est.real, est.imag)):
        a.upload(np.ascontiguousarray(h, np.float32))
    per = e.empty((npkt * nr * nt,))
    got_d = e.nmse_device(d[0], d[1], d[2], d[3], npkt * nr * nt, 234, per)
    assert got_d == got
    ratios = per.download()
    num = np.sum(np.abs(ref.astype(np.complex128) - est) ** 2, -1).reshape(-1)
    den = np.sum(np.abs(ref.astype(np.complex128)) ** 2, -1).reshape(-1)
    np.testing.assert_allclose(ratios, num / den, rtol=2e-6)
    # the metric the reference reports for a DNN estimate: DNN output against the true channel
    w_re, w_im = _weights(oracle, 5, nt, (8,))
    P = _pilot(rng, nt)
    e.load_weights('real', w_re); e.load_weights('imag', w_im); e.set_pilot(P)
    ltf, H = oracle.make_structured_packets(rng, 6, nr, oracle.hadamard(nt), snr_db=5.0)
    o_re, o_im = e.predict(ltf)
    assert abs(e.nmse(H, o_re + 1j * o_im) - oracle.nmse_subk(H, o_re + 1j * o_im)) <= 1e-5 * oracle.nmse_subk(H, o_re + 1j * o_im)


# ------------------------------------------------------------------------------------ split-f16 engine
HS_CASES = [
    (8, 2, 40, (64, 48)),          # two hidden layers: cast layer 0, fused pair layer (hs out), regressor
    (4, 1, 70, (128,)),            # one hidden layer: the fused pair kernel IS the regressor (fp32 out)
    (16, 2, 9, (64, 32, 48)),      # three hidden layers: hs -> hs generic layer in between
    (32, 4, 9, (1024, 1024)),      # the shipped model; ragged last row tile (1152 rows)
    (12, 3, 11, (48, 80)),         # Nt not a power of two (rows of one (packet, rx) straddle tiles)
    (128, 1, 3, (64, 64)),
    (4, 2, 40, (2048, 32)),        # wide first layer: bn0 vectors of 2048 / 4096 columns behind the LDS ring
    (4, 1, 70, (4096,)),           # (160 KiB in all at 4096, the widest the fused kernel serves)
    (4, 1, 30, (16, 16)),          # one sub-tile per GEMM
]


@pytest.mark.parametrize('nt,nr,npkt,hidden', HS_CASES)
def test_split_f16_engine_matches_fp64_oracle(pkg, oracle, nt, nr, npkt, hidden):
    """fp32 contexts run their large GEMMs on the f16 matrix cores with split (hi + lo) operands
    (gemm_hs.hip.h); 'f32_engine' = 1 forces that engine at any size.  Same 1e-5 contract as the native
    fp32 MFMA kernels, and both engines must agree far inside it."""
    rng = np.random.default_rng(nt * 77 + npkt)
    w_re, w_im = _weights(oracle, 4321 + nt, nt, hidden)
    pow2 = (nt & (nt - 1)) == 0
    P = _pilot(rng, nt, orthogonal=pow2)
    if pow2:
        ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=0.0)[0]
    else:
        ltf = rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('f32_engine', 1)
    s_re, s_im = e.predict(ltf)
    e.set_option('f32_engine', 0)
    n_re, n_im = e.predict(ltf)
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(s_re, r_re) < TOL and rel_rows(s_im, r_im) < TOL
    assert rel_rows(n_re, r_re) < TOL and rel_rows(n_im, r_im) < TOL
    assert rel_rows(s_re, n_re) < 5e-6
    assert not np.array_equal(s_re, n_re), 'the option did not switch engines'
    assert oracle.nmse_subk(r_re + 1j * r_im, s_re + 1j * s_im) < 1e-10


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(32, 4, 9, (1024, 1024)), (8, 2, 70, (64, 256)), (4, 1, 130, (128, 512))])
def test_split_f16_fused_regressor_option(pkg, oracle, nt, nr, npkt, hidden):
    """'hs_fuse_regressor' = 1 (two hidden layers, second width a multiple of 256): the regressor runs inside the pair
    layer's kernel - swapped-operand first stage, activations tile -> A image in LDS, second product on the same
    CU, partial sums of the column tiles combined in a fixed order behind a release / acquire flag.  Same contract,
    run-to-run identical, ragged last row tile, 1 / 2 / 4 column tiles; and the default (two kernels) stays what it was."""
    rng = np.random.default_rng(nt * 31 + npkt)
    w_re, w_im = _weights(oracle, 900 + nt, nt, hidden)
    P = _pilot(rng, nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=3.0)[0]
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('f32_engine', 1)
    assert e.get_option('hs_fuse_regressor') == 0
    u_re, u_im = e.predict(ltf)
    e.set_option('hs_fuse_regressor', 1)
    f_re, f_im = e.predict(ltf)
    r_re, r_im = oracle.predict_packets_shared(ltf.astype(np.complex64), P, w_re, w_im)
    assert rel_rows(f_re, r_re) < TOL and rel_rows(f_im, r_im) < TOL
    assert rel_rows(u_re, r_re) < TOL
    assert not np.array_equal(f_re, u_re) and rel_rows(f_re, u_re) < 5e-6
    g_re, g_im = e.predict(ltf)
    np.testing.assert_array_equal(g_re, f_re)
    np.testing.assert_array_equal(g_im, f_im)
    assert e.get_option('hs_range_fallbacks') == 0


@pytest.mark.parametrize('gain', [1e-3, 1e-2, 1.0, 60.0])
def test_split_f16_engine_input_scale(pkg, oracle, gain):
    """The f16 halves have a finite range: the engine scales operands by powers of two.  Results must
    hold the contract for preambles well below and above unit power."""
    rng = np.random.default_rng(5)
    nt, nr, npkt, hidden = 8, 2, 24, (64, 64)
    w_re, w_im = _weights(oracle, 99, nt, hidden)
    P = _pilot(rng, nt)
    ltf = gain * oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=10.0)[0]
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('f32_engine', 1)
    s_re, s_im = e.predict(ltf)
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(s_re, r_re) < TOL and rel_rows(s_im, r_im) < TOL
    if True:
        assert e.get_option('hs_range_fallbacks') == 0 and e.get_option('hs_launches') > 0       # served by the engine itself


def test_split_f16_engine_is_the_default_for_large_calls(pkg, oracle):
    """Automatic mode: the per-pair layers of a call that fills the chip take the split engine, a small
    call stays on the native kernels (bit-identical to 'f32_engine' = 0)."""
    rng = np.random.default_rng(11)
    nt, nr, hidden = 32, 4, (256, 256)
    w_re, w_im = _weights(oracle, 7, nt, hidden)
    P = _pilot(rng, nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    big = oracle.make_structured_packets(rng, 300, nr, oracle.hadamard(nt), snr_db=5.0)[0]     # 38400 rows
    a_re, _ = e.predict(big)
    e.set_option('f32_engine', 0)
    n_re, _ = e.predict(big)
    assert not np.array_equal(a_re, n_re), 'automatic mode did not use the split engine'
    assert rel_rows(a_re, n_re) < 5e-6
    e.set_option('f32_engine', -1)
    small = big[:2]
    a_re, _ = e.predict(small)
    e.set_option('f32_engine', 0)
    n_re, _ = e.predict(small)
    assert np.array_equal(a_re, n_re)
    e.set_option('f32_engine', -1)
    g_re, _ = e.predict(big)
    r_re, _ = oracle.predict_packets(big[:16].astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=16)
    assert rel_rows(g_re[:16], r_re) < TOL


def test_split_f16_engine_under_graph_replay(pkg, oracle):
    """use_graph: the captured launch sequence of a large call contains the split-engine kernels
    (dynamic LDS, range-guard pointer); replays must reproduce the eager result bit for bit."""
    rng = np.random.default_rng(12)
    nt, nr, npkt, hidden = 16, 4, 200, (128, 64)
    w_re, w_im = _weights(oracle, 8, nt, hidden)
    P = _pilot(rng, nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('f32_engine', 1)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(5, 0, npkt, d_re, d_im)
    o_re, o_im = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.predict_device(d_re, d_im, npkt, o_re, o_im); e.synchronize()
    eager = o_re.download()
    e.set_option('use_graph', 1)
    for _ in range(4):
        e.predict_device(d_re, d_im, npkt, o_re, o_im); e.synchronize()
        assert np.array_equal(o_re.download(), eager)
    assert e.get_option('hs_launches') > 0
    ltf = d_re.download(0, 4) + 1j * d_im.download(0, 4)
    r_re, _ = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=4)
    assert rel_rows(eager[:4], r_re) < TOL


def test_split_f16_engine_range_guard(pkg, oracle):
    """Operands that leave the f16 range after scaling: csi_predict repeats the call on the fp32 MFMA
    kernels by itself (results still inside the contract); after a device-pointer call csi_synchronize
    reports CSI_ERR_RANGE instead of handing back inf / nan silently.  The preamble scale is chosen from
    the data (any input gain is served); the hidden activations use a fixed shift."""
    rng = np.random.default_rng(3)
    nt, nr, npkt, hidden = 8, 2, 20, (64, 64)
    w_re, w_im = _weights(oracle, 17, nt, hidden)
    P = _pilot(rng, nt)
    base = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=10.0)[0]
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('f32_engine', 1)

    def check(ltf, fallbacks):
        s_re, s_im = e.predict(ltf)
        assert e.get_option('hs_range_fallbacks') == fallbacks and e.get_option('f32_engine') == 1
        r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
        assert rel_rows(s_re, r_re) < TOL and rel_rows(s_im, r_im) < TOL
        return r_re

    check(1.0e-12 * base, 0)                            # automatic input scale: tiny and
    check(2.0e2 * base, 0)                              # large preambles stay on the engine
    huge = 3.0e4 * base                                 # ... until the hidden activations overflow
    r_re = check(huge, 1)
    e.set_option('hs_in_shift', 4)                      # fixed input scale: lo halves all denormal -> low-side guard
    check(1.0e-5 * base, 2)
    e.set_option('hs_in_shift', 99)

    d_re, d_im = e.empty((npkt, nr, 320 * nt)), e.empty((npkt, nr, 320 * nt))
    d_re.upload(np.ascontiguousarray(huge.real, np.float32)); d_im.upload(np.ascontiguousarray(huge.imag, np.float32))
    o_re, o_im = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.predict_device(d_re, d_im, npkt, o_re, o_im)
    with pytest.raises(pkg.CsiError) as ei:
        e.synchronize()
    assert ei.value.code == -6 and 'f16' in str(ei.value)
    e.synchronize()                                     # the condition is reported once
    e.set_option('hs_act_shift', -8)                    # a smaller activation scale serves the same data on the engine
    n0 = e.get_option('hs_launches')
    e.predict_device(d_re, d_im, npkt, o_re, o_im)
    e.synchronize()
    assert e.get_option('hs_launches') > n0
    assert rel_rows(o_re.download(), r_re) < TOL


# ------------------------------------------------------------------------------------ bf16 mode
BF16_TOL_IMPL = 4e-3     # vs the bf16-operand emulation: only accumulation-order induced bf16 re-roundings
BF16_TOL_FMT = 3e-2      # vs the fp64 oracle: the format error of 8-bit-mantissa operands (NOT the fp32 contract)


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(8, 2, 6, (64, 64)), (4, 2, 40, (72, 40)), (64, 2, 3, (128, 64)),
                                               (16, 2, 5, (64,)), (8, 1, 9, (64, 32, 48))])
def test_bf16_mode_matches_bf16_emulation(pkg, oracle, nt, nr, npkt, hidden):
    """BASELINE config 3 dtype: bf16 operands, fp32 accumulate.  Checked against the oracle's
    bf16-operand emulation (tight) and against the fp64 oracle (format error, reported as NMSE)."""
    rng = np.random.default_rng(nt + 31 * npkt)
    w_re, w_im = _weights(oracle, 500 + nt, nt, hidden)
    P = _pilot(rng, nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=5.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, dtype='bf16')
    o_re, o_im = e.predict(ltf)
    b_re, b_im = oracle.predict_packets_bf16(ltf, P, w_re, w_im)
    assert rel_rows(o_re, b_re) < BF16_TOL_IMPL and rel_rows(o_im, b_im) < BF16_TOL_IMPL
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(o_re, r_re) < BF16_TOL_FMT and rel_rows(o_im, r_im) < BF16_TOL_FMT
    assert oracle.nmse_subk(r_re + 1j * r_im, o_re + 1j * o_im) < 1e-3
    # literal (un-shared) network in bf16
    x = oracle.samples_from_packets(ltf, P.astype(np.float32), 'real')
    y = e.predict_samples('real', x)
    assert rel_rows(y, oracle.fc_forward_bf16(x, w_re)) < BF16_TOL_IMPL
    # LS is unaffected by the DNN dtype
    h = e.ls_estimate(ltf)
    ref = oracle.ls_estimate(ltf, P)
    assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL


@pytest.mark.parametrize('tile,fused', [(128, 0), (256, 0), (256, 1)])
@pytest.mark.parametrize('nt,nr,npkt,hidden', [(8, 2, 37, (64, 64)), (64, 2, 5, (128, 72)), (4, 2, 70, (96, 40, 24)), (16, 2, 21, (64,))])
def test_bf16_both_tile_kernels(pkg, oracle, tile, fused, nt, nr, npkt, hidden):
    """The 256x256 ping-pong kernel (force_tile=256; with h1 materialised or generated in the kernel) and
    the 128x128 lock-step kernel (128) against the bf16 emulation on ragged shapes: row counts that are no multiple of 256, widths below one tile,
    k-extents that end inside a 32-column sub-tile, split-K slabs of layer 0."""
    rng = np.random.default_rng(7 * nt + npkt)
    w_re, w_im = _weights(oracle, 900 + nt, nt, hidden)
    P = _pilot(rng, nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=8.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, dtype='bf16')
    e.set_option('force_tile', tile)
    e.set_option('bf16_fused_h1', fused)      # 1: h1 generated inside the first per-pair GEMM (incl. a regressor-only model)
    o_re, o_im = e.predict(ltf)
    b_re, b_im = oracle.predict_packets_bf16(ltf, P, w_re, w_im)
    assert rel_rows(o_re, b_re) < BF16_TOL_IMPL and rel_rows(o_im, b_im) < BF16_TOL_IMPL
    x = oracle.samples_from_packets(ltf[:3], P.astype(np.float32), 'imag')
    y = e.predict_samples('imag', x)
    assert rel_rows(y, oracle.fc_forward_bf16(x, w_im)) < BF16_TOL_IMPL
    o2_re, o2_im = e.predict(ltf)
    assert np.array_equal(o_re, o2_re) and np.array_equal(o_im, o2_im)


def test_bf16_mode_shipped_model_slice(pkg, oracle):
    """Nt=64, Nr=4 (config 3 shape), shipped 1024x1024 model, a few packets: exercises the 256x256
    tile kernel through the layer sizes of the real model."""
    rng = np.random.default_rng(64)
    nt, nr, npkt, hidden = 64, 4, 4, (1024, 1024)
    w_re, w_im = _weights(oracle, 640, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = pkg.synth.white_packets(rng, npkt, nr, nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, dtype='bf16')
    o_re, o_im = e.predict(ltf)
    b_re, b_im = oracle.predict_packets_bf16(ltf[:2], P, w_re, w_im)
    assert rel_rows(o_re[:2], b_re) < BF16_TOL_IMPL and rel_rows(o_im[:2], b_im) < BF16_TOL_IMPL


# ------------------------------------------------------------------------------------ twin
def test_csipredictor_twin_mamimo_end_to_end(pkg, oracle, tmp_path):
    rng = np.random.default_rng(31)
    nt, nr, npkt, hidden = 8, 2, 4, (64, 64)
    w_re, w_im = _weights(oracle, 17, nt, hidden)
    P = _pilot(rng, nt)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    for d, w in (('real', w_re), ('imag', w_im)):
        pkg.CSIModel(e, d).load_weights(w).save(str(tmp_path / f'{d}_keras_model'), pilot=P)   # DNN.py:411
    pred = pkg.CSIPredictor(str(tmp_path), experiment='matlab_maMimo')
    ltf, H = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=10.0)
    csi = pred.inference(ltf)
    assert csi.shape == (npkt, nr, nt, 234) and csi.dtype == np.complex64
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=npkt)
    ref = oracle.recombine(r_re, r_im)
    assert rel_rows(np.concatenate([csi.real, csi.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    h_ls = pred.ls_estimate(ltf)
    ref_ls = oracle.ls_estimate(ltf, P)
    assert rel_rows(np.concatenate([h_ls.real, h_ls.imag], -1), np.concatenate([ref_ls.real, ref_ls.imag], -1)) < TOL
    with pytest.raises(SystemExit) as ex:
        pred.inference(ltf.astype(np.complex64))
    assert ex.value.code == -1


def test_reference_model_files_load_and_predict(pkg, oracle, golden_dir, tmp_path):
    """a-10 end to end: ``CSIModel.load_weights('<d>_weights-improvement.hdf5')`` (DNN.py:334) on the
    libhdf5-written Keras checkpoints and ``CSIPredictor(model_path)`` (inference.py:15-16) on SavedModel
    directories ``<d>_keras_model/`` - no h5py / TensorFlow - must predict what the oracle computes from the
    tensors that were written into those files."""
    import shutil
    exp = np.load(os.path.join(golden_dir, 'keras_weights_expected.npz'))
    nt, nr, npkt = int(exp['nt']), 3, 5
    w = {d: {k[len(d) + 1:]: exp[k] for k in exp.files if k.startswith(d + '.')} for d in ('real', 'imag')}
    rng = np.random.default_rng(41)
    P = _pilot(rng, nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=5.0)[0]
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P, w['real'], w['imag'], np.float64, pkt_batch=npkt)
    # Keras HDF5 checkpoints through the keras-shaped model object
    e = pkg.CsiEngine(nt, nr, hidden=(16, 8))
    for d in ('real', 'imag'):
        shutil.copy(os.path.join(golden_dir, f'keras_weights_{d}.hdf5'), tmp_path / f'{d}_weights-improvement.hdf5')
        pkg.CSIModel(e, d).load_weights(str(tmp_path / f'{d}_weights-improvement.hdf5'))
    e.set_pilot(P)
    o_re, o_im = e.predict(ltf)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL
    x = oracle.samples_from_packets(ltf[:2].astype(np.complex64), P.astype(np.float32), 'imag')
    assert rel_rows(pkg.CSIModel(e, 'imag').load_weights(str(tmp_path / 'imag_weights-improvement.hdf5')).predict(x),
                    oracle.fc_forward(x, w['imag'], np.float64)) < TOL
    # SavedModel directories through the deployment wrapper (no config.json, no pilot, no rx count inside)
    pred = pkg.CSIPredictor(os.path.join(golden_dir, 'savedmodel_fixture'), experiment='matlab_maMimo', pilot=P)
    h = pred.inference(ltf.astype(np.complex128))
    assert h.shape == (npkt, nr, nt, 234)
    assert rel_rows(h.real, r_re) < TOL and rel_rows(h.imag, r_im) < TOL
    h1 = pkg.CSIPredictor(os.path.join(golden_dir, 'savedmodel_fixture'), experiment='matlab_maMimo', pilot=P, nr=nr).inference(ltf.astype(np.complex128))
    np.testing.assert_array_equal(h1, h)


def test_csipredictor_twin_rice_renew_single_input(pkg, oracle, golden_dir, tmp_path):
    """The reference's implemented experiment: single-input FC model, 52 outputs re-inserted
    into 64 bins.  Input/recombination/post-processing behaviour is pinned by the golden
    vectors recorded from the reference class itself."""
    import json
    g = np.load(os.path.join(golden_dir, 'ref_inference_rice.npz'), allow_pickle=True)
    x = g['x']
    n_in = x.shape[1]
    rng = np.random.default_rng(3)
    ws = {}
    for d in ('real', 'imag'):
        w = oracle.make_weights(rng, n_in, [32], 52, use_bn=True)
        ws[d] = w
        p = tmp_path / f'{d}_keras_model'
        p.mkdir()
        pkg.save_weight_file(str(p / 'weights.safetensors'), w)
        (p / 'config.json').write_text(json.dumps(dict(nt=0, nr=1, len_ltf=n_in, hidden=[32], n_out=52, use_bn=True)))
    pred = pkg.CSIPredictor(str(tmp_path))                       # experiment='RICE_RENEW' default
    y = pred.inference(x)
    assert y.shape == (x.shape[0], 64) and y.dtype == np.complex128
    ref = oracle.postprocess_rice_renew(oracle.recombine(oracle.fc_forward(x.real, ws['real'], np.float64),
                                                          oracle.fc_forward(x.imag, ws['imag'], np.float64)))
    np.testing.assert_array_equal(y == 0, g['y'] == 0)            # same null pattern as the reference
    nz = ref[:, 1:27]
    assert rel_rows(np.concatenate([y[:, 1:27].real, y[:, 1:27].imag], -1), np.concatenate([nz.real, nz.imag], -1)) < TOL


def test_dataset_label_self_consistency(pkg, oracle):
    """dataset labels are the LS estimate of the same noisy preamble; a dataset in the
    reference's pickle layout, packed by the host code, must satisfy LS(ltf) == labels on the GPU."""
    rng = np.random.default_rng(12)
    nt, nr, npkt = 8, 2, 3
    P_rows = _pilot(rng, nt)
    ltf, _ = oracle.make_structured_packets(rng, npkt, nr, P_rows, snr_db=3.0)
    y = oracle.ls_estimate(ltf, P_rows).reshape(npkt * nr * nt, 234)          # what MATLAB stores as label
    X = np.zeros((npkt * nr * nt, 2), dtype=int)
    LTF = {}
    for p in range(npkt):
        for r in range(nr):
            key = 1000 + p * nr + r
            LTF[key] = {'real': ltf[p, r].real.copy(), 'imag': ltf[p, r].imag.copy()}
            for t in range(nt):
                X[p * nr * nt + r * nt + t] = [key, t]
    ds = {'X': X, 'y': {'real': y.real.copy(), 'imag': y.imag.copy()}, 'LTF': LTF, 'P': P_rows.T.copy(),
          'simParams': {'nTX': nt, 'nRX': nr}}
    packed = pkg.dataset.packets_from_dataset(ds)
    e = pkg.CsiEngine(nt, nr, hidden=(8,))
    assert pkg.dataset.label_consistency(e, packed) < TOL
    packed['pilot'] = packed['pilot'].T.copy()                                # wrong orientation must show
    assert pkg.dataset.label_consistency(e, packed) > 1e-2


def test_cli_test_run_end_to_end(pkg, oracle, tmp_path, capsys):
    """The command-line twin of `DNN.py --test`: pickle dataset + saved models in, evaluate() figure
    and per-packet .mat files out."""
    import pickle
    from scipy.io import loadmat
    rng = np.random.default_rng(77)
    nt, nr, npkt, hidden = 8, 2, 3, (64, 32)
    P_rows = _pilot(rng, nt)
    ltf, _ = oracle.make_structured_packets(rng, npkt, nr, P_rows, snr_db=3.0)
    y = oracle.ls_estimate(ltf, P_rows).reshape(npkt * nr * nt, 234)
    X = np.zeros((npkt * nr * nt, 2), dtype=int)
    LTF = {}
    for p in range(npkt):
        for r in range(nr):
            key = 500 + p * nr + r
            LTF[key] = {'real': ltf[p, r].real.copy(), 'imag': ltf[p, r].imag.copy()}
            for t in range(nt):
                X[p * nr * nt + r * nt + t] = [key, t]
    ds = {'X': X, 'y': {'real': y.real.copy(), 'imag': y.imag.copy()}, 'LTF': LTF, 'P': P_rows.T.copy(),
          'simParams': {'nTX': nt, 'nRX': nr}}
    with open(tmp_path / 'test.b', 'wb') as f:
        pickle.dump(ds, f)
    w_re, w_im = _weights(oracle, 3, nt, hidden)
    model_dir, work = tmp_path / 'model', tmp_path / 'out'
    model_dir.mkdir(); work.mkdir()
    pkg.save_weight_file(str(model_dir / 'real_weights-improvement.safetensors'), w_re)
    pkg.save_weight_file(str(model_dir / 'imag_weights-improvement.safetensors'), w_im)
    from dl_channel_estimation_mamimo_amd import cli
    # the pipeline's own invocation (full_pipeline_maMIMO_DNNEst.sh:47) passes --valSameTrain: every packet is tested
    rc = cli.main(['--test', '-x', str(tmp_path / 'test.b'), '--modeldir', str(model_dir), '-d', str(work), '--nn', '64', '32',
                   '--useBN', '--datasource', 'matlab_maMimo', '--valSameTrain', '--execTime'])
    assert rc == 0
    out = capsys.readouterr().out
    assert 'loss (mse vs labels)' in out and 'LS(GPU) vs stored LS labels' in out and 'pair_dense_gemm' in out
    r_re, r_im = oracle.predict_packets(ltf.astype(np.complex64), P_rows, w_re, w_im, np.float64, pkt_batch=npkt)
    for n in range(npkt):
        m = loadmat(str(work / f'test_csi_predictions_imag_{n + 1}.mat'))['all_pkts_csi_nn_out'][0, 0]
        assert rel_rows(m['y'], r_im[n].reshape(nr * nt, 234)) < TOL
        np.testing.assert_array_equal(m['true_y'], y.imag.reshape(npkt, nr * nt, 234)[n])
    # DNN.py:411: the test run leaves <d>_keras_model/ behind, the folder inference.CSIPredictor loads (inference.py:15-16)
    pred = pkg.CSIPredictor(str(work), experiment='matlab_maMimo')
    h = pred.inference(ltf[:1].astype(np.complex128))
    assert rel_rows(h.real, r_re[:1]) < TOL and rel_rows(h.imag, r_im[:1]) < TOL
    # without --valSameTrain the reference tests the LAST floor(Npkt * valTrainRatio) packets and numbers
    # their files from 1 (DNN.py:125-128, massiveMIMO_dataGenerator.py:46-55)
    work2 = tmp_path / 'out2'
    work2.mkdir()
    rc = cli.main(['--test', '-x', str(tmp_path / 'test.b'), '--modeldir', str(model_dir), '-d', str(work2), '--nn', '64', '32',
                   '--useBN', '--datasource', 'matlab_maMimo', '--valTrainRatio', '0.34'])
    assert rc == 0 and 'Validation separate from Training' in capsys.readouterr().out
    assert sorted(f for f in os.listdir(work2) if f.endswith('.mat')) == ['test_csi_predictions_imag_1.mat', 'test_csi_predictions_real_1.mat']
    m = loadmat(str(work2 / 'test_csi_predictions_real_1.mat'))['all_pkts_csi_nn_out'][0, 0]
    assert rel_rows(m['y'], r_re[npkt - 1].reshape(nr * nt, 234)) < TOL
    np.testing.assert_array_equal(m['true_y'], y.real.reshape(npkt, nr * nt, 234)[npkt - 1])
    # train -> test -> re-train -> test in ONE work directory (round-2 advice): the second test run must evaluate the
    # new checkpoint, not the <d>_keras_model/ folder the first test run left there (DNN.py:279-281,334 always loads
    # <d>_weights-improvement.hdf5)
    work3 = tmp_path / 'out3'
    work3.mkdir()
    pkg.save_weight_file(str(work3 / 'real_weights-improvement.safetensors'), w_re)
    pkg.save_weight_file(str(work3 / 'imag_weights-improvement.safetensors'), w_im)
    args3 = ['--test', '-x', str(tmp_path / 'test.b'), '-d', str(work3), '--nn', '64', '32', '--useBN', '--datasource', 'matlab_maMimo', '--valSameTrain']
    assert cli.main(args3) == 0 and os.path.isdir(work3 / 'real_keras_model')
    w_re2, w_im2 = _weights(oracle, 4, nt, hidden)
    pkg.save_weight_file(str(work3 / 'real_weights-improvement.safetensors'), w_re2)
    pkg.save_weight_file(str(work3 / 'imag_weights-improvement.safetensors'), w_im2)
    assert cli.main(args3) == 0
    capsys.readouterr()
    n_re, n_im = oracle.predict_packets(ltf.astype(np.complex64), P_rows, w_re2, w_im2, np.float64, pkt_batch=npkt)
    m = loadmat(str(work3 / 'test_csi_predictions_real_2.mat'))['all_pkts_csi_nn_out'][0, 0]
    assert rel_rows(m['y'], n_re[1].reshape(nr * nt, 234)) < TOL


def test_host_pipeline_many_chunks_pinned_and_pageable(pkg, oracle):
    """The host-buffer entry points pipeline upload / kernels / download over packet chunks through two
    slots: many chunks (Nt=4, Nr=2: 8192-packet chunks -> use 20000 packets), caller buffers pageable
    or pinned (csi_host_malloc), one or several copy threads - all bit-identical, and right on
    sampled packets."""
    rng = np.random.default_rng(123)
    nt, nr, npkt, hidden = 4, 2, 20000, (32, 16)
    w_re, w_im = _weights(oracle, 77, nt, hidden)
    P = oracle.hadamard(nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    re = rng.standard_normal((npkt, nr, 320 * nt), dtype=np.float32)
    im = rng.standard_normal((npkt, nr, 320 * nt), dtype=np.float32)
    o_re, o_im = e.predict(re, im)
    h = e.ls_estimate(re, im)
    sel = np.r_[0:2, 8191:8194, npkt - 2:npkt]
    ltf = (re[sel] + 1j * im[sel]).astype(np.complex64)
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=len(sel))
    assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL
    ref = oracle.ls_estimate(ltf, P)
    assert rel_rows(np.concatenate([h[sel].real, h[sel].imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    # pinned caller buffers, and a different number of copy threads
    p_re, p_im = e.pinned_empty(re.shape), e.pinned_empty(im.shape)
    p_re[...] = re
    p_im[...] = im
    out = (e.pinned_empty(o_re.shape), e.pinned_empty(o_im.shape))
    q_re, q_im = e.predict(p_re, p_im, out=out)
    assert q_re is out[0] and np.array_equal(q_re, o_re) and np.array_equal(q_im, o_im)
    e.set_option('host_threads', 1)
    s_re, s_im = e.predict(re, im)
    assert np.array_equal(s_re, o_re) and np.array_equal(s_im, o_im)
    hp = e.ls_estimate(re, im, out=(np.empty(o_re.shape, np.float32), np.empty(o_re.shape, np.float32)))
    assert np.array_equal(hp[0], h.real) and np.array_equal(hp[1], h.imag)


# ------------------------------------------------------------------------------------ device path
def test_estimate_c128_matches_plane_entry_points(pkg, oracle):
    """csi_estimate_c128 - complex128 preambles in, complex64 DNN and LS estimates out, one upload, the real / imag
    split and the complex assembly inside the pipeline's staging copies - must return exactly what csi_predict and
    csi_ls_estimate return for the float32 planes of the same packets: many chunks (Nt=4, Nr=2: 8192-packet chunks),
    a ragged last chunk, either output alone, caller-provided buffers, and the deployment wrapper on top of it."""
    rng = np.random.default_rng(21)
    nt, nr, npkt, hidden = 4, 2, 20011, (64, 32)
    w_re, w_im = _weights(oracle, 21, nt, hidden)
    P = _pilot(rng, nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    x = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt)))        # complex128
    p_re, p_im = e.predict(x)
    h = e.ls_estimate(x)
    dnn, ls = e.estimate(x)
    assert dnn.dtype == np.complex64 and dnn.shape == (npkt, nr, nt, 234) and ls.dtype == np.complex64
    np.testing.assert_array_equal(dnn.real, p_re)
    np.testing.assert_array_equal(dnn.imag, p_im)
    np.testing.assert_array_equal(ls, h)
    only_ls = e.estimate(x[:700], dnn=False)
    assert only_ls[0] is None
    np.testing.assert_array_equal(only_ls[1], h[:700])
    buf = np.zeros((3, nr, nt, 234), np.complex64)
    got, none = e.estimate(x[5:8], ls=False, out=(buf, None))
    assert got is buf and none is None
    np.testing.assert_array_equal(buf.real, e.predict(x[5:8])[0])        # (a 3-packet call takes other kernels than the big batch)
    r_re, r_im = oracle.predict_packets(x[:4].astype(np.complex64), P, w_re, w_im, np.float64, pkt_batch=4)
    assert rel_rows(dnn[:4].real, r_re) < TOL and rel_rows(dnn[:4].imag, r_im) < TOL
    with pytest.raises(pkg.CsiError):
        e.estimate(x[:2], dnn=False, ls=False)
    with pytest.raises(pkg.CsiError):
        e.estimate(x[:2, :1])


def test_device_resident_path_and_profile(pkg, oracle):
    rng = np.random.default_rng(41)
    nt, nr, npkt, hidden = 8, 2, 16, (64, 64)
    w_re, w_im = _weights(oracle, 23, nt, hidden)
    P = _pilot(rng, nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(2024, 0, npkt, d_re, d_im)
    d_ore, d_oim = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    d_hre, d_him = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.profile_enable(True)
    e.profile_reset()
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.ls_estimate_device(d_re, d_im, npkt, d_hre, d_him)
    e.synchronize()
    prof = e.profile()
    # round 5: a call of 32 preambles / 256 pair rows takes the one-packet path - ONE launch per layer for both component models
    per_layer = 1 if e.get_option('small_calls') == 1 else 2
    assert prof['pair_dense_gemm']['launches'] == per_layer and prof['pair_dense_gemm']['ms'] > 0
    assert prof['ls_estimate']['launches'] == 1 and prof['regressor_gemm']['launches'] == per_layer
    ltf = d_re.download() + 1j * d_im.download()
    # white generator: unit-variance circular Gaussian, reproducible, offset-consistent
    assert abs(np.mean(np.abs(ltf) ** 2) - 1.0) < 0.02 and abs(np.mean(ltf)) < 0.01
    d_re2, d_im2 = e.empty((4, nr, e.len_ltf)), e.empty((4, nr, e.len_ltf))
    e.synth_white(2024, 5, 4, d_re2, d_im2)
    np.testing.assert_array_equal(d_re2.download(), ltf.real[5:9].astype(np.float32))
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(d_ore.download(), r_re) < TOL and rel_rows(d_oim.download(), r_im) < TOL
    ref_ls = oracle.ls_estimate(ltf, P)
    assert rel_rows(d_hre.download(), ref_ls.real) < TOL and rel_rows(d_him.download(), ref_ls.imag) < TOL


def test_hipgraph_replay_matches_eager(pkg, oracle):
    """use_graph: the 2nd identical csi_predict_device call is captured, later ones replay the
    hipGraph.  Results must equal the eager ones bit for bit, follow new input data written into
    the same buffers, and survive a re-allocation (larger batch) and a weight reload."""
    rng = np.random.default_rng(5)
    nt, nr, npkt, hidden = 8, 2, 6, (256, 64)
    w_re, w_im = _weights(oracle, 41, nt, hidden)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, _pilot(rng, nt))
    a = pkg.synth.white_packets(rng, npkt, nr, nt)
    b = pkg.synth.white_packets(rng, npkt, nr, nt)
    d_re, d_im = e.to_device(a.real), e.to_device(a.imag)
    o_re, o_im = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.predict_device(d_re, d_im, npkt, o_re, o_im); e.synchronize()
    eager_a = (o_re.download(), o_im.download())
    eager_b = e.predict(b)
    e.set_option('use_graph', 1)
    for it in range(4):                       # eager, capture, replay, replay
        o_re.upload(np.zeros((npkt, nr, nt, 234), np.float32))
        e.predict_device(d_re, d_im, npkt, o_re, o_im); e.synchronize()
        np.testing.assert_array_equal(o_re.download(), eager_a[0])
        np.testing.assert_array_equal(o_im.download(), eager_a[1])
    d_re.upload(b.real); d_im.upload(b.imag)  # same pointers, new data -> the graph must see it
    e.predict_device(d_re, d_im, npkt, o_re, o_im); e.synchronize()
    np.testing.assert_array_equal(o_re.download(), eager_b[0])
    big = pkg.synth.white_packets(rng, 40, nr, nt)      # forces a workspace re-allocation
    e.predict(big)
    e.predict_device(d_re, d_im, npkt, o_re, o_im); e.synchronize()
    np.testing.assert_array_equal(o_im.download(), eager_b[1])
    w2_re, w2_im = _weights(oracle, 42, nt, hidden)     # new weights drop the cached graphs
    e.load_weights('real', w2_re); e.load_weights('imag', w2_im)
    for it in range(3):
        e.predict_device(d_re, d_im, npkt, o_re, o_im); e.synchronize()
    e.set_option('use_graph', 0)
    ref = e.predict(b)
    np.testing.assert_array_equal(o_re.download(), ref[0])
    np.testing.assert_array_equal(o_im.download(), ref[1])
    with pytest.raises(pkg.CsiError):
        e.set_option('no_such_option', 1)


def test_hipgraph_config5_scale_multi_chunk(pkg, oracle):
    """BASELINE configs[4] shape (Nt=128, Nr=16, shipped model, "hipGraph-captured batched inference"): 520 packets =
    1 064 960 pair rows per component model, more than the 4 GiB workspace holds, so one call is TWO packet chunks
    - LS kernel, range-guard memsets, magnitude sample, layer 0 (K = 40 960), slab sum, per-pair layer, regressor,
    x 2 models x 2 chunks in ONE captured graph (csi_estimate_device).  Replays must reproduce the eager results bit
    for bit, follow new data in the same buffers, keep the split engine's range-guard bookkeeping alive, and the
    sampled packets must meet the contract."""
    nt, nr, npkt, hidden = 128, 16, 520, (1024, 1024)
    w_re, w_im = _weights(oracle, 128, nt, hidden)
    P = oracle.hadamard(nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(55, 0, npkt, d_re, d_im)
    outs = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]            # dnn re, dnn im, ls re, ls im
    e.estimate_device(d_re, d_im, npkt, *outs)
    e.synchronize()
    eager = [o.download() for o in outs]
    n_eager = e.get_option('hs_launches')
    assert n_eager >= 2 * 2 * 2                                          # two chunks x two models x (layer 0, pair layer + regressor [one band kernel])
    e.set_option('use_graph', 1)
    for it in range(4):                                                  # eager, capture, replay, replay
        for o in outs:
            e._check(e._lib.csi_memcpy_h2d(e._ctx, o.ptr, np.zeros(1024, np.float32).ctypes.data, 4096))    # dirty the heads
        before = e.get_option('hs_launches')
        e.estimate_device(d_re, d_im, npkt, *outs)
        e.synchronize()                                                  # range guard checked after replays too
        assert e.get_option('hs_launches') - before == n_eager
        for o, ref in zip(outs, eager):
            np.testing.assert_array_equal(o.download(), ref)
    assert e.get_option('graph_replays') == 2
    # new data in the same buffers: the graph reads it (and the magnitude sample re-derives the input scale)
    e.synth_white(56, 0, npkt, d_re, d_im)
    e.estimate_device(d_re, d_im, npkt, *outs)
    e.synchronize()
    assert e.get_option('graph_replays') == 3
    pick = [0, 259, 260, npkt - 1]                                       # both sides of the chunk boundary
    for p in pick:
        ltf = d_re.download(p, 1) + 1j * d_im.download(p, 1)
        r_re, r_im = oracle.predict_packets_shared(ltf, P, w_re, w_im)
        assert rel_rows(outs[0].download(p, 1), r_re) < TOL and rel_rows(outs[1].download(p, 1), r_im) < TOL, p
        ref = oracle.ls_estimate(ltf, P)
        assert rel_rows(np.concatenate([outs[2].download(p, 1), outs[3].download(p, 1)], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL, p
    # a range-guard hit inside a replayed graph is still reported
    e.set_option('use_graph', 0)
    e.set_option('use_graph', 1)
    big = e.empty((4, nr, e.len_ltf))
    big.upload(np.full((4, nr, e.len_ltf), 3.0e4, np.float32))
    o4 = [e.empty((4, nr, nt, 234)) for _ in range(4)]
    e.set_option('f32_engine', 1)
    hits = 0
    for it in range(3):
        e.estimate_device(big, big, 4, *o4)
        try:
            e.synchronize()
        except pkg.CsiError as err:
            assert err.code == -6
            hits += 1
    assert hits == 3 and e.get_option('graph_replays') == 4


@pytest.mark.parametrize('engine', [-1, 0])
def test_full_size_properties_config2(pkg, oracle, engine):
    """(engine -1: the library default, i.e. the split-f16 engine at this size; 0: the fp32 MFMA kernels.)
    BASELINE config 2 at FULL size (Nt=32, Nr=4, 4000 device-generated packets = 512 000 pairs,
    shipped model), checked through size-independent properties plus the oracle on a random subset:
      * bit-identical results over two runs (no race in the LDS-DMA ring at full occupancy)
      * packets are independent: a packet alone gives the result it had inside the batch
      * LS is linear over the whole batch (checksum of every output)
      * sampled packets match the fp64 oracle within the contract."""
    rng = np.random.default_rng(2026)
    nt, nr, npkt, hidden = 32, 4, 4000, (1024, 1024)
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = oracle.hadamard(nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('f32_engine', engine)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(99, 0, npkt, d_re, d_im)
    d_ore, d_oim = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.synchronize()
    assert (e.get_option('hs_launches') > 0) == (engine != 0)
    o_re, o_im = d_ore.download(), d_oim.download()
    assert np.isfinite(o_re).all() and np.isfinite(o_im).all()
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.synchronize()
    np.testing.assert_array_equal(d_ore.download(), o_re)
    np.testing.assert_array_equal(d_oim.download(), o_im)
    pick = sorted(rng.choice(npkt, 3, replace=False).tolist()) + [npkt - 1]
    ltf = np.concatenate([d_re.download(p, 1) + 1j * d_im.download(p, 1) for p in pick])
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=len(pick))
    assert rel_rows(o_re[pick], r_re) < TOL and rel_rows(o_im[pick], r_im) < TOL
    # batch independence: the same packets alone take another chunk / tile / split-K geometry,
    # i.e. another fp32 summation order; two fp32 evaluations may differ by the sum of their errors
    s_re, s_im = e.predict(ltf)
    assert rel_rows(s_re, o_re[pick]) < 5e-6 and rel_rows(s_im, o_im[pick]) < 5e-6
    del o_re, o_im
    # LS at full size: oracle on the sample, linearity on everything
    d_hre, d_him = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.ls_estimate_device(d_re, d_im, npkt, d_hre, d_him)
    e.synchronize()
    h = d_hre.download() + 1j * d_him.download()
    ref = oracle.ls_estimate(ltf, P)
    assert rel_rows(np.concatenate([h[pick].real, h[pick].imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    x = d_re.download() + 1j * d_im.download()
    mixed = (0.5 * x + 0.25j * np.roll(x, 1, axis=0)).astype(np.complex64)
    del x
    hm = e.ls_estimate(mixed)
    lin = 0.5 * h + 0.25j * np.roll(h, 1, axis=0)
    num = np.linalg.norm((hm - lin).reshape(npkt, -1), axis=1)
    den = np.linalg.norm(lin.reshape(npkt, -1), axis=1)
    assert float(np.max(num / den)) < 5e-6


def test_config5_shape_long_accumulation(pkg, oracle):
    """BASELINE config 5 shape: Nt=128, Nr=16, shipped model.  Layer 0 accumulates K = 40 960
    products per output; with 512 packets its grid fills the chip without split-K, i.e. the
    LONGEST single fp32 accumulation chain the path can produce.  Sampled packets must still
    meet the contract, the LS estimate (despread-first kernel) too."""
    rng = np.random.default_rng(128)
    nt, nr, npkt, hidden = 128, 16, 512, (1024, 1024)
    w_re, w_im = _weights(oracle, 128, nt, hidden)
    P = oracle.hadamard(nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(11, 0, npkt, d_re, d_im)
    d_ore, d_oim = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.synchronize()
    pick = [0, npkt - 1]
    for p in pick:
        ltf = d_re.download(p, 1) + 1j * d_im.download(p, 1)
        r_re, r_im = oracle.predict_packets_shared(ltf, P, w_re, w_im)
        assert rel_rows(d_ore.download(p, 1), r_re) < TOL and rel_rows(d_oim.download(p, 1), r_im) < TOL
    ltf = d_re.download(3, 2) + 1j * d_im.download(3, 2)
    h = e.ls_estimate(ltf)
    ref = oracle.ls_estimate(ltf, P)
    assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL


def test_full_size_properties_config3_bf16(pkg, oracle):
    """BASELINE config 3 at FULL size (Nt=64, Nr=4, bf16, 5000 device-generated packets = 1 280 000 pairs):
    run-to-run determinism and the bf16-emulation oracle on sampled packets."""
    rng = np.random.default_rng(3)
    nt, nr, npkt, hidden = 64, 4, 5000, (1024, 1024)
    w_re, w_im = _weights(oracle, 64, nt, hidden)
    P = oracle.hadamard(nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, dtype='bf16')
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(5, 0, npkt, d_re, d_im)
    d_ore, d_oim = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.synchronize()
    o_re, o_im = d_ore.download(), d_oim.download()
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.synchronize()
    np.testing.assert_array_equal(d_ore.download(), o_re)
    pick = sorted(rng.choice(npkt, 2, replace=False).tolist())
    ltf = np.concatenate([d_re.download(p, 1) + 1j * d_im.download(p, 1) for p in pick])
    b_re, b_im = oracle.predict_packets_bf16(ltf, P, w_re, w_im)
    assert rel_rows(o_re[pick], b_re) < BF16_TOL_IMPL and rel_rows(o_im[pick], b_im) < BF16_TOL_IMPL




# ------------------------------------------------------------------------------------ round-2 config / input-realism gaps
@pytest.mark.parametrize('engine', [-1, 0])
def test_config4_shape_nt64_nr8(pkg, oracle, engine):
    """BASELINE configs[3] shape - Nt=64, Nr=8, shipped 1024x1024 model - on the HIP path: 256 packets
    (131 072 pair rows: the split-f16 engine engages in automatic mode, layer 0 runs with K = 20 480),
    sampled packets against the shared-layer-0 fp64 oracle, LS (Walsh-Hadamard and generic MFMA despread)
    against the oracle, plus run-to-run determinism.  engine 0 = the fp32 MFMA kernels on the same input."""
    rng = np.random.default_rng(6408)
    nt, nr, npkt, hidden = 64, 8, 256, (1024, 1024)
    w_re, w_im = _weights(oracle, 6408, nt, hidden)
    P = oracle.hadamard(nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('f32_engine', engine)
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    e.synth_white(64, 0, npkt, d_re, d_im)
    # a few structured packets (known channel + noise) among the white ones
    s_ltf = oracle.make_structured_packets(rng, 2, nr, P, snr_db=0.0)[0]
    d_re.upload(np.ascontiguousarray(s_ltf.real, np.float32), first=7)
    d_im.upload(np.ascontiguousarray(s_ltf.imag, np.float32), first=7)
    d_ore, d_oim = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    n0 = e.get_option('hs_launches')
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.synchronize()                                           # raises on a range-guard hit
    assert (e.get_option('hs_launches') > n0) == (engine != 0)
    o_re, o_im = d_ore.download(), d_oim.download()
    e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
    e.synchronize()
    np.testing.assert_array_equal(d_ore.download(), o_re)
    pick = [0, 7, 8, npkt - 1]
    ltf = np.concatenate([d_re.download(p, 1) + 1j * d_im.download(p, 1) for p in pick])
    r_re, r_im = oracle.predict_packets_shared(ltf, P, w_re, w_im)
    assert rel_rows(o_re[pick], r_re) < TOL and rel_rows(o_im[pick], r_im) < TOL
    assert oracle.nmse_subk(r_re + 1j * r_im, o_re[pick] + 1j * o_im[pick]) < 1e-10
    if engine == 0:
        return
    d_hre, d_him = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    ref = oracle.ls_estimate(ltf, P)
    ref2 = np.concatenate([ref.real, ref.imag], -1)
    for kernel in (0, 2):                                     # automatic (Walsh-Hadamard), chunked MFMA despread
        e.set_option('ls_kernel', kernel)
        e.ls_estimate_device(d_re, d_im, npkt, d_hre, d_him)
        e.synchronize()
        h = d_hre.download() + 1j * d_him.download()
        assert rel_rows(np.concatenate([h[pick].real, h[pick].imag], -1), ref2) < TOL, kernel


def test_mixed_snr_batch_config2(pkg, oracle):
    """BASELINE configs[1] as the pipeline runs it: 500 test packets at EACH of {-25..10} dB
    (setenv.sh:19-25, full_pipeline_maMIMO_DNNEst.sh:44-48) - structured channels, the reference's
    amplitude scaling (generate_maMIMO_LTF.m:303-304), signal level fixed and the noise moving by 35 dB -
    in ONE launch of 4000 packets.  The split-f16 engine picks one input scale per launch: the range
    guard must stay silent (no fallback, csi_synchronize clean) and packets of the lowest and of the
    highest SNR level must both meet the contract, on both engines, for the DNN and for LS."""
    nt, nr, hidden = 32, 4, (1024, 1024)
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = pkg.synth.hadamard(nt)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    jobs = pkg.synth.mixed_snr_jobs(2025, per_level=500)
    npkt = jobs[-1][0] + jobs[-1][1]
    assert npkt == 4000 and [j[2] for j in jobs[::2]] == [-25.0, -20.0, -15.0, -10.0, -5.0, 0.0, 5.0, 10.0]
    d_re, d_im = e.empty((npkt, nr, e.len_ltf)), e.empty((npkt, nr, e.len_ltf))
    rms = {}
    for first, snr, blk in pkg.synth.mixed_snr_batch(2025, nr, P, per_level=500):
        d_re.upload(np.ascontiguousarray(blk.real), first=first)
        d_im.upload(np.ascontiguousarray(blk.imag), first=first)
        rms[snr] = float(np.sqrt(np.mean(np.abs(blk) ** 2)))
    assert rms[-25.0] / rms[10.0] > 12.0                      # the batch really spans the amplitude range
    d_ore, d_oim = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    pick = [0, 499, 1750, 3500, npkt - 1]                     # -25 dB (x2), -10 dB, +10 dB (x2)
    ltf = np.concatenate([d_re.download(p, 1) + 1j * d_im.download(p, 1) for p in pick])
    # the uploaded packets are the generator's (any block can be regenerated alone)
    assert np.array_equal(ltf[-1], pkg.synth.mixed_snr_block(jobs[-1], nr, P)[-1])
    r_re, r_im = oracle.predict_packets_shared(ltf, P, w_re, w_im)
    outs = {}
    for engine in (-1, 0):
        e.set_option('f32_engine', engine)
        n0 = e.get_option('hs_launches')
        e.predict_device(d_re, d_im, npkt, d_ore, d_oim)
        e.synchronize()                                       # CSI_ERR_RANGE would raise here
        assert (e.get_option('hs_launches') > n0) == (engine != 0)
        g_re = np.concatenate([d_ore.download(p, 1) for p in pick])
        g_im = np.concatenate([d_oim.download(p, 1) for p in pick])
        assert np.isfinite(g_re).all() and np.isfinite(g_im).all()
        for i in range(len(pick)):                            # per packet: the quiet ones must not hide behind the loud ones
            assert rel_rows(g_re[i], r_re[i]) < TOL and rel_rows(g_im[i], r_im[i]) < TOL, (engine, pick[i])
        outs[engine] = g_re
    assert rel_rows(outs[-1], outs[0]) < 5e-6
    assert e.get_option('hs_range_fallbacks') == 0
    # host-buffer entry point on a slice that mixes the two extreme levels: served by the engine itself
    e.set_option('f32_engine', 1)
    mix = np.concatenate([d_re.download(0, 40) + 1j * d_im.download(0, 40), d_re.download(3960, 40) + 1j * d_im.download(3960, 40)])
    s_re, _ = e.predict(mix)
    assert e.get_option('hs_range_fallbacks') == 0
    m_re, _ = oracle.predict_packets_shared(mix[[0, 79]], P, w_re, w_im)
    assert rel_rows(s_re[0], m_re[0]) < TOL and rel_rows(s_re[79], m_re[1]) < TOL
    # LS over the same batch
    d_hre, d_him = e.empty((npkt, nr, nt, 234)), e.empty((npkt, nr, nt, 234))
    e.ls_estimate_device(d_re, d_im, npkt, d_hre, d_him)
    e.synchronize()
    ref = oracle.ls_estimate(ltf, P)
    h = np.concatenate([d_hre.download(p, 1) + 1j * d_him.download(p, 1) for p in pick])
    for i in range(len(pick)):
        assert rel_rows(np.concatenate([h[i].real, h[i].imag], -1), np.concatenate([ref[i].real, ref[i].imag], -1)) < TOL, pick[i]


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(8, 2, 40, (64, 48)), (32, 4, 24, (1024, 1024))])
def test_heavy_tailed_weights_split_engine(pkg, oracle, nt, nr, npkt, hidden):
    """Weights as a trained model can carry them: a few kernel entries 100x the glorot limit, dead /
    nearly dead BatchNormalization units (gamma ~ 0), units with a large |beta|, a large moving mean and a
    tiny moving variance.  The per-layer operand scales of the split-f16 engine come from max |w| and from
    |beta| + 6 |gamma|; with such tails the bulk of the operands sits far below the top of the f16 range.
    The result must hold the 1e-5 contract - by the engine itself or, if its range guard fires, by the
    automatic repeat on the fp32 MFMA kernels - and the counters must say which."""
    rng = np.random.default_rng(nt + 31)
    w_re, w_im = _weights(oracle, 555 + nt, nt, hidden)
    for w in (w_re, w_im):
        for i in range(len(hidden)):
            k = w[f'fc_dense{i}.kernel']
            lim = np.sqrt(6.0 / sum(k.shape))
            idx = (rng.integers(0, k.shape[0], 12), rng.integers(0, k.shape[1], 12))
            k[idx] = (100.0 * lim * rng.choice([-1.0, 1.0], 12)).astype(np.float32)
            n = k.shape[1]
            dead = rng.choice(n, max(2, n // 16), replace=False)
            w[f'bn{i}.gamma'][dead[: len(dead) // 2]] = 0.0
            w[f'bn{i}.gamma'][dead[len(dead) // 2:]] = 1e-6
            big = rng.choice(n, 3, replace=False)
            w[f'bn{i}.beta'][big] = np.float32([40.0, -25.0, 8.0])
            w[f'bn{i}.moving_mean'][rng.choice(n, 3, replace=False)] = np.float32([6.0, -4.0, 2.5])
            w[f'bn{i}.moving_variance'][rng.choice(n, 3, replace=False)] = np.float32([1e-6, 1e-4, 30.0])
        k = w['fc_regressor.kernel']
        k[rng.integers(0, k.shape[0], 6), rng.integers(0, k.shape[1], 6)] = np.float32(100.0 * np.sqrt(6.0 / sum(k.shape)))
    P = _pilot(rng, nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, oracle.hadamard(nt), snr_db=0.0)[0]
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    r_re, r_im = oracle.predict_packets_shared(ltf.astype(np.complex64), P, w_re, w_im)
    e.set_option('f32_engine', 0)
    n_re, n_im = e.predict(ltf)
    assert rel_rows(n_re, r_re) < TOL and rel_rows(n_im, r_im) < TOL
    e.set_option('f32_engine', 1)
    n0, f0 = e.get_option('hs_launches'), e.get_option('hs_range_fallbacks')
    s_re, s_im = e.predict(ltf)
    assert e.get_option('hs_launches') > n0
    assert rel_rows(s_re, r_re) < TOL and rel_rows(s_im, r_im) < TOL
    fell_back = e.get_option('hs_range_fallbacks') - f0
    assert fell_back in (0, 1)
    if fell_back == 0:
        assert not np.array_equal(s_re, n_re)                 # served by the split engine itself
    else:
        np.testing.assert_array_equal(s_re, n_re)             # guard -> the fp32 MFMA kernels' result
