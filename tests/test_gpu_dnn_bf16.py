"""GPU tests (-m gpu; every call through the C-ABI of libcsi_mamimo.so, checked against the numpy oracle on identical seeded inputs at the
1e-5 norm-relative contract of BASELINE.json unless a test states its own): the DNN path of bf16 contexts (BASELINE configs[2]): bf16 GEMM kernels, band kernels (8-wave and register-blocked), column split, weight-streaming layer 0."""


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


# ------------------------------------------------------------------------------------ bf16 mode
BF16_TOL_IMPL = 4e-3     # vs the bf16-operand emulation: only accumulation-order induced bf16 re-roundings


BF16_TOL_FMT = 3e-2      # vs the fp64 oracle: the format error of 8-bit-mantissa operands (NOT the fp32 contract)


BF16_TOL_IMPL_round5 = 4e-3     # vs the bf16-operand emulation (tests/test_gpu_*.py: the tolerance of every bf16 kernel)


# (nt, nr, packets, hidden, n_out): ragged last bands (M % 128 != 0), rows of a band that straddle three pair rows (nt = 40, 48), one and
# several column steps, the smallest K1 the kernel serves (256), outputs narrower than one 32-column tile group
BAND4_CASES = [(64, 4, 8, (1024, 1024), 234), (32, 4, 5, (256, 512), 234), (48, 2, 7, (512, 256), 234), (40, 3, 11, (384, 768), 234),
               (64, 4, 37, (1024, 1024), 234), (32, 1, 3, (256, 256), 52), (64, 2, 3, (512, 512), 200)]


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


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(64, 4, 1, (1024, 1024)),      # one packet of configs[2]'s shape: 4 preambles
                                               (32, 3, 11, (208, 512)),       # 33 preambles: two row tiles; N = 208 (ragged column group)
                                               (16, 2, 70, (128, 64)),        # 140 preambles, K = 5120
                                               (8, 4, 90, (64, 64))])         # 360 preambles: two row blocks
def test_bf16_layer0_weight_streaming_kernel(pkg, oracle, nt, nr, npkt, hidden):
    """l0_bf16_stream_kernel: layer 0 of small and mid-size calls of a bf16 context - against the oracle's bf16-operand emulation at the
    tolerance of every bf16 kernel, against the kernels it replaces, run-to-run bit-identical."""
    rng = np.random.default_rng(9000 + nt + npkt)
    w_re, w_im = _weights(oracle, 700 + nt, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=6.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, dtype='bf16')
    n0 = e.get_option('l0_stream_launches')
    o_re, o_im = e.predict(ltf)
    assert e.get_option('l0_stream_launches') == n0 + 2
    sel = sorted(set([0, npkt // 2, npkt - 1]))
    b_re, b_im = oracle.predict_packets_bf16(ltf[sel], P, w_re, w_im)
    assert rel_rows(o_re[sel], b_re) < BF16_TOL_IMPL_round5 and rel_rows(o_im[sel], b_im) < BF16_TOL_IMPL_round5
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im)
    e.set_option('l0_stream', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('l0_stream_launches') == n0 + 4
    assert rel_rows(o_re, g_re) < BF16_TOL_IMPL_round5 and rel_rows(o_im, g_im) < BF16_TOL_IMPL_round5
    e.close()


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(64, 4, 1, (1024, 1024)),      # one packet of configs[2]'s shape: 2 bands per model, 4 column splits
                                               (32, 3, 13, (256, 512)),       # 1248 rows = 9.75 bands (ragged), N1 = 512: 2 splits at most
                                               (64, 2, 24, (512, 1024))])     # 24 bands per model
def test_bf16_column_split_band_kernel(pkg, oracle, nt, nr, npkt, hidden):
    """csi_band8_bf16_cs: small calls of a bf16 context on the band kernel in its column-split launch (before: pair_h1 + two 128 x 128 GEMMs) -
    against the bf16-operand emulation, against the kernels it replaces ("band_split" = 0), run-to-run bit-identical."""
    rng = np.random.default_rng(9500 + nt + npkt)
    w_re, w_im = _weights(oracle, 800 + nt, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=4.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, dtype='bf16')
    n0 = e.get_option('band_split_launches')
    o_re, o_im = e.predict(ltf)
    assert e.get_option('band_split_launches') == n0 + 2
    sel = sorted(set([0, npkt // 2, npkt - 1]))
    b_re, b_im = oracle.predict_packets_bf16(ltf[sel], P, w_re, w_im)
    assert rel_rows(o_re[sel], b_re) < BF16_TOL_IMPL_round5 and rel_rows(o_im[sel], b_im) < BF16_TOL_IMPL_round5
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im)
    e.set_option('band_split', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('band_split_launches') == n0 + 4
    assert rel_rows(o_re, g_re) < BF16_TOL_IMPL_round5 and rel_rows(o_im, g_im) < BF16_TOL_IMPL_round5
    for sp in (2, 4):
        e.set_option('band_split', sp)
        s_re, s_im = e.predict(ltf)
        assert rel_rows(s_re[sel], b_re) < BF16_TOL_IMPL_round5 and rel_rows(s_im[sel], b_im) < BF16_TOL_IMPL_round5
    e.close()


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


@pytest.mark.gpu
@pytest.mark.parametrize('nt,nr,npkt,hidden', [(16, 4, 400, (1024, 64)),       # 1600 preambles, K = 5120: 8 x 4 launched tiles, five K ranges
                                               (32, 2, 700, (1024, 128))])     # 1400 preambles, K = 10240: eight ranges
def test_bf16_layer0_fused_kernel_with_k_ranges(pkg, oracle, nt, nr, npkt, hidden):
    """Round 6: between the weight-streaming kernel's range and 256 tiles of the fused 256 x 256 kernel, layer 0 of a bf16 context runs on the fused kernel
    with its K cut into ranges ("bf16_l0_fused_split", default) instead of a cast pass plus the 128 x 128 kernel: against the oracle's bf16-operand
    emulation at the tolerance of every bf16 kernel, against the form it replaces, run to run bit-identical."""
    rng = np.random.default_rng(9100 + nt + npkt)
    w_re, w_im = _weights(oracle, 710 + nt, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=6.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, dtype='bf16')
    n0, s0 = e.get_option('bf16_l0_fused_split_launches'), e.get_option('l0_stream_launches')
    o_re, o_im = e.predict(ltf)
    assert e.get_option('bf16_l0_fused_split_launches') == n0 + 2 and e.get_option('l0_stream_launches') == s0
    sel = sorted(set([0, npkt // 2, npkt - 1]))
    b_re, b_im = oracle.predict_packets_bf16(ltf[sel], P, w_re, w_im)
    assert rel_rows(o_re[sel], b_re) < BF16_TOL_IMPL_round5 and rel_rows(o_im[sel], b_im) < BF16_TOL_IMPL_round5
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im)
    e.set_option('bf16_l0_fused_split', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('bf16_l0_fused_split_launches') == n0 + 4
    assert rel_rows(g_re[sel], b_re) < BF16_TOL_IMPL_round5
    assert rel_rows(o_re, g_re) < BF16_TOL_IMPL_round5 and rel_rows(o_im, g_im) < BF16_TOL_IMPL_round5
    e.close()
