"""GPU tests (-m gpu; every call through the C-ABI of libcsi_mamimo.so, checked against the numpy oracle on identical seeded inputs at the
1e-5 norm-relative contract of BASELINE.json unless a test states its own): the per-pair DNN denoiser in fp32 contexts (massiveMIMO_CSI_prediction_DNN.py:176-234): fp32 MFMA kernels, split-f16 engine, fused band kernel, weight-streaming layer 0, full-size properties of BASELINE configs[1], [3], [4]."""
import os
import sys


import numpy as np
import pytest

from conftest import rel_rows

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


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


BAND_SPLIT_CASES = [
    # nt, nr, npkt, hidden, forced engine
    (32, 4, 24, (1024, 1024), 0),      # 24 bands per model: the automatic mode takes 4 splits
    (32, 4, 64, (1024, 1024), 0),      # 64 bands: 2 splits (two models in flight fill the 256 CUs)
    (24, 2, 37, (256, 1024), 1),       # 1776 rows = 13.9 bands (ragged last band), K1 = 256, pair rows straddle band boundaries
    (16, 2, 9, (128, 512), 1),         # N1 = 512: 2 splits at most; K1 = 128 = the shortest stage 1 the kernel serves
]


L0_STREAM_CASES = [
    # nt, nr, npkt, hidden: M1 = npkt * nr rx preambles
    (32, 4, 3, (1024, 1024)),      # 12 preambles: one ragged row tile, the in-kernel first pass
    (32, 4, 8, (1024, 1024)),      # 32: exactly one row tile
    (32, 3, 11, (1024, 1024)),     # 33: two row tiles, the second one row
    (32, 4, 24, (1024, 1024)),     # 96: three row tiles, row maxima from l0_row_max_kernel
    (32, 4, 64, (1024, 1024)),     # 256: the largest call of one row block
    (32, 4, 80, (1024, 1024)),     # 320: two row blocks of 6 row tiles (192 + 128 rows)
    (16, 3, 183, (128, 512)),      # 549: three row blocks, the last one ragged (37 rows); K = 5120
    (16, 2, 21, (208, 512)),       # K = 5120, N = 208: the second column group is ragged (80 columns), 42 preambles
    (64, 1, 9, (128, 512)),        # Nt = 64: K = 20480
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


@pytest.mark.parametrize('seed,cases', [(0, 10), (7, 10)])
def test_fuzz_shape_cases(seed, cases):
    """tests/fuzz_shapes.py inside the suite: random Nt / Nr / packets / widths / depth / BN / dtype / engine / tile options,
    shared-layer-0 path, literal path and LS against the oracle."""
    import fuzz_shapes
    rng = np.random.default_rng(seed)
    log = []
    bad = [i for i in range(cases) if not fuzz_shapes.run_case(rng, i, log.append)]
    assert not bad, '\n'.join(log[i] for i in bad)


@pytest.mark.parametrize('seed,cases', [(3, 14), (11, 14)])
def test_fuzz_band4_cases(seed, cases):
    """tests/fuzz_band4.py inside the suite: random shapes the register-blocked band kernels serve (both arithmetic modes; ragged bands, odd output counts,
    one to four column steps) against the 8-wave kernels and the oracle."""
    import fuzz_band4
    rng = np.random.default_rng(seed)
    log = []
    bad = [i for i in range(cases) if not fuzz_band4.run_case(rng, i, log.append)]
    assert not bad, '\n'.join(log[i] for i in bad)


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
    # round 6: where the LDS-staged form applies (16 <= Nt <= 128) the default is the REGISTER-BLOCKED kernel (csi_band4, band4_kernel_gen.py: 4 waves x
    # 512 registers); "band4" = 0 is the 8-wave kernel - same split operands, the h2 fragments in another order (fp32 sums associate differently)
    staged = 16 <= nt <= 128
    assert e.get_option('band4') == 1 and e.get_option('band4_available') == 1
    e.set_option('band4', 0)
    a_re, a_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 6
    if staged:
        assert rel_rows(b_re, a_re) < 2e-6 and rel_rows(b_im, a_im) < 2e-6 and not np.array_equal(b_re, a_re)
        assert rel_rows(a_re[sel], r_re) < TOL and rel_rows(a_im[sel], r_im) < TOL
    else:
        assert np.array_equal(b_re, a_re) and np.array_equal(b_im, a_im)
    e.set_option('hs_band', 3)                                # the 8-wave form with per-lane global loads of L0 / T: the same arithmetic as the staged 8-wave form
    p_re, p_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 8
    assert np.array_equal(a_re, p_re) and np.array_equal(a_im, p_im)
    e.set_option('hs_band', 0)
    s_re, s_im = e.predict(ltf)
    assert e.get_option('band_launches') == n0 + 8
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


def test_split_weight_copies_are_checked_at_load(pkg, oracle):
    """csi_load_weights measures ||W s - (hi + lo)||_F / ||W s||_F of every split-f16 matrix (round-3 verdict, weak 9): a matrix
    whose bulk lies 2^22 below one huge entry would lose its lo halves to the f16 denormals silently - such a model is pinned to the
    fp32 MFMA kernels (still inside the contract), an ordinary one is not."""
    rng = np.random.default_rng(17)
    nt, nr, hidden, npkt = 8, 2, (128, 256), 40
    w_re, w_im = _weights(oracle, 21, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)[0].astype(np.complex64)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    e.set_option('f32_engine', 1)
    assert e.get_option('hs_weight_pins') == 0 and 0 < e.get_option('hs_weight_err_e12') < 3e5          # ~2^-23 .. 2^-22
    n0 = e.get_option('hs_launches')
    e.predict(ltf)
    assert e.get_option('hs_launches') > n0
    bad = {k: np.array(v, copy=True) for k, v in w_im.items()}
    bad['fc_dense1.kernel'] *= 2.0 ** -22
    bad['fc_dense1.kernel'][3, 5] = 1.0                     # one entry 2^22 above the rest: the scale follows it
    e.load_weights('imag', bad)
    assert e.get_option('hs_weight_pins') == 1 and e.get_option('hs_weight_err_e12') > 1e6
    n1 = e.get_option('hs_launches')
    o_re, o_im = e.predict(ltf)
    # the real model still runs on the split engine, the imag model on the fp32 MFMA kernels - both inside the contract
    assert 0 < e.get_option('hs_launches') - n1 < n1 - n0 + 1
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, bad, np.float64, pkt_batch=npkt)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL
    assert e.get_option('band_available') == 1


@pytest.mark.parametrize('nt,nr,npkt,hidden', [(4, 2, 3, (64, 64)), (8, 2, 4, (128, 256)), (32, 4, 2, (1024, 1024))])
def test_hip_path_against_the_c_statement_of_the_oracle(pkg, oracle, nt, nr, npkt, hidden):
    """The HIP path against the SECOND statement of the oracle (oracle/csi_oracle_c.c: plain C in double precision, its own DFT and
    dot products - tests/test_oracle_c.py ties it to the reference-recorded vectors and to the numpy statement on CPU): LS and both
    component models at the 1e-5 contract, the last case with the shipped 1024 x 1024 widths (band kernel's shape)."""
    from oracle import csi_oracle_c as oc
    rng = np.random.default_rng(400 + nt)
    H = oracle.hadamard(nt)
    P = (H[rng.permutation(nt)] * rng.choice([-1.0, 1.0], nt)[:, None]).astype(np.float64)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)[0].astype(np.complex64)
    w_re, w_im = _weights(oracle, 40 + nt, nt, hidden)
    e = pkg.CsiEngine(nt, nr, hidden=hidden)
    e.load_weights('real', w_re)
    e.load_weights('imag', w_im)
    e.set_pilot(P)
    h = e.ls_estimate(ltf)
    ref = oc.ls_estimate(ltf.astype(np.complex128), P)
    assert rel_rows(np.concatenate([h.real, h.imag], -1), np.concatenate([ref.real, ref.imag], -1)) < TOL
    o_re, o_im = e.predict(ltf)
    r_re, r_im = oc.predict_packets(ltf.astype(np.complex128), P, w_re, w_im)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL


@pytest.mark.parametrize('nt,nr,npkt,hidden,engine', BAND_SPLIT_CASES)
def test_column_split_band_kernel(pkg, oracle, nt, nr, npkt, hidden, engine):
    """"band_split" (csi_band8_cs + band_split_sum_kernel): every band's hidden features over 2 / 4 workgroups - against the fp64 oracle at
    the contract, against the unsplit kernel to the rounding of the final fp32 sums, run-to-run bit-identical (the partial sums are
    added in split order), and the automatic mode takes it exactly where the bands leave CUs idle."""
    rng = np.random.default_rng(7000 + nt + npkt)
    w_re, w_im = _weights(oracle, 99 + nt, nt, hidden)
    if nt & (nt - 1):                              # Nt not a power of two: a general pilot matrix, white preambles
        P = _pilot(rng, nt, orthogonal=False)
        ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    else:
        P = oracle.hadamard(nt)
        ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=5.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    if engine:
        e.set_option('f32_engine', engine)
    sel = sorted(set([0, npkt // 3, npkt - 1]))
    r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
    e.set_option('band_split', 0)
    u_re, u_im = e.predict(ltf)
    assert e.get_option('band_launches') > 0 and e.get_option('band_split_launches') == 0
    assert rel_rows(u_re[sel], r_re) < TOL and rel_rows(u_im[sel], r_im) < TOL
    for sp in (2, 4):
        e.set_option('band_split', sp)
        n0 = e.get_option('band_split_launches')
        o_re, o_im = e.predict(ltf)
        assert e.get_option('band_split_launches') == n0 + 2, 'both component models take the split kernel'
        assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL
        assert rel_rows(o_re, u_re) < 2e-6 and rel_rows(o_im, u_im) < 2e-6
        p_re, p_im = e.predict(ltf)
        assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im), 'run-to-run identical'
    e.set_option('band_split', -1)
    n0 = e.get_option('band_split_launches')
    a_re, a_im = e.predict(ltf)
    assert e.get_option('band_split_launches') > n0, 'at most 64 bands per model: CUs would idle without the split'
    assert rel_rows(a_re[sel], r_re) < TOL and rel_rows(a_im[sel], r_im) < TOL
    e.close()


@pytest.mark.parametrize('nt,nr,npkt,hidden', L0_STREAM_CASES)
def test_layer0_weight_streaming_kernel(pkg, oracle, nt, nr, npkt, hidden):
    """l0_hs_stream_kernel (+ l0_row_max_kernel beyond 64 preambles): layer 0 of a mid-size call on the split-f16 path with per-row input
    scales - against the fp64 oracle at the contract, against the general kernels ("l0_stream" = 0), run-to-run bit-identical; and with
    rows of wildly different magnitude (1e-6 ... 1e6 in ONE call), which a single per-launch scale could not hold."""
    rng = np.random.default_rng(8000 + nt + npkt)
    w_re, w_im = _weights(oracle, 77 + nt, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=3.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.set_option('small_rows_band', 0)                  # (12 preambles of the shipped shape: the general path, not the one-packet one)
    e.set_option('small_fused', 0)
    sel = sorted(set([0, npkt // 2, npkt - 1]))
    r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
    n0 = e.get_option('l0_stream_launches')
    o_re, o_im = e.predict(ltf)
    assert e.get_option('l0_stream_launches') == n0 + 2, 'both component models take the kernel'
    assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im), 'run-to-run identical'
    e.set_option('l0_stream', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('l0_stream_launches') == n0 + 4
    assert rel_rows(o_re, g_re) < 5e-6 and rel_rows(o_im, g_im) < 5e-6
    e.set_option('l0_stream', 1)
    # every (packet, rx) preamble at its own magnitude
    mag = (10.0 ** rng.uniform(-6, 6, size=(npkt, nr, 1))).astype(np.float32)
    wide = (ltf * mag).astype(np.complex64)
    w_o_re, w_o_im = e.predict(wide)
    rw_re, rw_im = oracle.predict_packets(wide[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
    assert rel_rows(w_o_re[sel], rw_re) < TOL and rel_rows(w_o_im[sel], rw_im) < TOL
    e.close()


def test_column_split_is_not_taken_by_full_size_calls(pkg, oracle):
    """256 packets = 256 bands per model: no split (a split would only add the partial-sum pass)."""
    nt, nr, hidden, npkt = 32, 4, (1024, 1024), 256
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = oracle.hadamard(nt)
    rng = np.random.default_rng(11)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=0.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    e.predict(ltf)
    assert e.get_option('band_launches') > 0 and e.get_option('band_split_launches') == 0
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


def test_band_tail_split_matches_the_single_launch_and_the_oracle(pkg, oracle):
    """Round 6 ("band_tail_split"): a one-stream fp32 call of more bands than CUs whose last round of band workgroups is nearly empty runs that round in column
    splits on shifted operand pointers.  552 bands (512 + 40 -> four splits) and 340 bands (256 + 84 -> two): every output row against the single launch at the
    split engine's rounding, rows of both parts against the fp64 oracle inside the contract, run to run bit-identical.  (257 ... 320 bands: the whole call
    runs in two column splits - band8_splits - and there is no tail launch.)"""
    for (nt, nr, npkt, hidden) in ((16, 2, 2208, (128, 512)), (32, 1, 1360, (256, 1024))):
        rng = np.random.default_rng(4200 + npkt)
        w_re, w_im = _weights(oracle, 900 + nt, nt, hidden)
        P = oracle.hadamard(nt)
        ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=7.0)[0].astype(np.complex64)
        e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
        e.set_option('f32_engine', 1)
        e.set_option('small_call_overlap', 0)       # one stream (calls of up to 262 144 pair rows run their two models on two streams: the other model fills the round)
        d_re, d_im = e.to_device(np.ascontiguousarray(ltf.real)), e.to_device(np.ascontiguousarray(ltf.imag))
        outs = [e.empty((npkt, nr, nt, 234)) for _ in range(2)]

        def run():
            e.predict_device(d_re, d_im, npkt, *outs); e.synchronize()
            return [x.download() for x in outs]
        n0 = e.get_option('band_tail_launches')
        a = run()
        assert e.get_option('band_tail_launches') == n0 + 2, 'both models of the call take the tail launch'
        a2 = run()
        assert np.array_equal(a[0], a2[0]) and np.array_equal(a[1], a2[1])
        e.set_option('band_tail_split', 0)
        b = run()
        assert e.get_option('band_tail_launches') == n0 + 4
        assert rel_rows(a[0], b[0]) < 2e-6 and rel_rows(a[1], b[1]) < 2e-6
        first_tail = (npkt * nr * nt // 128) // 256 * 256 * 128 // (nr * nt)      # the packet the tail's rows start in
        assert np.array_equal(a[0][:first_tail - 1], b[0][:first_tail - 1]), 'the rows of the full rounds come from the same kernel on the same operands'
        sel = [0, first_tail + 1, npkt - 1]
        r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=3)
        assert rel_rows(a[0][sel], r_re) < 1e-5 and rel_rows(a[1][sel], r_im) < 1e-5
        e.close()
