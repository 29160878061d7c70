"""GPU tests added in round 5 (-m gpu): every call goes through the C-ABI of libcsi_mamimo.so and is checked against the numpy oracle on
identical seeded inputs at the 1e-5 norm-relative contract (BASELINE.json north_star), nothing loosened per test.

  * the one-packet regime (csrc/small_call.hip.h): both component models of a call of at most 8 rx preambles in 1 + n_hidden launches,
  * the 500-packet call of full_pipeline_maMIMO_DNNEst.sh:44-48 (one `--test` run per SNR level),
  * csi_estimate_c64 (complex64 batch in)."""
import numpy as np
import pytest

from conftest import rel_rows
from test_gpu_parity import _engine, _pilot, _weights

pytestmark = pytest.mark.gpu
TOL = 1e-5

SMALL_CASES = [
    # nt, nr, npkt, hidden, use_bn, n_out
    (32, 4, 1, (1024, 1024), True, 234),     # DNN.py:339-346: one packet of the shipped model = 128 rows
    (32, 4, 2, (1024, 1024), True, 234),     # 8 preambles: the largest call the path takes at Nr = 4
    (8, 2, 1, (64, 64), True, 234),          # 16 pair rows: one row tile, three of its four waves without rows
    (8, 2, 4, (64, 64), False, 234),         # --useBN off
    (4, 1, 7, (32, 48, 40), True, 234),      # three hidden layers (ping-pong buffers), K = 1280: the k loop's tail
    (8, 2, 3, (100, 36), True, 234),         # widths that are not multiples of 16: K tail of the tiles over zero-padded weights
    (8, 2, 2, (64,), True, 52),              # single hidden layer: the per-pair layer IS the regressor; 52 outputs (inference.py:58)
    (12, 2, 3, (40, 24), True, 234),         # Nt not a power of two: pair rows straddle (packet, rx) boundaries inside a tile
    (64, 2, 1, (96, 64), True, 234),         # Nt = 64
    (128, 1, 1, (64, 64), True, 234),        # Nt = 128: 128 pair rows from ONE preamble
    (12, 2, 3, (72, 1100), True, 234),       # a wide per-pair layer on the 32 x 32 tiles: ragged rows (72) and columns (1100), K = 72 (3 groups for 16 k-parts); regressor K = 1100
    (32, 4, 8, (1024, 1024), True, 234),     # 8 packets of the shipped model: 32 preambles - layer 0 on the tiles, its epilogue writes the per-pair input (EPI_H1)
    (32, 2, 16, (1024, 1024), True, 234),    # 16 packets of Nr = 2: 32 preambles, 1024 pair rows - the largest call the path takes by default
    (8, 2, 12, (64, 64), True, 234),         # 24 preambles, ragged row tiles
    (4, 1, 40, (32, 48, 40), False, 234),    # 40 preambles of Nt = 4, three hidden layers, no BN, K = 1280
    (12, 2, 9, (40, 24), True, 52),          # 18 preambles, Nt = 12, 52 outputs
    (8, 2, 3, (512, 320), True, 234),        # ("small_fused" = 0 leg: the split-K latency path of every general kernel, K >= 256)
    (8, 2, 1, (256,), False, 234),           # one packet, single hidden layer, no BN (general kernels: split-K regressor)
]


@pytest.mark.parametrize('nt,nr,npkt,hidden,use_bn,n_out', SMALL_CASES)
def test_small_call_path_matches_oracle_and_general_kernels(pkg, oracle, nt, nr, npkt, hidden, use_bn, n_out):
    """small_l0_gemv_kernel + small_tile_gemm_kernel (PAIR / plain, both epilogues) against the fp64 oracle on the same packets, against
    the general kernels ("small_fused" = 0) to rounding, and run-to-run bit-identical (no atomics, fixed summation order)."""
    rng = np.random.default_rng(5000 + nt * 10 + npkt)
    w_re, w_im = _weights(oracle, 4321 + nt, nt, hidden, use_bn, n_out)
    P = _pilot(rng, nt, orthogonal=False)
    ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn, n_out)
    assert e.get_option('small_fused') == 1
    e.set_option('small_rows_band', 65536)     # (the shipped model beyond 512 pair rows goes to the column-split band kernel by default: test below)
    n0 = e.get_option('small_calls')
    o_re, o_im = e.predict(ltf)
    assert e.get_option('small_calls') == n0 + 1, 'a call of %d preambles must take the one-packet path' % (npkt * nr)
    assert o_re.shape == (npkt, nr, nt, n_out) and o_re.dtype == np.float32
    r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
    assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im), 'run-to-run identical'
    e.set_option('small_fused', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('small_calls') == n0 + 2
    assert rel_rows(g_re, r_re) < TOL and rel_rows(g_im, r_im) < TOL
    assert rel_rows(o_re, g_re) < 5e-6 and rel_rows(o_im, g_im) < 5e-6


def test_small_call_limits_and_the_literal_predict(pkg, oracle):
    """More than "small_rows" pair rows (or more than 64 preambles) take the general kernels; the small path equals the literal un-shared network
    (csi_predict_samples: Keras Model.predict semantics, DNN.py:346) on the same packet; device-resident calls and a replayed hipGraph of
    the one-packet call give the same bits as the host call."""
    nt, nr, hidden = 32, 4, (256, 128)
    rng = np.random.default_rng(77)
    w_re, w_im = _weights(oracle, 99, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, 3, nr, P, snr_db=0.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    n0 = e.get_option('small_calls')
    e.set_option('small_rows', 256)
    e.predict(ltf[:3])                                        # 384 pair rows > "small_rows"
    assert e.get_option('small_calls') == n0
    e.set_option('small_rows', 4096)
    big = np.concatenate([ltf] * 6)[:17]                      # 17 packets = 68 preambles: beyond the path's 64 whatever "small_rows" says
    e.predict(big)
    assert e.get_option('small_calls') == n0
    o_re, o_im = e.predict(ltf[:1])
    assert e.get_option('small_calls') == n0 + 1
    # literal network on the assembled samples [LTF ; P_t] of packet 0 (gen.py:299-316 order)
    y_re = e.predict_samples('real', oracle.samples_from_packets(ltf[:1], P, 'real').astype(np.float32))
    y_im = e.predict_samples('imag', oracle.samples_from_packets(ltf[:1], P, 'imag').astype(np.float32))
    assert rel_rows(o_re.reshape(-1, 234), y_re) < 5e-6 and rel_rows(o_im.reshape(-1, 234), y_im) < 5e-6
    # device-resident + graph replay
    d_re, d_im = e.to_device(np.ascontiguousarray(ltf[:1].real)), e.to_device(np.ascontiguousarray(ltf[:1].imag))
    q_re, q_im = e.empty((1, nr, nt, 234)), e.empty((1, nr, nt, 234))
    e.set_option('use_graph', 1)
    g0 = e.get_option('graph_replays')
    for _ in range(4):
        e.predict_device(d_re, d_im, 1, q_re, q_im)
        e.synchronize()
        assert np.array_equal(q_re.download(), o_re) and np.array_equal(q_im.download(), o_im)
    assert e.get_option('graph_replays') >= g0 + 2
    e.set_option('use_graph', 0)


def test_500_packet_call_of_the_pipeline(pkg, oracle):
    """full_pipeline_maMIMO_DNNEst.sh:44-48 hands `--test` the 500 packets of ONE SNR level: that call (64 000 pair rows, the split-f16
    engine + band kernel) against the fp64 oracle on packets spread over the batch, at the lowest and the highest SNR of setenv.sh."""
    nt, nr, hidden, npkt = 32, 4, (1024, 1024), 500
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = oracle.hadamard(nt)
    for snr in (-25.0, 10.0):
        rng = np.random.default_rng(int(2000 + snr))
        ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=snr)[0].astype(np.complex64)
        e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
        o_re, o_im = e.predict(ltf)
        assert e.get_option('hs_launches') > 0 and e.get_option('hs_range_fallbacks') == 0 and e.get_option('band_launches') > 0
        sel = [0, 123, 250, 377, 499]
        r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
        assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL
        h = e.ls_estimate(ltf)
        r_ls = oracle.ls_estimate(ltf[sel], P)
        assert rel_rows(np.concatenate([h[sel].real, h[sel].imag], -1), np.concatenate([r_ls.real, r_ls.imag], -1)) < TOL
        e.close()


BAND_SPLIT_CASES = [
    # nt, nr, npkt, hidden, forced engine
    (32, 4, 24, (1024, 1024), 0),      # 24 bands per model: the automatic mode takes 4 splits
    (32, 4, 64, (1024, 1024), 0),      # 64 bands: 2 splits (two models in flight fill the 256 CUs)
    (24, 2, 37, (256, 1024), 1),       # 1776 rows = 13.9 bands (ragged last band), K1 = 256, pair rows straddle band boundaries
    (16, 2, 9, (128, 512), 1),         # N1 = 512: 2 splits at most; K1 = 128 = the shortest stage 1 the kernel serves
]


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


def test_default_routing_of_small_calls_of_the_shipped_model(pkg, oracle):
    """Nt = 32, Nr = 4, FC 1024 x 1024: up to 2 packets (256 pair rows, 8 preambles) the one-packet path, from 3 packets the general path
    with the weight-streaming layer 0 and the column-split band kernel (measured faster from there, profiles/r05_band_split_probe.txt) -
    each against the fp64 oracle."""
    nt, nr, hidden = 32, 4, (1024, 1024)
    w_re, w_im = _weights(oracle, 1234, nt, hidden)
    P = oracle.hadamard(nt)
    rng = np.random.default_rng(12)
    ltf = oracle.make_structured_packets(rng, 8, nr, P, snr_db=0.0)[0].astype(np.complex64)
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    for npkt, small in ((1, True), (2, True), (3, False), (8, False)):
        s0, b0 = e.get_option('small_calls'), e.get_option('band_split_launches')
        o_re, o_im = e.predict(ltf[:npkt])
        assert (e.get_option('small_calls') == s0 + 1) == small and (e.get_option('band_split_launches') == b0 + 2) == (not small), npkt
        r_re, r_im = oracle.predict_packets(ltf[:npkt], P, w_re, w_im, np.float64, pkt_batch=npkt)
        assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL, npkt
    e.close()


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


def test_estimate_c64_is_bit_identical_with_the_c128_call(pkg, oracle):
    """csi_estimate_c64 (round-4 verdict, next 7): a complex64 batch uploaded as it is and split on the device.  On values that single
    precision represents it must return the bits of csi_estimate_c128 - pageable and pinned input, pageable and pinned result arrays,
    either estimator alone, several pipeline slot sizes (short first / last chunks) - and both agree with the fp64 oracle."""
    nt, nr, hidden, npkt = 8, 2, (64, 64), 150
    rng = np.random.default_rng(64)
    w_re, w_im = _weights(oracle, 640, nt, hidden)
    P = oracle.hadamard(nt)
    x64 = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=3.0)[0].astype(np.complex64)
    x128 = x64.astype(np.complex128)                                   # the same values, widened
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    r_re, r_im = oracle.predict_packets(x64[:3], P, w_re, w_im, np.float64, pkt_batch=3)
    xp = e.pinned_empty(x64.shape, np.complex64)
    xp[...] = x64
    for chunk in (0, 7, 64, 149):
        e.set_option('hp_chunk_packets', chunk)
        # the complex128 call on the same schedule (a chunk's size picks its kernels - split-K factors, the one-packet path - so the bits
        # of two schedules differ by rounding; the two ENTRY POINTS on one schedule must not differ at all)
        ref_dnn, ref_ls = e.estimate(x128)
        assert rel_rows(ref_dnn[:3].real, r_re) < TOL and rel_rows(ref_dnn[:3].imag, r_im) < TOL
        for src in (x64, xp):
            dnn, ls = e.estimate(src)
            assert dnn.dtype == np.complex64 and np.array_equal(dnn, ref_dnn) and np.array_equal(ls, ref_ls), (chunk, src is xp)
            dnn, ls = e.estimate(src, pinned_results=True)
            assert np.array_equal(dnn, ref_dnn) and np.array_equal(ls, ref_ls), (chunk, src is xp, 'pinned results')
            only, none = e.estimate(src, ls=False)
            assert none is None and np.array_equal(only, ref_dnn)
            none, only = e.estimate(src, dnn=False)
            assert none is None and np.array_equal(only, ref_ls)
    e.set_option('hp_chunk_packets', 0)
    with pytest.raises(pkg.CsiError):
        e.estimate(x64[:, :1])                                          # wrong shape
    # the reference's wrapper keeps its complex128 contract (inference.py:39-43)
    assert e.estimate(x128[:2])[0].dtype == np.complex64


def test_tensorflow_written_model_predicts_what_tensorflow_predicted(pkg):
    """The GPU half of the f-1 landing test: once `tools/make_tf_fixture.py` has been run on a TensorFlow host and its
    output committed under tests/golden/tf_written/, the HIP path loads TensorFlow's OWN checkpoint and SavedModel files and must reproduce
    TensorFlow's OWN `Model.predict` on the recorded batch at the 1e-5 contract - the one place the Dense / BatchNormalization arithmetic
    is pinned to the reference's framework instead of to the oracle's restatement of it.  Skips loudly until then."""
    import os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'tf_written')
    if not os.path.exists(os.path.join(d, 'expected.npz')):
        pytest.skip('NO TensorFlow-written model files in tests/golden/tf_written/ (f-1 stays "partial"): run '
                    '`python tools/make_tf_fixture.py tests/golden/tf_written` on a TensorFlow 2.x host and commit the output')
    exp = np.load(os.path.join(d, 'expected.npz'))
    x = np.concatenate([exp['x_sig'][:, :, 0], exp['x_p']], axis=1).astype(np.float32)       # [Flatten(seq_in), seq_p], DNN.py:207-208
    nt = exp['x_p'].shape[1]
    for comp in ('real', 'imag'):
        for path in (os.path.join(d, comp + '_weights-improvement.hdf5'), os.path.join(d, comp + '_keras_model')):
            w = pkg.load_weight_file(path)
            hidden = tuple(int(w['fc_dense%d.bias' % i].shape[0]) for i in range(8) if 'fc_dense%d.bias' % i in w)
            e = pkg.CsiEngine(nt, 1, hidden=hidden, n_out=int(w['fc_regressor.bias'].shape[0]), use_bn=True)
            e.load_weights(comp, w)
            y = e.predict_samples(comp, x)
            assert rel_rows(y, exp[comp + '_y']) < TOL, (path, rel_rows(y, exp[comp + '_y']))
            e.close()


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


def test_fuzz_small_call_shapes(pkg, oracle):
    """Seeded, bounded fuzz of the one-packet path: random antenna counts, preamble counts (1 ... 64, at most 1024 pair rows), one to three
    hidden layers of random widths (multiples of 4, not of the 16 / 32 tiles), with and without BatchNormalization, random output widths -
    every case against the fp64 oracle and against the general kernels, and the path must have been taken."""
    rng = np.random.default_rng(20250930)
    done = 0
    while done < 14:
        nt = int(rng.choice([4, 8, 12, 16, 32, 64]))
        nr = int(rng.integers(1, 5))
        npkt = int(rng.integers(1, 17))
        if npkt * nr > 64 or npkt * nr * nt > 1024:
            continue
        nh = int(rng.integers(1, 4))
        hidden = tuple(int(4 * rng.integers(2, 76)) for _ in range(nh))
        use_bn = bool(rng.integers(0, 2))
        n_out = int(rng.choice([52, 234, 100]))
        w_re, w_im = _weights(oracle, int(rng.integers(1 << 30)), nt, hidden, use_bn, n_out)
        P = _pilot(rng, nt, orthogonal=False)
        ltf = (rng.standard_normal((npkt, nr, 320 * nt)) + 1j * rng.standard_normal((npkt, nr, 320 * nt))).astype(np.complex64)
        e = _engine(pkg, nt, nr, hidden, w_re, w_im, P, use_bn, n_out)
        o_re, o_im = e.predict(ltf)
        case = (nt, nr, npkt, hidden, use_bn, n_out)
        assert e.get_option('small_calls') == 1, case
        r_re, r_im = oracle.predict_packets(ltf, P, w_re, w_im, np.float64, pkt_batch=npkt)
        assert rel_rows(o_re, r_re) < TOL and rel_rows(o_im, r_im) < TOL, case
        e.set_option('small_fused', 0)
        g_re, g_im = e.predict(ltf)
        assert rel_rows(o_re, g_re) < 5e-6 and rel_rows(o_im, g_im) < 5e-6, case
        e.close()
        done += 1


def test_fuzz_mid_size_calls(pkg, oracle):
    """Seeded, bounded fuzz of the mid-size routing: random antenna counts, 9 ... 600 rx preambles, hidden widths that do and do not admit
    the band kernel / its column split - whatever combination of l0_hs_stream_kernel, csi_band8(_cs), the separate split-engine kernels
    and the fp32 MFMA kernels serves the call, the result meets the contract against the fp64 oracle and repeats bit for bit."""
    rng = np.random.default_rng(20251001)
    done, streamed, split = 0, 0, 0
    while done < 12:
        nt = int(rng.choice([16, 32, 64]))
        nr = int(rng.integers(1, 5))
        npkt = int(rng.integers(3, 160))
        if not 9 <= npkt * nr <= 600 or npkt * nr * nt > 40000:
            continue
        h1 = int(rng.choice([128, 192, 208, 256, 320]))
        h2 = int(rng.choice([256, 512, 1024, 96]))
        hidden = (h1, h2)
        w_re, w_im = _weights(oracle, int(rng.integers(1 << 30)), nt, hidden)
        P = oracle.hadamard(nt)
        ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=float(rng.uniform(-10, 20)))[0].astype(np.complex64)
        e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
        o_re, o_im = e.predict(ltf)
        case = (nt, nr, npkt, hidden)
        sel = sorted(set(int(i) for i in rng.integers(0, npkt, 3)) | {0, npkt - 1})
        r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
        assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL, case
        p_re, p_im = e.predict(ltf)
        assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im), case
        streamed += e.get_option('l0_stream_launches') > 0
        split += e.get_option('band_split_launches') > 0
        assert e.get_option('hs_range_fallbacks') == 0, case
        e.close()
        done += 1
    assert streamed >= 6 and split >= 2, (streamed, split)


BF16_TOL_IMPL = 4e-3     # vs the bf16-operand emulation (tests/test_gpu_parity.py: the tolerance of every bf16 kernel)


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
    assert rel_rows(o_re[sel], b_re) < BF16_TOL_IMPL and rel_rows(o_im[sel], b_im) < BF16_TOL_IMPL
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im)
    e.set_option('l0_stream', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('l0_stream_launches') == n0 + 4
    assert rel_rows(o_re, g_re) < BF16_TOL_IMPL and rel_rows(o_im, g_im) < BF16_TOL_IMPL
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
    assert rel_rows(o_re[sel], b_re) < BF16_TOL_IMPL and rel_rows(o_im[sel], b_im) < BF16_TOL_IMPL
    p_re, p_im = e.predict(ltf)
    assert np.array_equal(o_re, p_re) and np.array_equal(o_im, p_im)
    e.set_option('band_split', 0)
    g_re, g_im = e.predict(ltf)
    assert e.get_option('band_split_launches') == n0 + 4
    assert rel_rows(o_re, g_re) < BF16_TOL_IMPL and rel_rows(o_im, g_im) < BF16_TOL_IMPL
    for sp in (2, 4):
        e.set_option('band_split', sp)
        s_re, s_im = e.predict(ltf)
        assert rel_rows(s_re[sel], b_re) < BF16_TOL_IMPL and rel_rows(s_im[sel], b_im) < BF16_TOL_IMPL
    e.close()


def test_mid_size_call_under_a_small_workspace_and_under_graph_replay(pkg, oracle):
    """The mid-size routing in the two situations that change its launch plan: a workspace budget that cuts the call into several packet
    chunks (each chunk takes the streaming layer 0 with its own k ranges and the column-split band kernel) and a captured hipGraph of the
    device call (one stream, so four column splits instead of two) - both against the fp64 oracle and the one-chunk eager call."""
    nt, nr, hidden, npkt = 32, 2, (256, 512), 40
    rng = np.random.default_rng(31)
    w_re, w_im = _weights(oracle, 41, nt, hidden)
    P = oracle.hadamard(nt)
    ltf = oracle.make_structured_packets(rng, npkt, nr, P, snr_db=2.0)[0].astype(np.complex64)
    sel = [0, 17, 39]
    r_re, r_im = oracle.predict_packets(ltf[sel], P, w_re, w_im, np.float64, pkt_batch=len(sel))
    e = _engine(pkg, nt, nr, hidden, w_re, w_im, P)
    o_re, o_im = e.predict(ltf)
    assert e.get_option('l0_stream_launches') == 2 and e.get_option('band_split_launches') == 2
    assert rel_rows(o_re[sel], r_re) < TOL and rel_rows(o_im[sel], r_im) < TOL
    # ~13 packets per chunk: layer-0 slabs (33 k ranges of 26 preambles x 256) dominate the per-packet need
    per_pkt = nr * 256 * 4 * 33 + nr * nt * 512 * 4
    small = _engine(pkg, nt, nr, hidden, w_re, w_im, P, workspace_bytes=13 * per_pkt)
    c_re, c_im = small.predict(ltf)
    assert small.get_option('l0_stream_launches') >= 6, 'several chunks, each on the streaming kernel'
    assert rel_rows(c_re[sel], r_re) < TOL and rel_rows(c_im[sel], r_im) < TOL
    assert rel_rows(c_re, o_re) < 2e-6 and rel_rows(c_im, o_im) < 2e-6
    small.close()
    d_re, d_im = e.to_device(np.ascontiguousarray(ltf.real)), e.to_device(np.ascontiguousarray(ltf.imag))
    q = [e.empty((npkt, nr, nt, 234)) for _ in range(4)]
    e.estimate_device(d_re, d_im, npkt, *q); e.synchronize()
    eager = [a.download() for a in q]
    assert rel_rows(eager[0][sel], r_re) < TOL
    e.set_option('use_graph', 1)
    g0 = e.get_option('graph_replays')
    for _ in range(4):
        e.estimate_device(d_re, d_im, npkt, *q); e.synchronize()
    assert e.get_option('graph_replays') >= g0 + 2
    graph = [a.download() for a in q]
    assert rel_rows(graph[0][sel], r_re) < TOL and rel_rows(graph[1][sel], r_im) < TOL
    assert np.array_equal(graph[2], eager[2]) and np.array_equal(graph[3], eager[3]), 'LS planes: same kernel, same bits'
    assert rel_rows(graph[0], eager[0]) < 2e-6 and rel_rows(graph[1], eager[1]) < 2e-6
    e.set_option('use_graph', 0)
    e.close()
